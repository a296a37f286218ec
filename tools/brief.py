"""stdin: bench.py's JSON line -> one short line (A/B helper)."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r, p = d["roofline"], d["per_gpu"]
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["ms_per_step"], 4), "ms/step", "%.3e" % d["value"], "edges/s |",
      "extract_us", round(r["avg_launch_us"], 1), "frac", round(r["frac"], 3), "alone", round(r.get("frac_alone") or 0, 3),
      "| sample_ms", round(p["sample_ms_per_step"], 3), flush=True)
