"""Dead time between consecutive feature gathers of a traced bench run (tools/trace_gather_gaps.sh)."""
import csv
import glob
import json
import statistics
import sys

out = sys.argv[1]
best = None
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    g = [r for r in rows if "k_gather_rows<16" in r["Kernel_Name"] and "PlainRows" not in r["Kernel_Name"]]
    if best is None or len(g) > len(best[1]):
        best = (rows, g)
rows, g = best
g.sort(key=lambda r: int(r["Start_Timestamp"]))
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(g, g[1:])]
durs = [(int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3 for a in g]
steady = [x for x in gaps if 0.5 < x < 200]  # (block boundaries are milliseconds, the back-to-back `alone` loop is 0)
line = json.loads([l for l in open(out + "/bench.json") if l.startswith("{")][-1])
print(f"gathers {len(g)}; gap between consecutive gathers in the pipeline: median {statistics.median(steady):.1f} us "
      f"(min {min(steady):.1f}, max {max(steady):.1f}, n {len(steady)}); gather median {statistics.median(durs):.1f} us; "
      f"bench: {line['ms_per_step']:.4f} ms/step, gather {line['roofline']['avg_launch_us']:.1f} us (its own timing)")
print("gaps:", [round(x, 1) for x in gaps])
