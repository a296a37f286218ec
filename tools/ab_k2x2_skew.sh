#!/bin/bash
B="--no-engine --no-configs --no-staged-host --no-host-tier --no-cpu-baseline --no-sampler-roofline"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); r=d["roofline"]; print("%-22s" % sys.argv[1], "ms/step %.4f" % d["ms_per_step"], "edges/s %.4g" % d["value"], "gather us %.1f" % r["avg_launch_us"], "in flight %.2f" % r["launches_in_flight"], "sample ms %.3f" % d["per_gpu"]["sample_ms_per_step"])'
for rep in 1 2; do
  for kx in "1 2" "2 1" "2 2"; do set -- $kx
    python bench.py $B --neighbour-skew 1.0 --pipelines $1 --extract-streams $2 2>/dev/null | python -c "$pick" "papers_skew1 K$1 X$2"
  done
  for kx in "1 1" "2 1" "2 2"; do set -- $kx
    python bench.py $B --preset products --neighbour-skew 1.0 --pipelines $1 --extract-streams $2 2>/dev/null | python -c "$pick" "products_skew1 K$1 X$2"
  done
done
