#!/bin/bash
# Same box: one / two extract streams (--extract-streams) x one / two sampling pipelines.
F="--no-engine --no-configs --no-staged-host --no-host-tier --no-cpu-baseline --no-sampler-roofline $*"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print("%-10s" % sys.argv[1], "ms/step %.4f" % d["ms_per_step"], "edges/s %.4g" % d["value"], "gather us %.1f" % d["roofline"]["avg_launch_us"], "sample ms %.3f" % d["per_gpu"]["sample_ms_per_step"])'
for rep in 1 2 3; do
  python bench.py $F --pipelines 1 --extract-streams 1 2>/dev/null | python -c "$pick" K1_X1
  python bench.py $F --extract-streams 2 --pipelines 1 2>/dev/null | python -c "$pick" K1_X2
  python bench.py $F --pipelines 2 --extract-streams 1 2>/dev/null | python -c "$pick" K2_X1
  python bench.py $F --extract-streams 2 --pipelines 2 2>/dev/null | python -c "$pick" K2_X2
done
