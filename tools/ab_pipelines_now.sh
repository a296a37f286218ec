#!/bin/bash
F="--no-engine --no-configs --no-staged-host --no-host-tier --no-cpu-baseline --no-sampler-roofline $*"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print("%-14s" % sys.argv[1], "ms/step %.4f" % d["ms_per_step"], "edges/s %.4g" % d["value"], "gather us %.1f" % d["roofline"]["avg_launch_us"], "alone %.1f" % (d["roofline"]["avg_launch_us_alone"] or 0), "sample ms %.3f" % d["per_gpu"]["sample_ms_per_step"])'
for rep in 1 2; do
  python bench.py $F 2>/dev/null | python -c "$pick" K1
  python bench.py $F --pipelines 2 2>/dev/null | python -c "$pick" K2
  python bench.py $F --pipelines 2 --slots 4 2>/dev/null | python -c "$pick" K2_slots4
  python bench.py $F --no-overlap 2>/dev/null | python -c "$pick" serial
done
