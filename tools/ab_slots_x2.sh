#!/bin/bash
# Same box: batch slots with one sampling pipeline and two extract streams (default 3 = K + X)
F="--no-engine --no-configs --no-staged-host --no-host-tier --no-cpu-baseline --no-sampler-roofline --pipelines 1 --extract-streams 2 $*"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); r=d["roofline"]; print("%-8s" % sys.argv[1], "ms/step %.4f" % d["ms_per_step"], "edges/s %.4g" % d["value"], "gather us %.1f" % r["avg_launch_us"], "in flight %.2f" % r["launches_in_flight"], "sample ms %.3f" % d["per_gpu"]["sample_ms_per_step"])'
for rep in 1 2 3; do
  for n in 3 4 5; do
    python bench.py $F --slots $n 2>/dev/null | python -c "$pick" slots_$n
  done
done
