#!/bin/bash
# Same box: the gather's grid (GGMS_EXTRACT_BLOCKS, default 256 = one 4-wave workgroup per CU) with TWO extract streams
F="--no-engine --no-configs --no-staged-host --no-host-tier --no-cpu-baseline --no-sampler-roofline --pipelines 1 --extract-streams 2 $*"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); r=d["roofline"]; print("%-10s" % sys.argv[1], "ms/step %.4f" % d["ms_per_step"], "edges/s %.4g" % d["value"], "gather us %.1f" % r["avg_launch_us"], "in flight %.2f" % r["launches_in_flight"], "alone %.1f" % r["avg_launch_us_alone"], "sample ms %.3f" % d["per_gpu"]["sample_ms_per_step"])'
for rep in 1 2; do
  for nb in 256 128 192 384 512; do
    GGMS_EXTRACT_BLOCKS=$nb python bench.py $F 2>/dev/null | python -c "$pick" blocks_$nb
  done
done
