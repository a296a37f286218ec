#!/bin/bash
# Same box: HIP stream priorities on the sampling / extract streams (GGMS_BENCH_STREAM_PRIORITY), default = none.
F="--no-engine --no-configs --no-staged-host --no-host-tier --no-cpu-baseline --no-sampler-roofline --pipelines 1 --extract-streams 1 $*"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print("%-8s" % sys.argv[1], "ms/step %.4f" % d["ms_per_step"], "edges/s %.4g" % d["value"], "gather us %.1f" % d["roofline"]["avg_launch_us"], "sample ms %.3f" % d["per_gpu"]["sample_ms_per_step"])'
for rep in 1 2 3; do
  python bench.py $F 2>/dev/null | python -c "$pick" none
  GGMS_BENCH_STREAM_PRIORITY=sample python bench.py $F 2>/dev/null | python -c "$pick" sample
  GGMS_BENCH_STREAM_PRIORITY=extract python bench.py $F 2>/dev/null | python -c "$pick" extract
done
