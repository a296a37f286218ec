#!/bin/bash
# Same box: one / two extract streams on the other workloads (pipelines as the default picks them, or forced)
B="--no-engine --no-configs --no-staged-host --no-host-tier --no-cpu-baseline --no-sampler-roofline"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print("%-28s" % sys.argv[1], "ms/step %.4f" % d["ms_per_step"], "edges/s %.4g" % d["value"], "gather us %.1f" % d["roofline"]["avg_launch_us"], "sample ms %.3f" % d["per_gpu"]["sample_ms_per_step"], d["config"]["streams"][:22])'
run() { # name, flags...
  name=$1; shift
  for rep in 1 2; do
    python bench.py $B "$@" 2>/dev/null | python -c "$pick" "$name X1"
    python bench.py $B --extract-streams 2 "$@" 2>/dev/null | python -c "$pick" "$name X2"
  done
}
run products --preset products --pipelines 1
run papers2510_K1 --fanout 25,10 --pipelines 1
run papers2510_K2 --fanout 25,10 --pipelines 2
run friendster --preset friendster --sample-type random_walk --fanout 5,5,5 --pipelines 1
run products_khop0 --preset products --sample-type khop0
