#!/bin/bash
# Same box, back to back: round 4's tree (build/_r04: `git archive e784987` + make) against this one, headline lines of the
# single-GPU BASELINE workloads (python bench.py <flags>; this round's streams per workload by the trial)
cd $GRAFT_REPO_ROOT
F="--no-engine --no-host-tier --no-cpu-baseline --no-sampler-roofline"
run() { # label, dir, flags...
  local label=$1 dir=$2; shift 2
  (cd $dir && python bench.py $F "$@" 2>/dev/null) | python tools/brief.py "$label"
}
for rep in 1 2; do
  run "r04 default (papers100M GCN [5,10,15])" build/_r04
  run "r05 default (papers100M GCN [5,10,15])" . --no-configs --no-staged-host
  run "r04 products [25,10]" build/_r04 --preset products
  run "r05 products [25,10]" . --no-configs --no-staged-host --preset products
  run "r04 papers100M [25,10]" build/_r04 --fanout 25,10
  run "r05 papers100M [25,10]" . --no-configs --no-staged-host --fanout 25,10
  run "r04 friendster PinSAGE" build/_r04 --preset friendster --sample-type random_walk --fanout 5,5,5
  run "r05 friendster PinSAGE" . --no-configs --no-staged-host --preset friendster --sample-type random_walk --fanout 5,5,5
  run "r04 products khop0" build/_r04 --preset products --sample-type khop0
  run "r05 products khop0" . --no-configs --no-staged-host --preset products --sample-type khop0
  run "r04 papers100M hub-skewed" build/_r04 --neighbour-skew 1.0
  run "r05 papers100M hub-skewed" . --no-configs --no-staged-host --neighbour-skew 1.0
done
