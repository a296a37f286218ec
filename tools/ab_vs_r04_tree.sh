#!/bin/bash
# same-box A/B against round 4's tree (build/_r04: `git archive e784987` + make), headline lines only
cd $GRAFT_REPO_ROOT
F="--no-engine --no-host-tier --no-cpu-baseline --no-sampler-roofline"
run() { # label, dir, flags...
  local label=$1 dir=$2; shift 2
  (cd $dir && python bench.py $F "$@" 2>/dev/null) | python tools/brief.py "$label"
}
for rep in 1 2; do
  run "r04 papers100M khop0" build/_r04 --sample-type khop0
  run "r05 papers100M khop0" . --no-configs --sample-type khop0
  run "r04 papers100M khop0 hub-skewed" build/_r04 --sample-type khop0 --neighbour-skew 1.0
  run "r05 papers100M khop0 hub-skewed" . --no-configs --sample-type khop0 --neighbour-skew 1.0
  run "r04 products khop0" build/_r04 --preset products --sample-type khop0
  run "r05 products khop0" . --no-configs --preset products --sample-type khop0
  run "r04 products khop0 hub-skewed" build/_r04 --preset products --sample-type khop0 --neighbour-skew 1.0
  run "r05 products khop0 hub-skewed" . --no-configs --preset products --sample-type khop0 --neighbour-skew 1.0
done
