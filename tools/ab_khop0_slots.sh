#!/bin/bash
# same-box A/B: products khop0, default (distinct-seed promise) vs --no-distinct-seeds, by number of batch slots
cd $GRAFT_REPO_ROOT
B="python bench.py --preset products --sample-type khop0 --no-engine --no-host-tier --no-cpu-baseline --no-sampler-roofline"
for rep in 1 2 3; do
  for slots in 3 4 5; do
    for mode in "" "--no-distinct-seeds"; do
      $B --slots $slots $mode --host-profile 2> /tmp/err.txt | python tools/brief.py "slots=$slots ${mode:-distinct}"
      grep "host enqueue" /tmp/err.txt
    done
  done
done
