#!/bin/bash
# same-box A/B of the step's stream layout: round 4's bench.py (label gather + count totals + a memset node on the extract
# stream) against this round's (the extract stream runs the feature gather alone), same library, headline only
cd $GRAFT_REPO_ROOT
F="--no-engine --no-host-tier --no-cpu-baseline --no-sampler-roofline"
for rep in 1 2 3; do
  python bench_r04_ab.py $F 2>/dev/null | python tools/brief.py "r04 bench.py default"
  python bench.py $F --no-configs 2>/dev/null | python tools/brief.py "r05 bench.py default"
  python bench_r04_ab.py $F --preset products 2>/dev/null | python tools/brief.py "r04 bench.py products"
  python bench.py $F --no-configs --preset products 2>/dev/null | python tools/brief.py "r05 bench.py products"
done
