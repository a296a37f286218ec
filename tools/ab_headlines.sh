#!/bin/bash
# same-box headline lines: default workload and the small-frontier ones, twice each (tools/brief.py digests)
cd $GRAFT_REPO_ROOT
B="python bench.py --no-engine --no-host-tier --no-cpu-baseline --no-configs --no-sampler-roofline"
for rep in 1 2; do
  $B 2>/dev/null | python tools/brief.py "default"
  $B --preset products 2>/dev/null | python tools/brief.py "products"
  $B --fanout 25,10 2>/dev/null | python tools/brief.py "papers100M [25,10]"
  $B --preset products --sample-type khop0 2>/dev/null | python tools/brief.py "products khop0"
done
