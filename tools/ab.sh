#!/bin/bash
# Same-box A/B of bench.py variants (box-to-box variance is +-5-8 %, so variants are compared inside ONE gpurun call).
# Each argument is "ENV=.. ENV2=.. | bench flags"; either side may be empty.  Example:
#   tools/ab.sh " | --pipelines 1" " | --pipelines 2" "GGMS_EXTRACT_BLOCKS=512 | --pipelines 2" " | --no-distinct-seeds"
# The library reads one environment variable, GGMS_EXTRACT_BLOCKS (the gather's grid cap); everything else is a bench flag.
for v in "$@"; do
  envp="${v%%|*}"; flags="${v#*|}"
  for rep in 1 2; do
    env $envp python bench.py --no-cpu-baseline --no-host-tier --no-engine --no-sampler-roofline $flags 2>/dev/null | python tools/brief.py "[$v] rep$rep"
  done
done
