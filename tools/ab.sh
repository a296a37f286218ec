#!/bin/bash
# Same-box A/B of bench.py variants (box-to-box variance is +-5-8 %, so variants are compared inside ONE gpurun call).
# Each argument is "ENV=.. ENV2=.. | bench flags"; either side may be empty.  Example:
#   tools/ab.sh " | --pipelines 1" " | --pipelines 2" "GGMS_EXTRACT_BLOCKS=512 | --pipelines 2" "GGMS_KHOP3_GPW=4 | --no-overlap"
# Hooks: GGMS_OSCAN_GRID, GGMS_SEED_FILL, GGMS_EXTRACT_BLOCKS, GGMS_EXTRACT_DEEP, GGMS_EXTRACT_NT, GGMS_EXTRACT_NT_STORE, GGMS_KHOP3_GPW, GGMS_SCAN, GGMS_GRID_CAP.
for v in "$@"; do
  envp="${v%%|*}"; flags="${v#*|}"
  for rep in 1 2; do
    env $envp python bench.py --no-cpu-baseline --no-host-tier --no-engine --no-sampler-roofline $flags 2>/dev/null | python tools/brief.py "[$v] rep$rep"
  done
done
