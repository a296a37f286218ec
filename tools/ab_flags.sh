#!/bin/bash
# A/B of bench.py flag sets on ONE box: tools/ab_flags.sh "--no-overlap" "" ...
for v in "$@"; do
  for rep in 1 2; do
    python bench.py --steps 40 --warmup 5 --no-cpu-baseline $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['per_gpu']
print('[$v] rep$rep: step %.3f ms | sample %.3f | extract %.3f | edges/s %.3e | frac %.3f' % (d['ms_per_step'], p['sample_ms_per_step'], p['extract_ms_per_step'], d['value'], d['roofline']['frac']))"
  done
done
