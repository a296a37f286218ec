#!/bin/bash
# A/B of env-selected variants on ONE box: tools/ab.sh "VAR=a" "VAR=b" ...  (each run: bench.py 40 steps)
for v in "$@"; do
  for rep in 1 2; do
    env $v python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['per_gpu']
print('$v rep$rep: step %.3f ms | sample %.3f | extract %.3f' % (d['ms_per_step'], p['sample_ms_per_step'], p['extract_ms_per_step']))"
  done
done
