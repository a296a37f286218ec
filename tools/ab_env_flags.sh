#!/bin/bash
# usage: tools/ab_env_flags.sh "ENV=.. | flags" ...
for v in "$@"; do
  envp="${v%%|*}"; flags="${v#*|}"
  for rep in 1 2; do
    env $envp python bench.py --steps 40 --warmup 5 --no-cpu-baseline $flags 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['per_gpu']
print('[$v] rep$rep: step %.3f ms | sample %.3f | extract %.3f | edges/s %.3e | frac %.3f' % (d['ms_per_step'], p['sample_ms_per_step'], p['extract_ms_per_step'], d['value'], d['roofline']['frac']))"
  done
done
