#!/bin/bash
# same-box A/B of the khop3 kernel geometry (groups per wave), serial and pipelined
for g in 4 2 1 auto; do
  if [ $g = auto ]; then unset GGMS_KHOP3_GPW; else export GGMS_KHOP3_GPW=$g; fi
  for mode in "--no-overlap" "--pipelines 2"; do
  python bench.py --no-cpu-baseline $mode "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('GPW=$g $mode', round(d['ms_per_step'],4),'ms', '%.3e'%d['value'], 'extract_us', round(d['roofline']['avg_launch_us'],1), 'sample_ms', round(d['per_gpu']['sample_ms_per_step'],3))"
  done
done
