#!/bin/bash
# same-box A/B over the number of sampling pipelines (cross-box variance is +-5-8 %)
for k in 1 2 3 4; do
  python bench.py --no-cpu-baseline --pipelines $k "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K=$k', round(d['ms_per_step'],4),'ms', '%.3e'%d['value'], 'extract_us', round(d['roofline']['avg_launch_us'],1), 'frac', round(d['roofline']['frac'],3), 'sample_ms', round(d['per_gpu']['sample_ms_per_step'],3))"
done
