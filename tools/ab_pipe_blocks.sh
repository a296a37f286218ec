#!/bin/bash
# same-box sweep: sampling pipelines x gather grid cap
for k in 2 3; do for b in 192 256 384 512 1024; do
  GGMS_EXTRACT_BLOCKS=$b python bench.py --no-cpu-baseline --pipelines $k "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K=$k blocks=$b', round(d['ms_per_step'],4),'ms', '%.3e'%d['value'], 'extract_us', round(d['roofline']['avg_launch_us'],1), 'sample_ms', round(d['per_gpu']['sample_ms_per_step'],3))"
done; done
