for rep in 1 2; do
for v in "16 1" "32 1"; do
set -- $v
GGMS_HASH_DEDUP_G=$1 GGMS_HASH_DEDUP_R=$2 python bench.py --preset products --sample-type weighted_khop_hash_dedup --no-engine --no-cpu-baseline --no-host-tier --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
rs=d.get('roofline_sampler') or {}
print('[G=$1 R=$2] rep$rep ms/step', round(d['ms_per_step'],4), 'edges/s %.3e'%d['value'], 'sampler alone ms', round(rs.get('alone_ms',0),4), 'rows ok', d['rows_verified'])
"
done; done
