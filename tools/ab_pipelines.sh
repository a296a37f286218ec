set -e
for st in khop3 khop0 weighted_khop; do
for p in 1 2 3; do
python bench.py --preset products --sample-type $st --pipelines $p --no-engine --no-cpu-baseline --no-host-tier --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$st pipelines $p', 'ms/step', round(d['ms_per_step'],4), 'edges/s %.3e'%d['value'], 'each', ['%.3e'%x for x in d['repeats']['edges_per_s_each']], 'gather frac', round(d['roofline']['frac'],3))
"
done; done
