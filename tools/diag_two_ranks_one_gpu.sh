echo "== N=1 across the epoch boundary (183 steps, 151 per epoch)"
timeout -k 10 300 python bench.py --steps 60 --no-engine --no-cpu-baseline --no-host-tier --no-sampler-roofline 2>&1 | cut -c1-300
echo "rc $?"
export GGMS_BENCH_DEVICE=0 GGMS_BENCH_BACKEND=gloo
for i in 1 2 3; do
echo "== N=2 on one GPU, peer only, 3 x 30 steps (76 per epoch), run $i"
timeout -k 10 300 python bench.py --gpus 2 --steps 30 --other-stores "" --no-engine 2>&1 | grep "^{\|device status\|Error" | cut -c1-300
done
for i in 1 2 3; do
echo "== N=2 on one GPU, replica as the main store, 3 x 20 steps, run $i"
timeout -k 10 300 python bench.py --gpus 2 --store replica --other-stores "" --no-engine 2>&1 | grep "^{\|device status\|Error" | cut -c1-300
done
