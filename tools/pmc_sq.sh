#!/bin/bash
# SQ counters of the sampling kernels, serial run (one pass, 8 SQ slots); usage: bash tools/pmc_sq.sh <tag> [bench flags]
TAG=${1:-sq}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
  --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-host-tier --no-engine --no-sampler-roofline --no-overlap "$@" > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "ggms" in k and "gather_rows" not in k and "init_states" not in k:
        acc[k[:44] + " grid=" + r["Grid_Size"] + " wg=" + r["Workgroup_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    print(k)
    for n, v in c.items():
        print(f"   {n:22s} avg {sum(v)/len(v):14.0f}  (n={len(v)})")
PY
