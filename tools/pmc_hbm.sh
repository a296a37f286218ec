#!/bin/bash
# HBM-side counters per kernel (FETCH_SIZE and WRITE_SIZE need separate passes; no trace domains with --pmc on this pool).
# usage: bash tools/pmc_hbm.sh <tag> [bench flags]   -- prints raw KB per launch, per kernel and grid size
TAG=${1:-hbm}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-host-tier --no-engine --no-sampler-roofline --no-overlap $@"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > /dev/null 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for which in ("fetch", "write"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % which):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "ggms" in k:
                acc[k[:70] + " grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel | launches | FETCH_SIZE KB/launch (raw) | WRITE_SIZE KB/launch")
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0]))):
    fs, ws = c.get("FETCH_SIZE", [0]), c.get("WRITE_SIZE", [0])
    print(f"{k:100s} n={len(fs):3d} fetch {sum(fs)/max(1,len(fs)):12.0f}  write {sum(ws)/max(1,len(ws)):12.0f}")
PY
