#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_gather_calibration.sh <tag> [dims]   -> gpurun_out/pmc_gather_<tag>.txt
TAG=${1:-r04}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gather_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python $GRAFT_REPO_ROOT/tools/pmc_gather_calibration.py "$@" > $OUT/run.txt 2>/dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python $GRAFT_REPO_ROOT/tools/pmc_gather_calibration.py "$@" > /dev/null 2>&1
python - "$OUT" "$@" <<'PY'
import csv, glob, sys
out, dims = sys.argv[1], [int(x) for x in (sys.argv[2:] or ["100", "128"])]
n = 1_280_000
vals = {}
for which in ("fetch", "write"):
    rows = []
    for f in glob.glob(f"{out}/{which}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_gather_rows" in r["Kernel_Name"] and "PlainRows" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    vals[which] = [v for _, v in rows]
print("row bytes | counter | table order (known bytes) | random rows | random / table order | raw counter KB / known KB")
for i, dim in enumerate(dims):
    rb = dim * 4
    known_kb = n * rb / 1024
    for which, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        v = vals[which][4 * i:4 * i + 4]
        if len(v) < 4:
            print(f"{rb} | {name} | missing launches: {v}")
            continue
        seq, rnd = (v[0] + v[1]) / 2, (v[2] + v[3]) / 2
        print(f"{rb:5d} | {name} | {seq:12.0f} KB | {rnd:12.0f} KB | {rnd / seq:.3f} | {seq / known_kb:.3f}")
PY
