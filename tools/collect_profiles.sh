#!/bin/bash
# Run on the GPU box from the repo root: kernel-trace stats + HBM traffic counters of bench.py.
# usage: bash tools/collect_profiles.sh <tag>
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/bench_trace.json 2>/dev/null
# PMC passes on their own (no trace domains with --pmc on this pool); FETCH_SIZE and WRITE_SIZE do not fit one pass
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > /dev/null 2>&1
find $OUT -name "*.csv" | head -20
