#!/bin/bash
# Run on the GPU box from the repo root: kernel-trace stats + HBM traffic counters of the DEFAULT bench command.
# usage: bash tools/collect_profiles.sh <tag> [bench flags, e.g. --preset products]
#        (then: python tools/summarize_profiles.py gpurun_out/profiles_<tag> profiles/<round>)
TAG=${1:-r02}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
rm -rf $OUT  # (a directory of an earlier collection would leave its files beside the new ones)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH=$GRAFT_REPO_ROOT/bench.py
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $BENCH "$@" > $OUT/bench_trace.json 2>$OUT/bench_trace.err
echo "trace done"
# PMC passes on their own (no trace domains with --pmc on this pool); FETCH_SIZE and WRITE_SIZE do not fit one pass
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $BENCH --no-cpu-baseline "$@" > $OUT/bench_fetch.json 2>/dev/null
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $BENCH --no-cpu-baseline "$@" > $OUT/bench_write.json 2>/dev/null
echo "write done"
find $OUT -name "*.csv" | head -20
