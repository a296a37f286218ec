#!/bin/bash
# kernel trace of a short bench run; usage: bash tools/trace_kernels.sh <tag> [bench flags]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-tier --no-engine --no-sampler-roofline --repeats 1 "$@" > $OUT/bench.json 2>/dev/null
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT 23
