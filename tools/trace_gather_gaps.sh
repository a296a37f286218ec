#!/bin/bash
# rocprofv3 kernel trace of the default step; prints the dead time between consecutive feature gathers on the extract stream
# usage (GPU box, repo root): bash tools/trace_gather_gaps.sh <tag> [env assignments...] -- [bench flags]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/gaps_$TAG
mkdir -p $OUT
while [ "$1" != "--" ] && [ -n "$1" ]; do export "$1"; shift; done
[ "$1" = "--" ] && shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $GRAFT_REPO_ROOT/bench.py --no-engine --no-configs --no-staged-host --no-host-tier --no-cpu-baseline --no-sampler-roofline --pipelines 1 --extract-streams 1 "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $GRAFT_REPO_ROOT && python tools/gather_gaps.py $OUT
