#!/bin/bash
# usage (on the GPU box, from repo root): bash scripts_gpu_round.sh <tag>
set -o pipefail
TAG=${1:-r01}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_$TAG.log 2>&1; rc=$?; echo "pytest exit=$rc" >> gpurun_out/gpu_tests_$TAG.log; tail -4 gpurun_out/gpu_tests_$TAG.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; rc=$?; tail -2 gpurun_out/smoke_$TAG.log
[ $rc -eq 0 ] || exit $rc
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; rc=$?; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
[ $rc -eq 0 ] || exit $rc
