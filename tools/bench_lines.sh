#!/bin/bash
# Run on the GPU box from the repo root: the non-default workloads of profiles/README.md, one bench.py line each.
# usage: bash tools/bench_lines.sh <tag>   -> gpurun_out/lines_<tag>/*.json  (copy into profiles/ as rNN_bench_*.json)
TAG=${1:-r03}
OUT=gpurun_out/lines_$TAG
mkdir -p $OUT
run() { # name, flags...
  local name=$1; shift
  python bench.py --no-host-tier "$@" > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $name"
  python tools/brief.py "$name" < $OUT/$name.json
}
run bench_products_sage_25_10 --preset products
run bench_papers100M_skew1 --neighbour-skew 1.0 --no-cpu-baseline
run bench_papers100M_skew1_no_overlap --neighbour-skew 1.0 --no-overlap --no-cpu-baseline --no-engine
run bench_products_skew1 --preset products --neighbour-skew 1.0 --no-cpu-baseline
run bench_products_sage_25_10_no_overlap --preset products --no-overlap --no-cpu-baseline --no-engine
run bench_papers100M_sage_25_10 --fanout 25,10 --no-cpu-baseline
run bench_friendster_pinsage_5_5_5 --preset friendster --sample-type random_walk --fanout 5,5,5 --no-cpu-baseline
for st in khop0 khop2 khop1 weighted_khop weighted_khop_prefix weighted_khop_hash_dedup; do
  run bench_products_$st --preset products --sample-type $st --no-cpu-baseline --no-engine
done
