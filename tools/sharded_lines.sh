#!/bin/bash
# Same-box lines of the default workload through the plain graph view and through logical topology shards
# (DeviceDistGraph): run on the GPU box from the repo root; prints one summary row per line.
# usage: bash tools/sharded_lines.sh <tag> [extra bench flags]
TAG=${1:-r04}; shift
OUT=gpurun_out/sharded_$TAG
mkdir -p $OUT
run() {
  name=$1; shift
  python bench.py --no-engine --no-host-tier --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name FAILED"; tail -3 $OUT/$name.err; return; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rs = d.get("roofline_sampler", {})
print(f"{sys.argv[2]:34s} step {d['ms_per_step']:.4f} ms  edges/s {d['value']:.3e}  sampler alone {rs.get('alone_ms', float('nan')):.4f} ms  "
      f"gather frac {d['roofline']['frac']:.3f} alone {d['roofline']['frac_alone']:.3f}  {d['config'].get('topology')}")
PY
}
run plain "$@"
run shards2_all_cached --dist-graph 1.0 --topology-shards 2 "$@"
run shards8_all_cached --dist-graph 1.0 --topology-shards 8 "$@"
run shards2_half_on_host --dist-graph 0.5 --topology-shards 2 "$@"
run shards8_half_on_host --dist-graph 0.5 --topology-shards 8 "$@"
run shards2_half_on_host_host_indptr --dist-graph 0.5 --topology-shards 2 --host-indptr "$@"
run plain_again "$@"
