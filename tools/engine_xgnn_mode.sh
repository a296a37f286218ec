#!/bin/bash
# XGNN mode through the operator surface at papers100M size, rehearsed on ONE GPU: the reference's quick-start shape
# (/root/reference README.md:184: arch6, --part-cache --gpu-extract --use-dist-graph 1.0 --cache-percentage 0.64,
# batch 6000, khop3 GCN) with W engine workers forked by tools/engine_epoch.py, all on device 0
# (SAMGRAPH_FORCE_DEVICE): topology shards + feature shards behind hipIpc, host tier zero-copy.  Beside it the same
# deployment with the whole CSR on every worker and with half of the edges' nodes in the host-CSR slot.
# usage (GPU box, repo root): bash tools/engine_xgnn_mode.sh [workers, default 4]
W=${1:-4}
D=/dev/shm/ggms_papers_ds
export HSA_ENABLE_IPC_MODE_LEGACY=0 SAMGRAPH_FORCE_DEVICE=0 SAMGRAPH_IPC_TIMEOUT_S=240
python - <<PY
import sys, time
sys.path.insert(0, ".")
from xgnn_amd import datagen
t0 = time.time()
g = datagen.make_graph("papers100M", seed=42)
datagen.write_dataset("$D", g, minimal=True)
print(f"dataset written in {time.time() - t0:.0f} s", g["meta"], flush=True)
PY
run() {
  name=$1; shift
  t0=$(date +%s)
  timeout -k 10 500 python tools/engine_epoch.py $D --fanout 5 10 15 --batch-size 6000 --arch6 $W "$@" > /tmp/xgnn_$name.json 2> /tmp/xgnn_$name.err
  rc=$?
  echo "== $name (rc $rc, $(( $(date +%s) - t0 )) s): $*"
  if [ $rc -eq 0 ]; then tail -1 /tmp/xgnn_$name.json; else tail -5 /tmp/xgnn_$name.err; fi
}
run whole_csr --cache-percentage 0.64
run xgnn_quickstart --cache-percentage 0.64 --use-dist-graph 1.0
run half_on_host --cache-percentage 0.64 --use-dist-graph 0.5
rm -rf $D
