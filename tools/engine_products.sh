#!/bin/bash
# Engine-level throughput on a products-shaped dataset written in the reference's on-disk format
# (run on the GPU box): the C++ driver over the samgraph_* ABI, arch1, khop3, batch 8000, cache 100 %.
set -e
D=/tmp/products_ds
python - <<PY
import numpy as np, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from xgnn_amd import datagen
g = datagen.make_graph("products", seed=42)
n, dim = g["meta"]["num_node"], g["meta"]["feat_dim"]
feat = (np.arange(n * dim, dtype=np.int64) & 0xFFFF).astype(np.float32).reshape(n, dim)
datagen.write_dataset("$D", g, feat=feat, label=(np.arange(n) % 47).astype(np.int64))
print("dataset written")
PY
make -s -C $GRAFT_REPO_ROOT/xgnn_amd/csrc driver
# arch1 keeps the whole feature table in HBM whatever cache_percentage says (run_config.h:124-126), as the reference does
$GRAFT_REPO_ROOT/build/samgraph_no_train --dataset-path $D --num-epoch 3 --seed 1 "$@" 2>&1 | grep "^\[epoch" || true
