#!/bin/bash
# Engine-level throughput at the headline size (run on the GPU box): the C++ driver over the samgraph_* ABI, arch1,
# khop3 [5,10,15], batch 8000, on a papers100M-shaped dataset written in the reference's on-disk format WITHOUT feat.bin /
# label.bin (the loader then maps zero-filled tables, engine.cc:199-235) -- topology and sizes are the real ones.
set -e
D=/tmp/papers_ds
python - <<PY
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from xgnn_amd import datagen
g = datagen.make_graph("papers100M", seed=42)
datagen.write_dataset("$D", g)
print("dataset written", g["meta"])
PY
make -s -C $GRAFT_REPO_ROOT/xgnn_amd/csrc driver
$GRAFT_REPO_ROOT/build/samgraph_no_train --dataset-path $D --num-epoch 2 --seed 1 --fanout "5 10 15" "$@" 2>&1 | grep "^\[epoch\|FATAL\|Check" || true
