"""Diagnostic: host enqueue time vs GPU time of one mini-batch (not part of the product)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xgnn_amd import datagen, ops

dev = torch.device("cuda", 0)
g = datagen.make_graph("products", seed=42)
to_dev = lambda a: torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(dev)
graph = ops.DeviceGraph(to_dev(g["indptr"]), to_dev(g["indices"]))
bs = ops.BatchSampler(graph, [25, 10], 8000, seed=1, device=dev)
seeds = [to_dev(g["train_set"][i * 8000:(i + 1) * 8000].copy()) for i in range(20)]
for s in seeds[:5]:
    bs.sample(s)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for s in seeds:
        bs.sample(s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"20 batches: host enqueue {1e3*(t1-t0)/20:.3f} ms/batch, total {1e3*(t2-t0)/20:.3f} ms/batch")
# single kernels, host cost
x = torch.zeros(1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(1000):
    bs.ht.reset()
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"ht.reset (memsetAsync 4B): {1e6*(t1-t0)/1000:.1f} us host each")
idx = seeds[0]
lab = torch.arange(2449029, dtype=torch.int64, device=dev)
o = torch.empty(8000, dtype=torch.int64, device=dev)
t0 = time.perf_counter()
for _ in range(1000):
    ops.extract(lab, idx, out=o)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"extract label launch: {1e6*(t1-t0)/1000:.1f} us host each, {1e6*(t2-t0)/1000:.1f} us incl gpu")
