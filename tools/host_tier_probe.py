"""Diagnostic: PCIe ceiling of this box (pinned H2D copy) next to the zero-copy gather of rows from pinned host memory."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xgnn_amd import ops

dev = torch.device("cuda", 0)
for dim in (100, 128, 256):
    n = 2_000_000
    host = torch.empty((n, dim), dtype=torch.float32, pin_memory=True)
    host.fill_(1.0)
    d = torch.empty_like(host, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        d.copy_(host, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"dim {dim}: pinned H2D copy {host.numel() * 4 / dt / 1e9:.1f} GB/s")
    for rows in (200_000, 1_000_000):
        idx = torch.from_numpy(np.random.RandomState(1).randint(0, n, rows).astype(np.int32)).to(dev)
        out = torch.empty((rows, dim), dtype=torch.float32, device=dev)
        ops.gather_scatter(out, host, idx, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ops.gather_scatter(out, host, idx, None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"  zero-copy gather of {rows} random rows: {rows * dim * 4 / dt / 1e9:.1f} GB/s")
        idx2, _ = torch.sort(idx)
        ops.gather_scatter(out, host, idx2, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ops.gather_scatter(out, host, idx2, None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"  same rows, sorted by id: {rows * dim * 4 / dt / 1e9:.1f} GB/s")
