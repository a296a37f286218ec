"""Diagnostic micro-benchmark of the row gather (not part of the product)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xgnn_amd import ops

dev = torch.device("cuda", 0)
N = 2_449_029
for dim in (100, 128, 256):
    feat = torch.randn(N, dim, device=dev)
    n = 1_280_000
    idx = torch.randperm(N, device=dev)[:n].to(torch.int32)
    out = torch.empty(n, dim, device=dev)
    def timeit(fn, reps=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    t = timeit(lambda: ops.extract(feat, idx, out=out))
    by = n * (4 + 2 * dim * 4)
    print(f"dim {dim}: ggms_extract {t*1e6:8.1f} us  {by/t/1e12:.3f} TB/s algorithmic (NT={os.environ.get('GGMS_EXTRACT_NT','0')})")
    idx64 = idx.long()
    t = timeit(lambda: torch.index_select(feat, 0, idx64, out=out))
    print(f"dim {dim}: torch.index_select {t*1e6:8.1f} us  {by/t/1e12:.3f} TB/s")
    src = feat[:n]
    t = timeit(lambda: out.copy_(src))
    print(f"dim {dim}: straight copy  {t*1e6:8.1f} us  {2*n*dim*4/t/1e12:.3f} TB/s")
    sidx = torch.sort(idx)[0]
    t = timeit(lambda: ops.extract(feat, sidx, out=out))
    print(f"dim {dim}: ggms_extract sorted idx {t*1e6:8.1f} us  {by/t/1e12:.3f} TB/s")
    del feat, out
