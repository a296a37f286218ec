"""Diagnostic micro-benchmark of the row gather over row widths and dtypes (not part of the product).
Prints, per case: the gather, torch.index_select and a straight copy of the same bytes, as algorithmic TB/s
(4-B index + row read + row write) and as a fraction of the 8 TB/s HBM peak."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xgnn_amd import ops

dev = torch.device("cuda", 0)
N, n = 2_449_029, 1_280_000


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


idx = torch.randperm(N, device=dev)[:n].to(torch.int32)
idx64 = idx.long()
print("| row | bytes | gather us | gather TB/s (frac of 8) | index_select TB/s | straight copy TB/s |")
print("|---|---|---|---|---|---|")
for name, dtype, dim in [("f32 x 100", torch.float32, 100), ("f32 x 128", torch.float32, 128),
                         ("f32 x 256", torch.float32, 256), ("f32 x 602", torch.float32, 602),
                         ("f16 x 100", torch.float16, 100), ("f16 x 128", torch.float16, 128),
                         ("u8 x 100", torch.uint8, 100), ("i64 x 1", torch.int64, 1), ("f32 x 16", torch.float32, 16)]:
    if dtype.is_floating_point:
        feat = torch.randn(N, dim, device=dev).to(dtype)
    else:
        feat = torch.randint(0, 100, (N, dim), device=dev, dtype=dtype)
    out = torch.empty((n, dim), dtype=dtype, device=dev)
    rb = dim * feat.element_size()
    by = n * (4 + 2 * rb)
    t = timeit(lambda: ops.extract(feat, idx, out=out))
    t_sel = timeit(lambda: torch.index_select(feat, 0, idx64, out=out))
    src = feat[:n]
    t_cp = timeit(lambda: out.copy_(src))
    print(f"| {name} | {rb} | {t * 1e6:.1f} | {by / t / 1e12:.2f} ({by / t / 8e12:.2f}) | {by / t_sel / 1e12:.2f} | "
          f"{2 * n * rb / t_cp / 1e12:.2f} |", flush=True)
    del feat, out
