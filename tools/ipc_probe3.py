"""ipc_probe2 under torchrun (env:// rendezvous), as bench.py is launched, with optional bench-like preludes:
    GGMS_BENCH_DEVICE=0 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 \
        tools/ipc_probe3.py <MiB> [h2d_GiB] [arange_M]"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xgnn_amd import ops  # noqa: E402

mib = int(sys.argv[1])
h2d_gib = float(sys.argv[2]) if len(sys.argv) > 2 else 0
arange_m = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
keep = []
if h2d_gib:  # a pageable host array copied to the device, like the graph upload
    a = np.random.RandomState(rank).randint(0, 1 << 30, size=int(h2d_gib * (1 << 28)), dtype=np.int64).astype(np.int32)
    keep.append(torch.from_numpy(a).to("cuda"))
    keep.append(a)
if arange_m:
    keep.append(torch.arange(arange_m * 1_000_000, dtype=torch.int64, device="cuda") % 172)
torch.cuda.synchronize()
print(f"rank {rank}: prelude done", flush=True)
sh = ops.SharedShard((mib * (1 << 18) // 128, 128), torch.float32, torch.device("cuda", 0))
sh.tensor[:8] = float(rank + 1)
torch.cuda.synchronize()
handles = [None, None]
dist.all_gather_object(handles, sh.export_handle())
t0 = time.perf_counter()
p = sh.import_peer(handles[1 - rank])
t1 = time.perf_counter()
v = torch.as_tensor(ops._RawDevice(p, (1024,), "<f4"), device="cuda:0")
print(f"rank {rank}: import of {mib} MiB took {t1 - t0:.3f} s, peer value {float(v[0].item())}", flush=True)
dist.barrier()
dist.destroy_process_group()
