"""Link / topology probe (include/ggms.h, PartitionSolver::DetectTopo of cuda/dist_graph.cu:684-938) on the one-GPU box:
the all-devices probe and its file, a rank's own measurements on buffers it holds, the engine's forked probe child."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_detect_topology_and_file_round_trip(tmp_path):
    """In a child process (the probe creates a context on every device): the P2P matrix has a 1 on the diagonal, the
    diagonal rate is a local copy counted read + write as the reference counts it, and the file loads back."""
    code = f"""
import sys, ctypes as C; sys.path.insert(0, {ROOT!r})
from xgnn_amd import _lib
h = _lib.lib(); t = _lib.Topology()
assert h.ggms_detect_topology(C.byref(t), 64 << 20, 2) == 0, h.ggms_last_error()
n = t.num_device
assert n >= 1 and all(t.can_access[i][i] == 1 for i in range(n)) and all(t.copy_GBps[i][i] > 200 for i in range(n))
p = {str(tmp_path / '.detect_topo_test')!r}.encode()
assert h.ggms_topology_write_host(C.byref(t), p, b"test") == 0
u = _lib.Topology()
assert h.ggms_topology_read_host(C.byref(u), p) == 0 and u.num_device == n
assert abs(u.copy_GBps[0][0] - t.copy_GBps[0][0]) < 0.01 and u.can_access[0][0] == 1
print("topo-ok", n, round(t.copy_GBps[0][0]))
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "topo-ok" in r.stdout, r.stderr[-2000:]
    text = open(tmp_path / ".detect_topo_test").read()
    assert text.startswith("GPU Count ") and "P2P Matrix" in text and "Bandwidth Matrix" in text


def test_link_probe_on_local_buffers():
    """ggms_link_probe_copy / _gather with local HBM standing in for the peers: rates are positive, the gather really
    gathered (row i of the output is the row its generated index names, across two parts)."""
    import torch
    from xgnn_amd import _lib, ops
    h = _lib.lib()
    dev = torch.device("cuda", 0)
    rows_per_part, row_bytes, num_rows = 1 << 16, 512, 1 << 17
    parts = [(torch.arange(rows_per_part * row_bytes // 4, dtype=torch.int32, device=dev) + (p << 28)).reshape(rows_per_part, -1)
             for p in range(2)]
    out = torch.zeros((num_rows, row_bytes // 4), dtype=torch.int32, device=dev)
    index = torch.zeros(num_rows, dtype=torch.int32, device=dev)
    tab = ops.PartTable([p.data_ptr() for p in parts])
    rate = C.c_double(0)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert h.ggms_link_probe_gather(C.c_void_p(out.data_ptr()), tab.ptr(), 2, rows_per_part, row_bytes, num_rows, 7, 2,
                                    C.c_void_p(index.data_ptr()), C.byref(rate), s) == 0, h.ggms_last_error()
    assert rate.value > 50
    idx = index.cpu().numpy().view(np.uint32)
    assert idx.max() < 2 * rows_per_part and np.unique(idx).size > num_rows // 4  # spread over both parts
    got = out.cpu().numpy()
    for i in range(0, num_rows, 997):
        part, row = int(idx[i]) % 2, int(idx[i]) // 2
        assert got[i, 0] == (part << 28) + row * (row_bytes // 4) and got[i, -1] == got[i, 0] + row_bytes // 4 - 1
    dst = torch.empty(rows_per_part * row_bytes, dtype=torch.uint8, device=dev)
    for with_kernel in (0, 1):
        dst.zero_()
        assert h.ggms_link_probe_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(parts[1].data_ptr()), dst.numel(), 2, with_kernel,
                                      C.byref(rate), s) == 0
        assert rate.value > 50 and torch.equal(dst.view(torch.int32).reshape(rows_per_part, -1), parts[1])
    can = C.c_int(0)
    assert h.ggms_peer_access(0, 0, C.byref(can)) == 0 and can.value == 1


def test_engine_probes_the_topology_before_it_places_shards(tmp_path):
    """arch6 with two workers on a box that shows ONE GPU (no SAMGRAPH_FORCE_DEVICE rehearsal hook): data_init forks the
    probe child, reads its file back and refuses the deployment with the GPU count named -- before any worker is
    forked or any shard built.  The file the child wrote is a valid topology file."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_engine import make_dataset
    d = make_dataset(tmp_path / "ds")
    topo = str(tmp_path / ".detect_topo_engine")
    env = {k: v for k, v in os.environ.items() if k != "SAMGRAPH_FORCE_DEVICE"}
    env.update(SAMGRAPH_TOPO_FILE=topo, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "engine_driver.py"), d["path"], str(tmp_path / "out"),
                        "arch6", "2", "cache_percentage=0.4", "part_cache=True", "gpu_extract=True"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "arch6 with 2 workers, but the node shows 1 GPUs" in r.stderr, r.stderr[-2000:]
    text = open(topo).read()
    assert text.startswith("GPU Count 1") and "Bandwidth Matrix" in text
