"""Host logic around the multi-GPU run that needs no GPU: bench.py's own launcher (`--gpus N` becomes N ranks or
a non-zero exit, never a line about another GPU count) and the deadlines on every wait for a peer while GGMS
shards are connected (xgnn_amd.ggms_store.with_deadline)."""
import os
import socket
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _json_lines(stdout):
    return [l for l in stdout.splitlines() if l.startswith("{")]


def test_world_size_and_gpus_flag_must_agree():
    """Started by a launcher with WORLD_SIZE != --gpus the script refuses to run (it used to measure WORLD_SIZE
    ranks and could be told anything in --gpus)."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--preset", "tiny"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and not _json_lines(r.stdout)


def test_gpus_flag_launches_the_ranks_itself_and_propagates_their_failure():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks (torch.distributed.run) before touching
    the GPU.  On a box with fewer than two GPUs (this container has none, the test box one) both ranks refuse to
    start and the launcher's non-zero code comes back -- with no JSON line at all, in particular none saying
    n_gpus: 1."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GGMS_BENCH_DEVICE")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--preset", "tiny", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0 and not _json_lines(r.stdout)
    assert "needs 2 GPUs" in r.stderr  # said by the ranks themselves: the launcher really started them


def test_with_deadline_passes_values_and_errors_through():
    from xgnn_amd.ggms_store import with_deadline
    assert with_deadline(lambda: 41 + 1, "sum", seconds=5) == 42
    with pytest.raises(KeyError):
        with_deadline(lambda: {}["x"], "lookup", seconds=5)
    said = []
    assert with_deadline(lambda: time.sleep(30), "sleeper", seconds=0.2, on_timeout=lambda m: said.append(m) or "late") == "late"
    assert "sleeper" in said[0]


def test_with_deadline_ends_the_process_with_a_message():
    code = (f"import sys, time; sys.path.insert(0, {ROOT!r})\n"
            "from xgnn_amd.ggms_store import with_deadline\n"
            "with_deadline(lambda: time.sleep(600), 'rank 0: hipIpcOpenMemHandle of rank 1 (123 bytes)')\n"
            "print('not reached')\n")
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, GGMS_IPC_TIMEOUT_S="1"))
    assert r.returncode == 3 and time.time() - t0 < 30
    assert "hipIpcOpenMemHandle of rank 1 (123 bytes)" in r.stderr and "not reached" not in r.stdout


_PEER_SCRIPT = """
import os, sys, time
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from xgnn_amd import ggms_store

class HostShard:  # what connect_peers needs of ops.SharedShard, on the host
    def __init__(self):
        self.tensor = torch.zeros((8, 4)); self.shape = (8, 4); self.ptr = self.tensor.data_ptr()
    def export_handle(self): return b"h" * 64
    def import_peer(self, h): return 1234

rank = int(sys.argv[1])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2])
dist.init_process_group("gloo", rank=rank, world_size=2)
if rank == 1:
    time.sleep(60)   # never publishes its shard
    os._exit(0)
sh = HostShard()
st = ggms_store.FeatureShards(sh.tensor, None, 2, rank, mode="peer", dist=dist, leaf=object())
st.connect_peers(sh)
print("not reached")
"""


def test_a_rank_that_never_publishes_its_shard_ends_the_others_with_a_message(tmp_path):
    """The exporter never publishes: the importing rank exits non-zero within the deadline and says what it waited
    for, instead of holding the run until the driver's limit."""
    script = tmp_path / "peer.py"
    script.write_text(_PEER_SCRIPT.format(root=ROOT))
    port = str(_free_port())
    env = dict(os.environ, GGMS_IPC_TIMEOUT_S="3")
    stuck = subprocess.Popen([sys.executable, str(script), "1", port], env=env)
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, str(script), "0", port], capture_output=True, text=True, timeout=120, env=env)
    finally:
        stuck.kill()
        stuck.wait()
    assert r.returncode == 3 and time.time() - t0 < 30, (r.returncode, r.stderr[-1000:])
    assert "rank 0 of 2" in r.stderr and "never published its shard" in r.stderr and "not reached" not in r.stdout


_REFUSED_SCRIPT = """
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from xgnn_amd import ggms_store

rank = int(sys.argv[1])

class HostShard:
    def __init__(self):
        self.tensor = torch.zeros((8, 4)); self.shape = (8, 4); self.ptr = self.tensor.data_ptr()
    def export_handle(self):
        if sys.argv[3] == "export" and rank == 1:
            raise RuntimeError("hipIpcGetMemHandle: invalid argument")
        return b"h" * 64
    def import_peer(self, h):
        if sys.argv[3] == "import" and rank == 1:
            raise RuntimeError("hipIpcOpenMemHandle: invalid device pointer")
        self.opened = getattr(self, "opened", 0) + 1
        return 1234
    def release_peers(self):  # the mappings that did open are closed before the verdict is raised
        print("RELEASED", getattr(self, "opened", 0))

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2])
dist.init_process_group("gloo", rank=rank, world_size=2)
sh = HostShard()
st = ggms_store.FeatureShards(sh.tensor, None, 2, rank, mode="peer", dist=dist, leaf=object())
try:
    st.connect_peers(sh)
    print("MAPPED")
except ggms_store.PeerConnectError as e:
    print("REFUSED", e)
dist.barrier()   # both ranks are still in step: the failure did not strand the other one
print("in step")
"""


@pytest.mark.parametrize("where", ["import", "export"])
def test_a_refused_ipc_call_on_one_rank_is_raised_on_every_rank(tmp_path, where):
    """An IPC call that returns an error on ONE rank: every rank gets PeerConnectError with that rank's message, and
    the group stays in step (the caller may then take the same turn everywhere, as bench.py does)."""
    script = tmp_path / "refused.py"
    script.write_text(_REFUSED_SCRIPT.format(root=ROOT))
    port = str(_free_port())
    env = dict(os.environ, GGMS_IPC_TIMEOUT_S="30")
    ps = [subprocess.Popen([sys.executable, str(script), str(r), port, where], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = [p.communicate(timeout=120) for p in ps]
    for p, (out, err) in zip(ps, outs):
        assert p.returncode == 0, err[-1000:]
        assert "REFUSED" in out and "rank 1 of 2" in out and "in step" in out and "MAPPED" not in out
        assert "RELEASED" in out  # nobody is left owning a peer mapping
        assert ("hipIpcOpenMemHandle: invalid device pointer" if where == "import" else "hipIpcGetMemHandle") in out


_OOM_SCRIPT = """
import os, sys, time
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from xgnn_amd import ggms_store
rank = int(sys.argv[1])

class HostShard:
    def __init__(self):
        self.tensor = torch.zeros((8, 4)); self.shape = (8, 4); self.ptr = self.tensor.data_ptr()
    def export_handle(self): return b"h" * 64
    def import_peer(self, h): return 1234
    def release_peers(self): print("RELEASED")

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2])
dist.init_process_group("gloo", rank=rank, world_size=2)
# rank 1 "ran out of memory" building its shard: it still takes part in the exchange, with its reason
sh = ggms_store.FailedShard("OutOfMemoryError: tried to allocate 28.4 GiB") if rank == 1 else HostShard()
try:
    ggms_store.connect_shared(sh, 2, rank, dist, what="feature shard")
    print("MAPPED")
except ggms_store.PeerConnectError as e:
    print("REFUSED", e)
dist.barrier()
print("in step")
"""


def test_a_rank_that_cannot_build_its_shard_is_a_verdict_of_every_rank(tmp_path):
    """One rank fails BEFORE the handle exchange (out of memory while filling its shard): it walks through the same
    collectives with its reason (ggms_store.FailedShard), every rank raises PeerConnectError with that reason, nobody
    waits for a deadline and the group stays in step."""
    script = tmp_path / "oom.py"
    script.write_text(_OOM_SCRIPT.format(root=ROOT))
    port = str(_free_port())
    env = dict(os.environ, GGMS_IPC_TIMEOUT_S="30")
    t0 = time.time()
    ps = [subprocess.Popen([sys.executable, str(script), str(r), port], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = [p.communicate(timeout=120) for p in ps]
    assert time.time() - t0 < 25  # nobody sat out the 30-s deadline
    for p, (out, err) in zip(ps, outs):
        assert p.returncode == 0, err[-1000:]
        assert "REFUSED" in out and "could not build its feature shard" in out and "28.4 GiB" in out
        assert "in step" in out and "MAPPED" not in out


_TOPO_SCRIPT = """
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from xgnn_amd import ggms_store
rank, mode = int(sys.argv[1]), sys.argv[3]

class HostShard:  # what TopologyShards needs of ops.SharedShard, on the host
    made = 0
    def __init__(self, shape, dtype, device):
        HostShard.made += 1
        if mode == "alloc" and rank == 1 and HostShard.made == 2:
            raise RuntimeError("ggms_device_alloc failed (status -2): out of memory")
        self.tensor = torch.zeros(shape, dtype=dtype); self.shape = tuple(shape); self.ptr = self.tensor.data_ptr()
        self.no = HostShard.made
    def export_handle(self): return b"h" * 64
    def import_peer(self, h):
        if mode == "import2" and rank == 1 and self.no == 2:
            raise RuntimeError("hipIpcOpenMemHandle: invalid device pointer")
        return 1234
    def release_peers(self): print("RELEASED", self.no)
    def close(self): print("CLOSED", self.no)

def build(indptr, indices, world, ncn, only=None):
    if mode == "build" and rank == 1:
        raise MemoryError("tried to allocate 13.2 GiB")
    return ggms_store.topology_shards(indptr, indices, world, ncn, only=only)

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2])
dist.init_process_group("gloo", rank=rank, world_size=2)
ip = torch.tensor([0, 2, 3, 5, 6], dtype=torch.int32)
ix = torch.tensor([1, 2, 0, 3, 0, 2], dtype=torch.int32)
try:
    t = ggms_store.TopologyShards(ip, ix, 2, rank, 4, dist, (ip, ix), shard_alloc=HostShard, build=build)
    print("MAPPED", t.graph.c.num_part, t.graph.c.num_cache_node)
except ggms_store.PeerConnectError as e:
    print("REFUSED", e)
dist.barrier()   # the group is still in step
print("in step")
"""


@pytest.mark.parametrize("mode", ["build", "alloc", "import2", "fine"])
def test_topology_shards_failures_are_a_verdict_of_every_rank(tmp_path, mode):
    """a rank whose topology shard cannot be built ('build': out of memory in topology_shards) or
    allocated ('alloc': the second SharedShard) never used to enter the handle exchange, and its peers sat out the
    deadline in all_gather_object.  Now it walks through the exchange as a FailedShard: both ranks raise
    PeerConnectError with its reason and stay in step; what was allocated or mapped before is released ('import2': the
    SECOND exchange fails -- the indptr holder and its mappings are closed too)."""
    script = tmp_path / "topo.py"
    script.write_text(_TOPO_SCRIPT.format(root=ROOT))
    port = str(_free_port())
    env = dict(os.environ, GGMS_IPC_TIMEOUT_S="30")
    t0 = time.time()
    ps = [subprocess.Popen([sys.executable, str(script), str(r), port, mode], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = [p.communicate(timeout=120) for p in ps]
    assert time.time() - t0 < 25  # nobody sat out the 30-s deadline
    for r, (p, (out, err)) in enumerate(zip(ps, outs)):
        assert p.returncode == 0, err[-1000:]
        assert "in step" in out
        if mode == "fine":
            assert "MAPPED 2 4" in out and "REFUSED" not in out
            continue
        assert "REFUSED" in out and "MAPPED" not in out and "rank 1 of 2" in out
        if mode == "build":
            assert "could not build its topology indptr shard" in out and "13.2 GiB" in out
            assert "CLOSED 1" in out and "CLOSED 2" in out if r == 0 else "CLOSED" not in out  # rank 1 never allocated
        if mode == "alloc":
            assert "could not build its topology indptr shard" in out and "out of memory" in out
            assert "CLOSED 1" in out  # both ranks drop what they did allocate
        if mode == "import2":
            assert "topology indices shard" in out and "invalid device pointer" in out
            assert "CLOSED 1" in out and "CLOSED 2" in out  # the first exchange's holder goes too


def test_more_shards_than_the_kernels_carry_is_refused_before_anything_is_built():
    """a group larger than GGMS_MAX_PARTS used to exchange handles and map peers before its first
    gather returned GGMS_ERR_INVALID; the stores now refuse it up front."""
    import torch
    from xgnn_amd import ggms_store, ops
    with pytest.raises(ValueError, match="GGMS_MAX_PARTS"):
        ggms_store.FeatureShards(torch.zeros((4, 4)), None, 9, 0, mode="peer", leaf=object())
    ggms_store.FeatureShards(torch.zeros((4, 4)), None, 9, 0, mode="a2a", leaf=object())  # no pointers by value there
    with pytest.raises(ValueError, match="GGMS_MAX_PARTS"):
        ggms_store.TopologyShards(None, None, 9, 0, 0, None, (None, None))
    with pytest.raises(ValueError, match="GGMS_MAX_PARTS"):
        ops.PartTable(list(range(10)))
    ops.PartTable(list(range(9)))  # topology: 8 shards + the host slot


_PROBE_SCRIPT = """
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from xgnn_amd import ggms_store
rank, mode = int(sys.argv[1]), sys.argv[3]

class Holder:
    def __init__(self, fill): self.tensor = torch.full((8, 4), fill, dtype=torch.int32); self.shape = (8, 4); self.ptr = 1000 + fill
    def export_handle(self): return bytes([self.ptr - 1000]) * 64
    def import_peer(self, h): return 1000 + h[0] if mode != "crossed" else 1000 + rank + 1   # 'crossed': maps its OWN buffer
    def release_peers(self): pass
    def close(self): print("CLOSED")

class Leaf:  # what link_probe / peer_access_preflight need of the device, on the host
    def peer_access(self, dev, peer): return 0 if (mode == "refused" and dev == 1 and peer == 0) else 1
    def shard(self, rows, words, fill): return Holder(fill)
    def scratch(self, nbytes): return [0]
    def copy_rate(self, dst, src_ptr, nbytes, reps, with_kernel=0): dst[0] = src_ptr - 1000; return 100.0 * rank + (src_ptr - 1000)
    def gather_rate(self, out, ptrs, rpp, rb, n, seed, reps, ws): return 10.0 * rank + sum(p - 1000 for p in ptrs)
    def first_word(self, t): return t[0]

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2])
dist.init_process_group("gloo", rank=rank, world_size=2)
pre = ggms_store.peer_access_preflight(2, rank, dist, rank, Leaf())
print("PRE", pre)
try:
    rec = ggms_store.link_probe(2, rank, dist, Leaf(), probe_bytes=4096, row_bytes=512, reps=1)
    print("REC", rec["per_pair_copy_GBps"], rec["per_pair_gather_GBps"], rec["inbound_all_peers_gather_GBps"],
          rec["inbound_all_peers_min_GBps"], rec["per_pair_gather_min_GBps"])
except ggms_store.PeerConnectError as e:
    print("REFUSED", e)
dist.barrier()
print("in step")
"""


@pytest.mark.parametrize("mode", ["fine", "refused", "crossed"])
def test_link_probe_plumbing_two_ranks(tmp_path, mode):
    """The xGMI probe's host logic on two gloo ranks with a stand-in device: the preflight matrix is the same on both
    ranks and names the refused pair; the probe's matrices are [reader][owner] with every rank's row in place; a mapping
    that does not show its owner's memory is a PeerConnectError on EVERY rank; the probe buffer is released either way."""
    script = tmp_path / "probe.py"
    script.write_text(_PROBE_SCRIPT.format(root=ROOT))
    port = str(_free_port())
    env = dict(os.environ, GGMS_IPC_TIMEOUT_S="30")
    ps = [subprocess.Popen([sys.executable, str(script), str(r), port, mode], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = [p.communicate(timeout=120) for p in ps]
    for p, (out, err) in zip(ps, outs):
        assert p.returncode == 0, err[-1000:]
        assert "in step" in out and "CLOSED" in out
        if mode == "refused":
            assert "'refused': [[1, 0]]" in out and "'devices': [0, 1]" in out
        else:
            assert "'refused': []" in out
        if mode == "crossed":
            assert "REFUSED" in out and "do not show their owner's memory" in out and "[0, 1]" in out and "[1, 0]" in out
        else:
            # copy: 100 * reader + (owner + 1); gather: 10 * reader + (owner + 1); inbound: 10 * reader + (peer + 1)
            assert "REC [[1.0, 2.0], [101.0, 102.0]] [[1.0, 2.0], [11.0, 12.0]] [2.0, 11.0] 2.0 2.0" in out


def test_streams_trial_verdict_and_slot_counts():
    """bench.py's host logic around the streams trial (no GPU): the baseline stays unless another candidate is at least
    2 % faster; batch slots = batches in flight + one spare, a multiple of the sampling pipelines."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)  # (imports numpy only: torch is imported inside main())
    assert bench.choose_streams({(1, 1): 0.700, (1, 2): 0.690, (2, 1): 0.710}) == (1, 1)   # 1.4 %: not enough
    assert bench.choose_streams({(1, 1): 0.700, (1, 2): 0.650, (2, 1): 0.640}) == (2, 1)
    assert bench.choose_streams({(1, 1): 0.300, (1, 2): 0.301, (2, 1): 0.320}) == (1, 1)
    assert bench.choose_streams({(2, 1): 0.380, (2, 2): 0.369}) == (2, 2)
    assert [bench.slots_for(*kx) for kx in ((1, 1), (1, 2), (2, 1), (2, 2))] == [3, 4, 4, 6]
    for k in (1, 2, 3):
        for x in (1, 2):
            n = bench.slots_for(k, x)
            assert n >= k + x + 1 and n % k == 0
