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
