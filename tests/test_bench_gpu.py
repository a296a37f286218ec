"""bench.py contract on the GPU box: one JSON line with the required keys; the N = 2 launch (two ranks pinned to
the box's one GPU through the bench's test hooks, gloo for the timing barrier) for every feature-store mode."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json(stdout):
    return json.loads([l for l in stdout.strip().splitlines() if l.startswith("{")][-1])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--preset", "tiny", "--steps", "4", "--warmup", "1",
                        "--batch", "512", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d) and "cpu_baseline" in d and "host_tier" in d
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["bound"] == "hbm" and d["config"]["workload"]
    assert d["rows_verified"] and d["repeats"]["blocks"] == 3 and 0 < d["roofline"]["hbm_read_frac"] < 0.5
    # one or two sampling pipelines: tried before the timed region, the line says what was chosen and why
    pt = d["config"]["pipelines_trial"]
    ch = pt["chosen"]
    assert ch["pipelines"] in (1, 2) and ch["extract_streams"] in (1, 2) and len(pt["ms_per_step"]) == 3
    assert all(v > 0 for v in pt["ms_per_step"].values())
    assert d["config"]["streams"].startswith(f"{ch['pipelines']} sampling pipeline")
    rf = d["roofline"]
    assert rf["extract_streams"] == ch["extract_streams"] and 1.0 <= rf["launches_in_flight"] < 2.0
    # achieved = algorithmic bytes per launch x launches in flight / a launch's own duration
    assert abs(rf["frac"] - rf["frac_of_one_launch"] * rf["launches_in_flight"]) < 1e-9
    assert d["host_tier"]["feature_extract_GBps"] > 0 and d["host_tier"]["pinned_h2d_copy_GBps"] > 0
    cb = d["cpu_baseline"]
    assert cb["cores"] >= 1 and cb["value"] > 0 and set(cb["seconds"]) == {"sample", "remap", "extract", "total"}
    # the sampler's own bound: algorithmic bytes vs 8 TB/s, and its requests against ceilings probed in this process
    rs = d["roofline_sampler"]
    assert rs["bound"] == "hbm" and 0 < rs["frac"] < 1 and rs["alone_ms"] > 0
    ms = rs["memory_side"]
    assert ms["atomics_per_s_ceiling"] > 1e9 and ms["random_loads_per_s_ceiling"] > 1e9 and ms["chain_over_floor"] > 0.5
    # the same workload through the samgraph.torch surface, child process
    en = d["engine"]
    assert "error" not in en, en
    assert en["ms_per_step"] > 0 and en["edges_per_s"] > 0 and en["feature_GBps"] > 0 and en["steps"] == 8
    # two extract streams: the epoch's copy time counts the streams' BUSY time once, not every gather's own duration
    assert 0 < en["log_items"]["kLogEpochCopyTime"] <= 1.05 * en["wall_s"], en


def test_single_gpu_line_carries_its_box_the_other_configs_and_the_staged_host_tier():
    """the N = 1 line carries `box` (what this box sustains: the yardstick between boxes) and
    `value_over_box`, the north_star read-fraction target next to the ceiling a copy can reach, a `configs` sub-record
    (the other single-GPU BASELINE configurations, one child each -- here on the tiny preset through the test hook) and
    `host_tier.staged` (the engine's host-staged path, cache 0 and 0.64: per-phase rates of the serial sequence, effective
    rate of the chunked pipeline)."""
    env = dict(os.environ, GGMS_BENCH_TEST_CONFIGS="tiny")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--preset", "tiny", "--steps", "4", "--warmup", "1",
                        "--batch", "512", "--no-cpu-baseline", "--no-engine"], capture_output=True, text=True, timeout=900,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    box = d["box"]
    assert box["d2d_copy_GBps"] > 500 and box["pinned_h2d_GBps"] > 1 and box["atomics_per_s"] > 1e9 and box["loads_per_s"] > 1e9
    assert abs(d["value_over_box"] - d["value"] / box["d2d_over_guide"]) < 1e-6 * d["value_over_box"]
    rf = d["roofline"]
    assert rf["hbm_read_frac_target"] == 0.60 and 0.2 < rf["hbm_read_frac_ceiling_for_a_copy"] < 0.5
    cf = d["configs"]
    assert set(cf) == {"products_graphsage_25_10", "friendster_pinsage_rw", "papers100M_graphsage_25_10"}
    for name, c in cf.items():
        assert "error" not in c, (name, c)
        assert c["ms_per_step"] > 0 and c["edges_per_s"] > 0 and c["feature_extract_GBps"] > 0 and c["rows_verified"]
        assert 0 < c["gather_frac"] < 1
    assert "random_walk" in cf["friendster_pinsage_rw"]["workload"]
    # configs[1] also through the samgraph.torch surface (the child's own engine record)
    en = cf["products_graphsage_25_10"]["engine"]
    assert "error" not in en, en
    assert en["ms_per_step"] > 0 and en["edges_per_s"] > 0
    st = d["host_tier"]["staged"]
    assert st["reference_published"]["h2d_GBps"] == 23.33
    for name in ("cache_0", "cache_0.64"):
        c = st[name]
        assert "error" not in c, (name, c)
        sr, ov = c["serial"], c["overlapped"]
        assert sr["cpu_gather_GBps"] > 0 and sr["h2d_GBps"] > 0 and sr["effective_GBps"] > 0
        assert (sr["combine_miss_GBps"] is None) == (name == "cache_0")  # no cache: rows land in the batch, nothing to scatter
        assert ov["effective_GBps"] > 0 and ov["over_min_of_cpu_gather_and_h2d"] > 0 and ov["edges_per_s"] > 0


def test_headline_is_out_before_the_sub_records_and_the_budget_bounds_them():
    """The headline line is out as soon as the main region is measured (stdout at N > 1, stderr at N = 1, where stdout keeps
    the contract's single line); the optional sub-records share one wall-clock budget.  Here the engine child never
    finishes (test hook) and the budget is short: the run still ends with rc 0 inside the budget (+ the interpreter's
    start-up), the early line is a complete headline, the final one carries `engine.error`, and what no longer fitted
    says `skipped: budget`."""
    import time
    env = dict(os.environ, GGMS_BENCH_TEST_ENGINE_SLEEP="600")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--preset", "tiny", "--steps", "4", "--warmup", "1",
                        "--batch", "512", "--cpu-seconds", "20", "--budget-s", "45"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=env)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.strip().splitlines() if l.startswith("{")]
    early = [json.loads(l) for l in r.stderr.strip().splitlines() if l.startswith("{")]
    # N = 1: ONE line on stdout (the contract), the early headline on stderr; N > 1 prints both on stdout, last line wins
    assert len(lines) == 1 and len(early) == 1 and wall < 45 + 30, (len(lines), len(early), wall)
    first, last = early[0], lines[0]
    assert KEYS <= set(first) and first["rows_verified"] and "engine" not in first and first["value"] == last["value"]
    assert first["budget"]["headline_at_s"] < 45
    assert "error" in last["engine"] and "budget" in last["engine"]["error"]
    # the child took what the budget left (minus a margin to finish in): what no longer fits after it says so
    assert last["cpu_baseline"]["skipped"] == "budget" and last["cpu_baseline"]["needed_s"] > last["cpu_baseline"]["left_s"]
    assert last["budget"]["finished_at_s"] < 45 + 10


def test_sampling_through_logical_topology_shards_one_gpu():
    """--dist-graph 0.5 --topology-shards 2 (N = 1): the timed sampler runs through DeviceDistGraph -- two logical
    shards in HBM, the other half of the nodes read from the whole CSR in registered host memory; the rows gathered
    for the sampled input nodes still check against the generator, and the line says what it measured."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--preset", "tiny", "--steps", "4", "--warmup", "1",
                        "--batch", "512", "--no-cpu-baseline", "--no-host-tier", "--no-engine", "--dist-graph", "0.5",
                        "--topology-shards", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    topo = d["config"]["topology"]
    assert d["rows_verified"] and topo["logical_shards"] == 2 and 0.3 < topo["cached_node_fraction"] < 0.7
    assert "registered host memory" in d["config"]["workload"] and d["roofline_sampler"]["alone_ms"] > 0


def test_two_ranks_xgnn_mode_topology_shards_in_the_main_region():
    """--gpus 2 --dist-graph 0.5: rank r keeps topology shard r in its HBM, maps the peer's with hipIpc, the other half of
    the nodes read their lists from registered host memory -- and the batches are the same batches: edges and rows per
    step equal the run on the whole CSR (same seeds, same generator pool), the gathered rows check out."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(GGMS_BENCH_DEVICE="0", GGMS_BENCH_BACKEND="gloo")
    lines = []
    for extra in ([], ["--dist-graph", "0.5"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--preset", "tiny", "--steps", "4",
                            "--warmup", "1", "--batch", "512", "--other-stores", "", "--no-engine", "--no-sampler-roofline",
                            "--repeats", "1", "--no-xgnn-mode"]  # one block: the line reports exactly these batches
                           + extra, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        lines.append(_last_json(r.stdout))
    plain, xgnn = lines
    assert xgnn["n_gpus"] == 2 and xgnn["rows_verified"] and xgnn["config"]["topology"]["shards"] == 2
    assert 0.3 < xgnn["config"]["topology"]["cached_node_fraction"] < 0.7 and "over xGMI" in xgnn["config"]["workload"]
    assert xgnn["per_gpu"]["edges_per_step"] == plain["per_gpu"]["edges_per_step"]
    assert xgnn["per_gpu"]["rows_per_step"] == plain["per_gpu"]["rows_per_step"]


def test_two_ranks_default_is_the_planned_placement():
    """--gpus 2 with no --store: the main region runs on the GGMS placement planned from the per-GPU HBM budget (hybrid:
    hot prefix on every GPU, the rest sharded, one gather kernel over replica / local shard / peer shard); the pure
    shards (peer) and the replicas are measured beside it."""
    env = dict(os.environ, GGMS_BENCH_DEVICE="0", GGMS_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--preset", "tiny", "--steps", "4", "--warmup", "1", "--batch", "512"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 2  # N > 1: headline first, enriched line last
    assert KEYS <= set(d) and d["n_gpus"] == 2 and "feature store: hybrid" in d["config"]["workload"]
    assert "48-GB per-GPU budget" in d["config"]["workload"]
    st = d["stores"]
    assert set(st) == {"peer", "replica", "hybrid"}
    assert 0.3 < st["peer"]["remote_row_fraction"] < 0.7 and st["peer"]["xgmi_bytes_per_step"] > 0
    assert st["replica"]["remote_row_fraction"] == 0 and st["replica"]["xgmi_bytes_per_step"] == 0
    # default plan: the largest hot prefix that fits the per-GPU budget is replicated (at this size: all but the tail)
    assert 0 <= st["hybrid"]["remote_row_fraction"] < st["peer"]["remote_row_fraction"]
    assert st["hybrid"]["replicated_fraction"] > 0.9 and st["hybrid"]["hbm_budget_gb"] == 48.0
    assert st["hybrid"]["edges_per_s"] == d["value"]
    # beside it, by default: the same store with the topology sharded over the GPUs too (XGNN mode)
    x = d["xgnn_mode"]
    assert "error" not in x, x
    assert x["use_dist_graph"] == 1.0 and x["edges_per_s"] > 0 and 0.2 < x["vs_main_edges_per_s"] < 2.0
    # the line explains its own xGMI side.  Preflight (hipDeviceCanAccessPeer per pair, before anything
    # was placed) and the link probe (before the placement was planned: it sizes the sharded tail) in the early headline
    # already; next to every sharded store the remote time its xgmi bytes predict.  (Two ranks on ONE GPU: the rates are HBM rates, the plumbing is what is tested.)
    early = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    for line in (early, d):
        pa = line["peer_access"]
        assert pa["devices"] == [0, 0] and pa["can_access"] == [[1, 1], [1, 1]] and pa["refused"] == []
        assert line["box"]["d2d_copy_GBps"] > 500 and line["value_over_box"] > 0
    assert "error" not in early["xgmi"]  # measured before the store was placed: the early headline carries it already
    xg = d["xgmi"]
    assert "error" not in xg, xg
    for key in ("per_pair_copy_GBps", "per_pair_stream_kernel_GBps", "per_pair_gather_GBps"):
        m = xg[key]
        assert len(m) == 2 and all(len(row) == 2 and all(v > 1.0 for v in row) for row in m), (key, m)
    assert len(xg["inbound_all_peers_gather_GBps"]) == 2 and xg["inbound_all_peers_min_GBps"] > 1.0
    assert xg["probe_bytes"] == 128 << 20 and xg["row_bytes"] == 512 and xg["seconds"] < 30
    for kind in ("peer", "hybrid"):
        rec = st[kind]
        assert rec["xgmi_bytes_per_step_per_gpu"] == rec["xgmi_bytes_per_step"] / 2
        want = rec["xgmi_bytes_per_step_per_gpu"] / (xg["inbound_all_peers_min_GBps"] * 1e9) * 1e3
        assert abs(rec["predicted_remote_ms_per_step"] - want) <= 1e-9 + 1e-6 * want
    assert st["replica"]["predicted_remote_ms_per_step"] == 0


_BROKEN_IPC = """
import runpy, sys
sys.path.insert(0, {root!r})
from xgnn_amd import ops
def refuse(self, handle):
    raise RuntimeError("hipIpcOpenMemHandle: invalid argument (test)")
ops.SharedShard.import_peer = refuse
sys.argv = [{bench!r}] + sys.argv[1:]
runpy.run_path({bench!r}, run_name="__main__")
"""


def test_two_ranks_whose_shards_cannot_be_mapped_report_replicas_and_say_so(tmp_path):
    """hipIpcOpenMemHandle returns an error on every rank: all ranks learn of it together, the line is measured on
    whole-table replicas and carries what was asked for and why it was not measured."""
    bench = os.path.join(ROOT, "bench.py")
    script = tmp_path / "broken_ipc_bench.py"
    script.write_text(_BROKEN_IPC.format(root=os.path.abspath(ROOT), bench=os.path.abspath(bench)))
    env = dict(os.environ, GGMS_BENCH_DEVICE="0", GGMS_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script),
                        "--gpus", "2", "--preset", "tiny", "--steps", "3", "--warmup", "1", "--batch", "512", "--no-engine"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and "feature store: replica" in d["config"]["workload"]
    assert d["config"]["store_requested"] == "hybrid" and "invalid argument (test)" in d["config"]["store_error"]
    assert "rank 0 of 2" in d["config"]["store_error"] and "rank 1 of 2" in d["config"]["store_error"]
    assert "error" in d["stores"]["peer"] and d["stores"]["replica"]["edges_per_s"] > 0
    assert "cannot be built" in r.stderr


def test_gpus_flag_alone_starts_the_ranks():
    """The bare command `python bench.py --gpus 2` (no launcher around it, as the driver runs N = 1): the script
    starts its two ranks itself and the line says n_gpus: 2 -- here both pinned to the box's one GPU by the hooks."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(GGMS_BENCH_DEVICE="0", GGMS_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--preset", "tiny", "--steps", "4",
                        "--warmup", "1", "--batch", "512", "--other-stores", ""],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["rows_verified"]
    # N > 1: the engine sub-record is the multi-GPU deployment -- a child process that forks one arch6 worker per GPU
    en = d["engine"]
    assert "error" not in en, en
    assert en["arch"] == "arch6" and en["workers"] == 2 and en["edges_per_s"] > 0 and en["feature_GBps"] > 0
    assert en["replicate_percentage"] > 0.9  # the main region's placement (at this size the plan replicates all but the tail)


def test_five_ranks_one_gpu_peer_and_hybrid():
    """Rehearsal of the 8-GPU launch as far as a one-GPU box allows (its process guard admits six processes on the
    card, and the test runner is one of them): five ranks, five-way GGMS shards, every rank mapping four peers
    through hipIpc; main region on the sharded store, the hybrid one (hot prefix replicated) measured beside it."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(GGMS_BENCH_DEVICE="0", GGMS_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "5", "--preset", "tiny", "--steps", "3",
                        "--warmup", "1", "--batch", "128", "--repeats", "1", "--store", "peer", "--other-stores", "hybrid",
                        "--replicate-frac", "0.25", "--no-engine"],
                       capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 5 and d["rows_verified"] and set(d["stores"]) == {"peer", "hybrid"}
    assert 0.7 < d["stores"]["peer"]["remote_row_fraction"] < 0.9  # 4/5 of the rows live on a peer
    assert 0 < d["stores"]["hybrid"]["remote_row_fraction"] < d["stores"]["peer"]["remote_row_fraction"]


_NCCL_SCRIPT = """
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle
from xgnn_amd import ggms_store
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1], HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
# the collectives of the exchange form, on the device, through RCCL
send = torch.arange(12, dtype=torch.int32, device=dev)
recv = torch.empty_like(send)
ggms_store._all_to_all(dist, recv, send, [12], [12])
assert torch.equal(recv, send) and recv.is_cuda
N, dim = 5000, 24
feat = np.random.RandomState(1).standard_normal((N, dim)).astype(np.float32)
st = ggms_store.FeatureShards(torch.from_numpy(feat).to(dev), None, 1, 0, mode="a2a", dist=dist)
assert st._exchange_counts([7]) == [7]
nodes = np.random.RandomState(2).randint(0, N, 3000).astype(np.uint32)
out = torch.zeros((3000, dim), dtype=torch.float32, device=dev)
st.extract(torch.from_numpy(nodes.view(np.int32)).to(dev), 3000, out)
assert out.cpu().numpy().tobytes() == oracle.extract(feat, nodes).tobytes()
# the other collectives bench.py and connect_peers issue at N > 1, on the same backend: object broadcast / gather
# (staged on the device by RCCL, the gather from the deadline helper thread), the timing barrier, float64 reductions
where = ["/dev/shm/somewhere"]
dist.broadcast_object_list(where, src=0)
assert where == ["/dev/shm/somewhere"]
got = [None]
ggms_store.with_deadline(lambda: dist.all_gather_object(got, (b"h" * 64, 123)), "all_gather_object", seconds=60)
assert got == [(b"h" * 64, 123)]
dist.barrier()
torch.cuda.synchronize()
# the host-side meeting point of the long wait (engine sub-record): the rendezvous store, no kernel on the GPU
import datetime
store = dist.distributed_c10d._get_default_store()
store.set("ggms_bench_engine_done", "1")
store.wait(["ggms_bench_engine_done"], datetime.timedelta(seconds=10))
stats = torch.tensor([1.5, 2.0, 3.0], dtype=torch.float64, device=dev)
mx, sm = stats.clone(), stats.clone()
dist.all_reduce(mx, op=dist.ReduceOp.MAX)
dist.all_reduce(sm, op=dist.ReduceOp.SUM)
assert mx.tolist() == [1.5, 2.0, 3.0] and sm.tolist() == [1.5, 2.0, 3.0]
dist.destroy_process_group()
print("nccl-ok")
"""


def test_rccl_branches_of_the_exchange_store(tmp_path):
    """A world-size-1 RCCL process group drives `_all_to_all` / `_exchange_counts` through their on-device branches
    (one rank per GPU is RCCL's rule, so one rank is what this box can give them)."""
    script = tmp_path / "nccl1.py"
    script.write_text(_NCCL_SCRIPT.format(root=os.path.abspath(ROOT)))
    r = subprocess.run([sys.executable, str(script), str(_free_port())], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "nccl-ok" in r.stdout, r.stderr[-3000:]


@pytest.mark.parametrize("store", ["replica", "peer", "a2a", "hybrid"])
def test_two_ranks_one_gpu(store):
    env = dict(os.environ, GGMS_BENCH_DEVICE="0", GGMS_BENCH_BACKEND="gloo")
    port = _free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--preset", "tiny", "--steps", "4", "--warmup", "1", "--batch", "512",
                        "--store", store, "--cache-ratio", "0.6", "--other-stores", ""],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["value"] > 0
    assert store in d["config"]["workload"]


def test_two_ranks_sharing_the_gpu_at_full_size_never_stall_each_other():
    """Two processes, each with its own look-back chains (fused samplers, owner scans), on ONE GPU at papers100M size:
    the waiting workgroups of one used to hold the slots the other's next workgroup needed (GGMS_STATUS_SCAN_SPIN in 3 of
    6 runs, profiles/r03_two_ranks_one_gpu_before_selfserve_lookback.txt).  A look-back now computes a predecessor
    that does not show up itself, so both ranks always finish."""
    env = dict(os.environ, GGMS_BENCH_DEVICE="0", GGMS_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--store", "replica",
                        "--other-stores", "", "--no-engine", "--steps", "30"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["rows_verified"] and "device status" not in r.stderr


def _two_ranks_full_size(flags, timeout=900):
    env = dict(os.environ, GGMS_BENCH_DEVICE="0", GGMS_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "5", "--warmup", "1", "--repeats", "1", "--other-stores", ""] + flags,
                       capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return _last_json(r.stdout)


def test_configs3_papers100m_graphsage_feature_shards_full_size():
    """BASELINE configs[3] at its real size, as far as one GPU allows: papers100M-shaped CSR, GraphSAGE [25,10], the
    56.9-GB feature table split into GGMS shards over two ranks (two processes on the box's one GPU, 28.4 GB each --
    a size hipIpc could not open before ggms_ipc_safe_bytes), rows read from the peer's shard inside the gather
    kernel; bench.py checks the gathered rows of the last batch against the generator (rows_verified)."""
    d = _two_ranks_full_size(["--fanout", "25,10", "--store", "peer"])
    assert d["rows_verified"] and d["n_gpus"] == 2 and "N=111059956" in d["config"]["workload"]
    st = d["stores"]["peer"]
    assert 0.45 < st["remote_row_fraction"] < 0.55 and st["rows_by_tier"]["host"] == 0


def test_configs4_friendster_pinsage_hybrid_store_full_size():
    """BASELINE configs[4] at its real size on two ranks: Friendster-scale CSR (N 65.6 M, 256-dim rows), PinSAGE random
    walks, hybrid GGMS -- the hotter half of the rows in HBM (a quarter of those replicated on both ranks, the rest
    sharded), every row also in pinned host DRAM; one gather serves replica, local shard, peer shard and host rows."""
    d = _two_ranks_full_size(["--preset", "friendster", "--sample-type", "random_walk", "--fanout", "5,5,5",
                              "--store", "hybrid", "--cache-ratio", "0.5", "--replicate-frac", "0.25"], timeout=1200)
    assert d["rows_verified"] and d["n_gpus"] == 2 and "N=65608366" in d["config"]["workload"]
    t = d["stores"]["hybrid"]["rows_by_tier"]
    assert min(t["host"], t["remote_shard"], t["local_shard"], t["replica"]) > 0
    assert abs(sum(t.values()) - d["stores"]["hybrid"]["rows_per_step"] * 5) < 1
