"""Slower, result-identical forms of kernels that default sizes rarely reach, forced through the library's test aids
(ggms_debug_set_knob, include/ggms.h -- the library reads no environment variable for any of them), and the per-batch
status word."""
import numpy as np
import pytest

import oracle
import test_gpu_parity as P
from graphgen import powerlaw_csr

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

KHOP0_DRAW_CAP, OWNER_SCAN_CHUNKS, OWNER_SCAN_TILES = 0, 1, 2
P_dev, P_u32 = P.dev, P.host_u32


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (run with -m gpu on an MI355X box)")
    from xgnn_amd import ops as o
    return o


@pytest.fixture()
def knob():
    from xgnn_amd import lib
    used = []

    def set_knob(which, value):
        used.append(which)
        lib().ggms_debug_set_knob(which, value)
    yield set_knob
    for w in used:
        lib().ggms_debug_set_knob(w, -1)


@pytest.mark.parametrize("which,value", [(OWNER_SCAN_CHUNKS, 24), (OWNER_SCAN_TILES, 1)])
def test_owner_scan_forms(ops, knob, which, value):
    """The ordered owner scan has three forms with the same results: chunks held in registers (default), chunks too
    long for that (forced here by allowing only 24 chunks: every chunk is re-read for the write-out), and the
    tile-chained decoupled look-back kernel (by default only beyond 65 M items per fill)."""
    knob(which, value)
    for n in (1025, 100_000):
        P.test_hashtable_fill_and_map(ops, n, True)
    P.test_hashtable_reference_unittest_vectors(ops, True)
    for stype in ("khop3", "khop0"):
        P.test_sample_batch_vs_oracle(ops, stype, [5, 10, 15], 300, True)
        P.test_sample_batch_vs_oracle(ops, stype, [25, 10], 1000, True)
    P.test_fused_khop3_random_shapes(ops)
    P.test_samplers_on_hub_graph(ops, "khop3", 127)
    P.test_full_size_batch_properties(ops)
    assert ops.device_status() == 0


@pytest.mark.parametrize("cap", [0, 3000])
def test_khop0_draw_buffer_overflow_path(ops, knob, cap):
    """khop0 parks raw draws in a buffer and resolves them in a second kernel; seeds that do not fit are resolved in
    place by the generating lanes.  A tiny buffer must give the same results."""
    knob(KHOP0_DRAW_CAP, cap)
    for direct in (True, False):
        P.test_sample_batch_vs_oracle(ops, "khop0", [25, 10], 1000, direct)
        P.test_sample_batch_vs_oracle(ops, "khop0", [5, 10, 15], 300, direct)
    P.test_samplers_on_hub_graph(ops, "khop0", 2048)
    P.test_samplers_on_hub_graph(ops, "khop0", 100)
    P.test_khop0_fanout_beyond_the_lds_slots(ops, 2049)
    assert ops.device_status() == 0


def test_a_failed_batch_reports_its_own_status_and_only_it(ops):
    """Two batches in flight on two pipelines (own stream, table and workspace each).  Pipeline 0's hashed table is
    too small for its batch (GGMS_STATUS_TABLE_FULL inside the batch's kernels); pipeline 1's is not.  Every kernel of
    a batch reports into the BATCH's own status word (the word behind its table's item counter), so the failure
    comes out in the failed batch's counts[3 L + 1] and ONLY there -- whichever batch ends first -- and the next
    batch on the same table starts clean.  The device's sticky word records that a batch failed."""
    ip, ix = powerlaw_csr(20_000, mean_deg=30, seed=9)
    g = ops.DeviceGraph(P.dev(ip), P.dev(ix))
    fanouts, nseed, L = [10, 5], 1500, 2
    assert ops.device_status(clear=True) == 0
    bs = ops.BatchSampler(g, fanouts, nseed, sample_type=ops.KHOP3, seed=5, num_slots=4, num_pipelines=2, direct_table=False)
    full_size = bs.hts[0].c.o2n_size
    rng = np.random.RandomState(3)
    seeds = [rng.permutation(20_000)[:nseed].astype(np.uint32) for _ in range(4)]
    t_seeds = [P.dev(x) for x in seeds]
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    bs.hts[0].c.o2n_size = 64  # 64 buckets for ~1500 + 15000 + ... keys: the probe sequence finds no free bucket
    for b in range(2):
        with torch.cuda.stream(streams[b % 2]):
            bs.sample(t_seeds[b], slot=b, copy_input_nodes=True)
    torch.cuda.synchronize()
    st = [int(bs.counts_slots[b][3 * L + 1].item()) for b in range(2)]
    assert st[0] & 2 and st[1] == 0, st
    assert ops.device_status(clear=True) & 2  # the sticky record of the failed batch
    # the clean neighbour's results are what the one-at-a-time loop gives for batch 1 AFTER batch 0's draws: the RNG
    # pool is consumed in batch order whatever happened to batch 0's table -- so check it structurally instead
    c = bs.counts_slots[1].cpu().tolist()
    inp = P.host_u32(bs.input_nodes[1], c[3 * L])
    assert np.unique(inp).size == inp.size and np.array_equal(inp[:nseed], seeds[1])
    # the same table, full size again: its next batch is clean (the failed batch's last kernel took the word)
    bs.hts[0].c.o2n_size = full_size
    for b in (2, 3):
        with torch.cuda.stream(streams[b % 2]):
            bs.sample(t_seeds[b], slot=b, copy_input_nodes=True)
    torch.cuda.synchronize()
    assert [int(bs.counts_slots[b][3 * L + 1].item()) for b in (2, 3)] == [0, 0]
    assert ops.device_status() == 0


# ------------------------------------------------------------------ distinct seeds (ggms_sample_extra_t.seeds_distinct)
def _batch_equal(got, want, L, data=False):
    np.testing.assert_array_equal(P.host_u32(got["input_nodes"]), want["input_nodes"])
    for i in range(L):
        gl, wl = got["layers"][i], want["layers"][i]
        assert (gl["num_src"], gl["num_dst"]) == (wl["num_src"], wl["num_dst"]), i
        np.testing.assert_array_equal(P.host_u32(gl["row"]), wl["row"], err_msg=f"row layer {i}")
        np.testing.assert_array_equal(P.host_u32(gl["col"]), wl["col"], err_msg=f"col layer {i}")
        if data:
            np.testing.assert_array_equal(P.host_u32(gl["data"]), wl["data"])


@pytest.mark.parametrize("stype", ["khop3", "khop0", "khop2", "khop1", "weighted", "random_walk"])
@pytest.mark.parametrize("fanouts,nseed", [([25, 10], 1000), ([5, 10, 15], 300), ([3], 129), ([4, 3], 17_000), ([7], 1)])
def test_distinct_seed_promise_gives_the_oracles_batch(ops, stype, fanouts, nseed):
    """A caller that knows its seeds to be distinct (a slice of a shuffled train set) says so: the seeds' insert /
    ordered scan / look-up launches are skipped -- khop3 enters the seeds inside the first layer's launch, where a
    neighbour instance that reached a seed's word first is beaten through `lost`; the others use one small launch.
    The batch is the oracle's FillWithDupRevised(seeds) batch (dist_loops.cc:105-111), bit for bit, on a graph small
    enough that many seeds are also sampled as neighbours of other seeds (both arrival orders occur)."""
    ip, ix = powerlaw_csr(20_000, mean_deg=30, seed=2)
    N = ip.size - 1
    t_ix = P.dev(ix)
    g = ops.DeviceGraph(P.dev(ip), t_ix)
    rng = np.random.RandomState(len(fanouts) * 1000 + nseed)
    kw, okw = {}, {}
    if stype == "weighted":
        prob = rng.random_sample(ix.size).astype(np.float32)
        alias = rng.randint(0, N, ix.size).astype(np.uint32)
        kw = dict(prob_table=P.dev(prob), alias_table=P.dev(alias))
        okw = dict(prob=prob, alias=alias)
    if stype == "random_walk":
        kw = dict(random_walk_length=3, random_walk_restart_prob=0.5, num_random_walk=4)
        okw = dict(walk_length=3, restart_prob=0.5, num_walk=4)
    code = {"khop3": ops.KHOP3, "khop0": ops.KHOP0, "khop2": ops.KHOP2, "khop1": ops.KHOP1, "weighted": ops.WEIGHTED_KHOP,
            "random_walk": ops.RANDOM_WALK}[stype]
    ocode = {"khop3": oracle.KHOP3, "khop0": oracle.KHOP0, "khop2": oracle.KHOP2, "khop1": oracle.KHOP1,
             "weighted": oracle.WEIGHTED_KHOP, "random_walk": oracle.RANDOM_WALK}[stype]
    bs = ops.BatchSampler(g, fanouts, nseed, sample_type=code, seed=77, **kw)
    orc_states = oracle.random_states(bs.states.shape[0], 77) if stype != "khop0" else None
    ix = ix.copy()  # khop2 permutes the oracle's CSR too
    for rep in range(3):
        seeds = rng.permutation(N)[:nseed].astype(np.uint32)
        bs.sample(P.dev(seeds), distinct=(rep != 1))  # the middle batch takes the general path: the two interleave freely
        got = bs.result()
        want = oracle.do_sample(ocode, ip, ix, seeds, fanouts, orc_states, **okw)
        _batch_equal(got, want, len(fanouts), data=(stype == "random_walk"))
    np.testing.assert_array_equal(P.host_u32(t_ix), ix)
    assert ops.device_status() == 0


def test_distinct_seeds_beyond_one_tile_per_workgroup(ops):
    """khop3 enters distinct seeds inside the first layer's launch only while every tile has a workgroup of its own
    (<= 2048 tiles = 262144 seeds); a larger first layer takes the one-launch k_seed_enter + the ticketed layer kernel.
    300 000 distinct seeds: the oracle's batch."""
    ip, ix = powerlaw_csr(400_000, mean_deg=6, seed=12)
    g = ops.DeviceGraph(P_dev(ip), P_dev(ix))
    fanouts, nseed = [3, 2], 300_000
    bs = ops.BatchSampler(g, fanouts, nseed, sample_type=ops.KHOP3, seed=21)
    st = oracle.random_states(bs.states.shape[0], 21)
    rng = np.random.RandomState(5)
    for rep in range(2):
        seeds = rng.permutation(400_000)[:nseed].astype(np.uint32)
        bs.sample(P_dev(seeds), distinct=True)
        _batch_equal(bs.result(), oracle.do_sample(oracle.KHOP3, ip, ix, seeds, fanouts, st), 2)
    assert ops.device_status() == 0


def test_distinct_seed_batches_in_flight(ops):
    """Three pipelines, seven batches, every one with the distinct-seed promise: batch order on the RNG pool holds and
    every batch equals the one-at-a-time loop's (each pipeline's prologue rides on its own first-layer launch)."""
    ip, ix = powerlaw_csr(20_000, mean_deg=30, seed=4)
    g = ops.DeviceGraph(P.dev(ip), P.dev(ix))
    fanouts, nseed, K, NB = [10, 5], 1500, 3, 7
    bs = ops.BatchSampler(g, fanouts, nseed, sample_type=ops.KHOP3, seed=5, num_slots=NB, num_pipelines=K)
    orc_states = oracle.random_states(bs.states.shape[0], 5)
    rng = np.random.RandomState(8)
    seeds = [rng.permutation(20_000)[:nseed].astype(np.uint32) for _ in range(NB)]
    t_seeds = [P.dev(x) for x in seeds]
    streams = [torch.cuda.Stream() for _ in range(K)]
    torch.cuda.synchronize()
    for b in range(NB):
        with torch.cuda.stream(streams[b % K]):
            bs.sample(t_seeds[b], slot=b, copy_input_nodes=True, distinct=True)
    torch.cuda.synchronize()
    for b in range(NB):
        want = oracle.do_sample(oracle.KHOP3, ip, ix, seeds[b], fanouts, orc_states)
        c = bs.counts_slots[b].cpu().tolist()
        assert c[3 * len(fanouts) + 1] == 0
        np.testing.assert_array_equal(P.host_u32(bs.input_nodes[b], c[3 * len(fanouts)]), want["input_nodes"])
        for i in range(len(fanouts)):
            wl = want["layers"][i]
            assert (c[3 * i], c[3 * i + 1], c[3 * i + 2]) == (wl["row"].size, wl["num_src"], wl["num_dst"]), (b, i)
            np.testing.assert_array_equal(P.host_u32(bs.rows[b][i], c[3 * i]), wl["row"], err_msg=f"row {b}/{i}")
            np.testing.assert_array_equal(P.host_u32(bs.cols[b][i], c[3 * i]), wl["col"], err_msg=f"col {b}/{i}")


# ------------------------------------------------------------------ sharded topology (DeviceDistGraph)
@pytest.mark.parametrize("P", [1, 2, 3, 8])
@pytest.mark.parametrize("stype", ["khop3", "khop0", "random_walk"])
def test_batches_through_topology_shards(ops, P, stype):
    """DeviceDistGraph (cuda/dist_graph.h:114-158): nodes below num_cache_node in P shards (v % P, v / P), the rest in
    the whole CSR of the last slot.  Shards from the oracle's _DatasetPartition restatement AND from the GPU builder
    (ggms_store.topology_shards) -- equal -- and a batch through the view equal to the oracle's on the plain CSR."""
    from xgnn_amd import ggms_store
    ip, ix = powerlaw_csr(30_000, mean_deg=20, seed=6)
    N = ip.size - 1
    ncn = ggms_store.num_cache_node_for(ip, 0.6)
    assert 0 < ncn < N and ncn == oracle.num_cache_node(ip, 0.6)
    t_ip, t_ix = P_dev(ip), P_dev(ix)
    pip, pix = ggms_store.topology_shards(t_ip, t_ix, P, ncn)
    for p_ in range(P):
        oip, oix = oracle.partition_graph(ip, ix, p_, P, ncn)
        np.testing.assert_array_equal(P_u32(pip[p_]), oip)
        np.testing.assert_array_equal(P_u32(pix[p_])[:oix.size], oix)
    g = ops.DeviceGraph(None, None, part_indptr=pip + [t_ip], part_indices=pix + [t_ix], num_cache_node=ncn)
    kw, okw, fanouts = {}, {}, [10, 5]
    if stype == "random_walk":
        kw = dict(random_walk_length=3, random_walk_restart_prob=0.5, num_random_walk=4)
        okw = dict(walk_length=3, restart_prob=0.5, num_walk=4)
        fanouts = [5, 5]
    code = {"khop3": ops.KHOP3, "khop0": ops.KHOP0, "random_walk": ops.RANDOM_WALK}[stype]
    ocode = {"khop3": oracle.KHOP3, "khop0": oracle.KHOP0, "random_walk": oracle.RANDOM_WALK}[stype]
    bs = ops.BatchSampler(g, fanouts, 1200, sample_type=code, seed=9, **kw)
    orc_states = oracle.random_states(bs.states.shape[0], 9) if stype != "khop0" else None
    rng = np.random.RandomState(P)
    for rep in range(2):
        seeds = rng.permutation(N)[:1200].astype(np.uint32)
        bs.sample(P_dev(seeds), distinct=(rep == 0))
        _batch_equal(bs.result(), oracle.do_sample(ocode, ip, ix, seeds, fanouts, orc_states, **okw), 2,
                     data=(stype == "random_walk"))
    assert ops.device_status() == 0


def test_more_shards_than_the_kernels_carry_is_refused(ops):
    """GGMS_MAX_PARTS = 8 shard pointers travel in the kernel arguments; nine are an argument error, not a fault -- in the
    Python wrapper before anything is built (ops.PartTable), and in the library for a caller that has no wrapper."""
    ip, ix = powerlaw_csr(2000, mean_deg=8, seed=5)
    t_ip, t_ix = P_dev(ip), P_dev(ix)
    parts = [oracle.partition_graph(ip, ix, r, 9, 900) for r in range(9)]
    with pytest.raises(ValueError, match="GGMS_MAX_PARTS"):
        ops.DeviceGraph(None, None, part_indptr=[P_dev(p[0]) for p in parts] + [t_ip],
                        part_indices=[P_dev(p[1]) for p in parts] + [t_ix], num_cache_node=900)
    g = ops.DeviceGraph(None, None, part_indptr=[P_dev(p[0]) for p in parts[:8]] + [t_ip],
                        part_indices=[P_dev(p[1]) for p in parts[:8]] + [t_ix], num_cache_node=900)
    g.c.num_part = 9  # what a C caller could hand over: refused before any table entry is read
    with pytest.raises(RuntimeError, match="at most 8"):
        ops.sample_khop3(g, P_dev(np.arange(10, dtype=np.uint32)), 3, ops.random_states(256, 1))


def test_a_device_pointer_table_is_an_argument_error(ops):
    """the shard pointer tables became HOST arrays with ABI 3; a caller still passing a device array must get
    GGMS_ERR_INVALID, not a host dereference of device memory."""
    import ctypes as C
    import torch
    from xgnn_amd import _lib
    dev = torch.device("cuda", 0)
    src = torch.arange(64 * 8, dtype=torch.float32, device=dev).reshape(64, 8)
    dev_table = torch.tensor([src.data_ptr()], dtype=torch.int64, device=dev)  # the pre-ABI-3 form
    idx = torch.arange(16, dtype=torch.int32, device=dev)
    out = torch.zeros((16, 8), dtype=torch.float32, device=dev)
    rc = _lib.lib().ggms_gather_scatter_partition(C.c_void_p(out.data_ptr()), C.c_void_p(dev_table.data_ptr()), 1,
                                                  C.c_void_p(idx.data_ptr()), None, 16, None, 8, 0, None)
    assert rc == -1 and b"HOST arrays" in _lib.lib().ggms_last_error()
    tab = ops.PartTable([src.data_ptr()])
    assert _lib.lib().ggms_gather_scatter_partition(C.c_void_p(out.data_ptr()), tab.ptr(), 1, C.c_void_p(idx.data_ptr()), None,
                                                    16, None, 8, 0, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, src[:16])
