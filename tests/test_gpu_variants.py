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
