"""The ordered scans must not depend on when or where another workgroup runs (two processes, or two batches, sharing
a GPU: waiting workgroups of one kernel can hold the slots the other kernel's next workgroup needs).  A look-back
therefore waits a bounded number of polls and then computes the missing predecessor words from the scan's input.
Here: patience 0 -- every look-back that finds a word missing serves itself at once, owners and helpers publish the
same words concurrently -- on a cross-section of the parity tests; results stay bit-exact."""
import pytest

import test_gpu_parity as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (run with -m gpu on an MI355X box)")
    from xgnn_amd import ops as o
    return o


def session_patience():
    """What the session runs with (conftest.scan_patience_of_this_run): GGMS_TEST_SCAN_PATIENCE, else the default."""
    import os
    return int(os.environ.get("GGMS_TEST_SCAN_PATIENCE", "2048"))


@pytest.fixture()
def impatient(ops):
    from xgnn_amd import lib
    lib().ggms_debug_set_scan_patience(0)
    yield
    lib().ggms_debug_set_scan_patience(session_patience())


def test_fused_khop3_and_every_sampler_random_shapes(ops, impatient):
    P.test_fused_khop3_random_shapes(ops)
    P.test_all_samplers_random_shapes(ops)


@pytest.mark.parametrize("direct", [True, False])
@pytest.mark.parametrize("stype", ["khop3", "khop0", "khop2", "khop1"])
def test_batches(ops, impatient, stype, direct):
    P.test_sample_batch_vs_oracle(ops, stype, [5, 10, 15], 300, direct)
    P.test_sample_batch_vs_oracle(ops, stype, [25, 10], 1000, direct)


@pytest.mark.parametrize("stype", ["weighted", "random_walk", "random_walk_long"])
def test_weighted_and_random_walk_batches(ops, impatient, stype):
    P.test_sample_batch_weighted_and_random_walk(ops, stype)


@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("n", [1025, 100_000])
def test_table_fill(ops, impatient, n, direct):
    P.test_hashtable_fill_and_map(ops, n, direct)
    P.test_hashtable_reference_unittest_vectors(ops, direct)


def test_split_and_hub_graph(ops, impatient):
    P.test_get_miss_cache_index(ops, 100_000, 0.64)
    P.test_samplers_on_hub_graph(ops, "khop3", 127)
    P.test_samplers_on_hub_graph(ops, "khop0", 2048)


def test_full_size_batch(ops, impatient):
    P.test_full_size_batch_properties(ops)
    assert ops.device_status() == 0


@pytest.mark.parametrize("n,direct", [(5000, True), (300_000, True), (60_000, False)])
def test_owner_of_the_first_chunk_arrives_after_the_total_is_out(ops, n, direct):
    """The table already holds items (its count is updated IN PLACE by the scan).  Chunk 0's workgroup is held back for
    about 2 ms; with a patience of 4 polls every other chunk counts chunk 0 itself, the last one replaces the count by
    the total -- and only then does chunk 0's owner start: it must take the base from the word the others published,
    not from the (already overwritten) count.  Local ids, the unique list and the count stay the oracle's."""
    import numpy as np
    import oracle
    from xgnn_amd import lib
    rng = np.random.RandomState(n)
    universe = n // 2 + 7
    # direct layout: the chunked owner scan; hashed layout: the generic single-pass scan over the same in-place count
    ht = ops.OrderedHashTable(4 * n + 16, num_node=(universe + 1) if direct else None)
    orc = oracle.HashTable(universe + 1, 4 * n + 16)
    ht.reset()
    orc.reset()
    first = rng.randint(0, universe, n // 3).astype(np.uint32)
    P._check_fill(ops, ht, orc, first)  # base != 0 for the fill under test
    assert ht.num_items > 0
    lib().ggms_debug_set_scan_patience(4)
    try:
        for _ in range(3):
            items = rng.randint(0, universe, n).astype(np.uint32)
            lib().ggms_debug_delay_next_scan(500)  # one shot: about 2 ms
            P._check_fill(ops, ht, orc, items)
            ns, _ = ht.map_edges(P.dev(items), None)
            os_, _ = orc.map_edges(items, items)
            np.testing.assert_array_equal(P.host_u32(ns), os_)
        assert ops.device_status() == 0
    finally:
        lib().ggms_debug_set_scan_patience(session_patience())
