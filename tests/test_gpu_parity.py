"""GPU parity: HIP path (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import oracle
from graphgen import hub_csr, exact_features, powerlaw_csr

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (run with -m gpu on an MI355X box)")
    from xgnn_amd import ops as _ops
    return _ops


def dev(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).cuda()


def host_u32(t, n=None):
    a = t.cpu().numpy()
    if n is not None:
        a = a[:n]
    return a.view(np.uint32) if a.dtype == np.int32 else a


def states_np(t):
    a = t.cpu().numpy().view(np.uint32)
    return a


# ------------------------------------------------------------------ RNG
def test_xorwow_state_pool(ops):
    st = ops.random_states(1000, 0x5EED)
    want = oracle.random_states(1000, 0x5EED)
    got = states_np(st)
    np.testing.assert_array_equal(got[:, 0], want["d"])
    np.testing.assert_array_equal(got[:, 1:], want["v"])
    # the committed known-answer file (constants + first states; tests/golden/gen_xorwow_golden.py)
    import json, os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xorwow_constants.json")))
    for seed in (0, 1, 0x5EED):
        one = states_np(ops.random_states(1, seed))[0]
        assert int(one[0]) == gold["first_states"][str(seed)]["d"]
        assert [int(x) for x in one[1:]] == gold["first_states"][str(seed)]["v"]


# ------------------------------------------------- device-side failures are reported, not swallowed
def test_full_hashed_table_sets_the_status_word(ops):
    """A hashed dedup table with fewer buckets than keys: the reference would probe forever (its CHECKs assume the
    PredictNumNodes sizing); here the insert gives up after one full probe cycle and says so."""
    assert ops.device_status(clear=True) == 0
    ht = ops.OrderedHashTable(8)  # TableSize(8) = 32 buckets
    ht.reset()
    keys = dev(np.arange(1000, 1200, dtype=np.uint32))
    ht.fill_with_duplicates(keys)
    assert ops.device_status() & 2  # GGMS_STATUS_TABLE_FULL
    with pytest.raises(RuntimeError):
        ops.check_device_status("fill")
    assert ops.device_status() == 0  # cleared by the check


def test_stuck_scan_sets_the_status_word(ops, monkeypatch):
    """Look-back that can never complete (ticket poisoned so that tile 0 is skipped): bounded spin, status bit 1,
    no hang and no out-of-range write."""
    assert ops.device_status(clear=True) == 0
    N = 50_000
    ht = ops.OrderedHashTable(N, num_node=N)
    ht.reset()
    keys = dev(np.random.RandomState(1).randint(0, N, 10_000).astype(np.uint32))
    from xgnn_amd import lib
    lib().ggms_debug_poison_next_scan()  # one shot
    ht.fill_with_duplicates(keys)
    assert ops.device_status(clear=True) & 1  # GGMS_STATUS_SCAN_SPIN
    ht.reset()
    ht.fill_with_duplicates(keys)  # the same table and a clean scan area work again
    assert ops.device_status() == 0
    assert ht.num_items == np.unique(host_u32(keys)).size


# -------------------------------------------------------------- extract
@pytest.mark.parametrize("dtype,dim", [(np.float32, 100), (np.float32, 128), (np.float32, 256), (np.float32, 1),
                                       (np.float64, 5), (np.int16, 7), (np.uint8, 3), (np.uint8, 33),
                                       (np.int32, 32), (np.int64, 1), (np.float32, 602)])
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000, 4097])
def test_extract(ops, dtype, dim, n):
    rng = np.random.RandomState(dim * 31 + n)
    N = 5000
    table = exact_features(N, dim, dtype)
    idx = rng.randint(0, N, n).astype(np.uint32)
    t_table = dev(table)
    got = ops.extract(t_table, dev(idx)) if n else ops.extract(t_table, torch.zeros(0, dtype=torch.int32, device="cuda"))
    want = oracle.extract(table, idx)
    assert got.cpu().numpy().tobytes() == want.tobytes()


@pytest.mark.parametrize("bits", [1, 7, 13])
def test_mock_extract(ops, bits):
    """GPUMockExtract (cuda_extraction.cu:51-70): rows of a 2^k-row stand-in table, index & (2^k - 1)."""
    rng = np.random.RandomState(bits)
    table = exact_features(1 << bits, 100, np.float32)
    idx = rng.randint(0, 2 ** 31, 5000).astype(np.uint32)
    got = ops.mock_extract(dev(table), dev(idx), bits)
    assert got.cpu().numpy().tobytes() == table[idx & ((1 << bits) - 1)].tobytes()


def test_extract_row_sizes_around_the_tile_sweep_limit(ops):
    """Rows up to 8191 chunks of 16 B (131 056 B) go through the tile sweep, wider ones through the one-row-per-
    workgroup kernel; odd widths pick narrower chunks.  All bit-exact, cached / miss-counted variant included."""
    rng = np.random.RandomState(4)
    for dim, dtype in ((4 * 8191, np.float32), (4 * 8192, np.float32), (4 * 8192 + 4, np.float32),
                       (40001, np.float32), (70001, np.uint8), (300001, np.uint8)):
        table = rng.randint(0, 250, (40, dim)).astype(dtype)
        idx = rng.randint(0, 40, 23).astype(np.uint32)
        out = ops.extract(dev(table), dev(idx))
        assert out.cpu().numpy().tobytes() == oracle.extract(table, idx).tobytes(), dim
    # a cached gather of long rows over 2 shards: 40 % of the nodes cached, the others read from the "host" copy
    dim, N, P = 4 * 9000, 64, 2
    feat = rng.randint(0, 1000, (N, dim)).astype(np.float32)
    rank_s, table = oracle.cache_build(rng.permutation(N).astype(np.uint32), 26, True)
    parts = [dev(oracle.partition_feature(feat, rank_s, 26, p, P)) for p in range(P)]
    ptab = ops.part_pointer_table(parts, "cuda")
    nodes = rng.randint(0, N, 50).astype(np.uint32)
    out = torch.zeros((50, dim), dtype=torch.float32, device="cuda")
    nmiss = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.extract_cached(out, dev(nodes), dev(table), ptab, P, dev(feat), num_miss=nmiss)
    assert out.cpu().numpy().tobytes() == feat[nodes].tobytes()
    assert int(nmiss.item()) == int((table[nodes] == 0xFFFFFFFF).sum())


def test_extract_large_properties(ops):
    """products-sized gather: checksum + idempotence (size-independent properties)."""
    N, dim, n = 2_449_029, 100, 1_190_000
    feat = torch.arange(N * dim, dtype=torch.int32, device="cuda").bitwise_and_(0xFFFF).to(torch.float32).view(N, dim)
    g = torch.Generator(device="cuda").manual_seed(1)
    idx = torch.randint(0, N, (n,), dtype=torch.int32, device="cuda", generator=g)
    out = ops.extract(feat, idx)
    ref = feat[idx.long()]
    assert torch.equal(out, ref)
    again = ops.extract(out, torch.arange(n, dtype=torch.int32, device="cuda"))
    assert torch.equal(again, out)


# ------------------------------------------------------------ hash table
def _check_fill(ops, ht_gpu, ht_orc, items):
    ht_gpu.fill_with_duplicates(dev(items))
    ht_orc.fill_with_duplicates(items)
    assert ht_gpu.num_items == ht_orc.num_items
    np.testing.assert_array_equal(host_u32(ht_gpu.unique()), ht_orc.unique())


@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("n", [1, 17, 512, 1024, 1025, 100_000])
def test_hashtable_fill_and_map(ops, n, direct):
    """Semantics pinned by samgraph/unittest/test_hashmap.cc:84-276 (set equality, prefix
    stability, local id == position) plus first-occurrence order vs the oracle."""
    rng = np.random.RandomState(n)
    universe = max(4, n // 2)
    cap = 4 * n + 16
    ht = ops.OrderedHashTable(cap, num_node=(universe + 1) if direct else None)
    orc = oracle.HashTable(universe + 1, cap)
    for rnd in range(3):  # reuse with Reset (test_hashmap.cc:275-276)
        ht.reset()
        orc.reset()
        a = rng.randint(0, universe, n).astype(np.uint32)
        _check_fill(ops, ht, orc, a)
        prefix = host_u32(ht.unique()).copy()
        b = rng.randint(0, universe, 2 * n).astype(np.uint32)
        _check_fill(ops, ht, orc, b)
        uniq = host_u32(ht.unique())
        np.testing.assert_array_equal(uniq[: prefix.size], prefix)          # prefix stability
        assert set(uniq.tolist()) == set(a.tolist()) | set(b.tolist())      # set equality
        assert len(set(uniq.tolist())) == uniq.size                         # no duplicates
        ns, nd = ht.map_edges(dev(a), dev(b[:n]))
        os_, od = orc.map_edges(a, b[:n])
        np.testing.assert_array_equal(host_u32(ns), os_)
        np.testing.assert_array_equal(host_u32(nd), od)
        # SearchO2N(unique[i]).local == i  (test_hashmap.cu:24-37)
        loc, _ = ht.map_edges(ht.unique().contiguous(), None)
        np.testing.assert_array_equal(host_u32(loc), np.arange(uniq.size, dtype=np.uint32))


@pytest.mark.parametrize("direct", [False, True])
def test_hashtable_reference_unittest_vectors(ops, direct):
    """The reference's own unit test, samgraph/unittest/test_hashmap.cc:84-276 (DupRevised_Ref, MixedFillMethod,
    Large), with its literal inputs and its assertions; first-occurrence order on top (canonical semantics)."""
    from graphgen import REF_HASHMAP_VECTORS, REF_HASHMAP_EXPECT

    def run(fills, num_node, cap):
        ht = ops.OrderedHashTable(cap, num_node=num_node if direct else None)
        ht.reset()
        seen, prev = [], np.zeros(0, np.uint32)
        sizes = []
        for data in fills:
            a = np.asarray(data, np.uint32)
            ht.fill_with_duplicates(dev(a))
            uniq = host_u32(ht.unique()).copy()
            seen = list(dict.fromkeys(list(seen) + a.tolist()))
            assert len(set(uniq.tolist())) == uniq.size           # unique
            np.testing.assert_array_equal(uniq[: prev.size], prev)  # prefix kept
            np.testing.assert_array_equal(uniq, np.array(seen, np.uint32))  # set equality + first-occurrence order
            loc, _ = ht.map_edges(ht.unique().contiguous(), None)  # ValidateSearch (test_hashmap.cu:24-37)
            np.testing.assert_array_equal(host_u32(loc), np.arange(uniq.size, dtype=np.uint32))
            prev = uniq
            sizes.append(uniq.size)
        return sizes

    for name, fills in REF_HASHMAP_VECTORS.items():
        sizes = run(fills, 2048, 4096)
        if name in REF_HASHMAP_EXPECT:
            assert sizes == REF_HASHMAP_EXPECT[name]
    # Large (:200-273): 1 M draws from [1, 800000], then (800000, 1600000], then [1, 1600000]
    rng = np.random.RandomState(12345)
    fills = [rng.randint(1, 800001, 1_000_000), rng.randint(800001, 1600001, 1_000_000),
             rng.randint(1, 1600001, 1_000_000)]
    run(fills, 1_600_001, 3_000_000)


def test_hashtable_unique_out_and_empty(ops):
    ht = ops.OrderedHashTable(64)
    ht.reset()
    out = torch.full((64,), -1, dtype=torch.int32, device="cuda")
    ht.fill_with_duplicates(dev(np.array([5, 5, 9, 5, 1], np.uint32)), unique_out=out)
    np.testing.assert_array_equal(host_u32(out)[:3], [5, 9, 1])
    ht.fill_with_duplicates(torch.zeros(0, dtype=torch.int32, device="cuda"), num_input=0, unique_out=out)
    assert ht.num_items == 3


# --------------------------------------------------------------- samplers
GRAPHS = {
    "small": dict(num_node=300, mean_deg=12, seed=1),
    "mid": dict(num_node=20_000, mean_deg=30, seed=2),
    "wide": dict(num_node=70_000, mean_deg=8, seed=3),  # ids need 3 key bytes in the samplers' radix sort
    "ids23": dict(num_node=4_500_000, mean_deg=2, seed=4),  # ids beyond 2^22: three 11-bit passes
}


@pytest.fixture(scope="module")
def graphs(ops):
    out = {}
    for k, kw in GRAPHS.items():
        ip, ix = powerlaw_csr(**kw)
        out[k] = (ip, ix, ops.DeviceGraph(dev(ip), dev(ix)))
    return out


@pytest.mark.parametrize("gname,n,fanout", [("small", 1, 5), ("small", 129, 3), ("small", 300, 25), ("small", 0, 4),
                                            ("mid", 8000, 10), ("mid", 20000, 25), ("mid", 5000, 5),
                                            ("mid", 1000, 40), ("mid", 777, 100)])
def test_sample_khop3(ops, graphs, gname, n, fanout):
    ip, ix, g = graphs[gname]
    rng = np.random.RandomState(n + fanout)
    inp = rng.randint(0, ip.size - 1, n).astype(np.uint32)
    nstates = max(64, (n + 127) // 128 * 8)
    st_gpu = ops.random_states(nstates, 0xABCDEF)
    st_orc = oracle.random_states(nstates, 0xABCDEF)
    for rep in range(2):  # states persist across calls (khop3.cu:145)
        src, dst, num = ops.sample_khop3(g, dev(inp) if n else torch.zeros(0, dtype=torch.int32, device="cuda"),
                                         fanout, st_gpu)
        wsrc, wdst = oracle.sample_khop3(ip, ix, inp, fanout, st_orc)
        m = int(num.item())
        assert m == wsrc.size
        np.testing.assert_array_equal(host_u32(src, m), wsrc)
        np.testing.assert_array_equal(host_u32(dst, m), wdst)
        got_states = states_np(st_gpu)
        np.testing.assert_array_equal(got_states[:, 0], st_orc["d"])
        np.testing.assert_array_equal(got_states[:, 1:], st_orc["v"])


@pytest.mark.parametrize("gname,n,fanout", [("small", 1, 5), ("small", 65, 3), ("small", 300, 25), ("small", 0, 4),
                                            ("mid", 8000, 10), ("mid", 3000, 25), ("mid", 1000, 64)])
def test_sample_khop0(ops, graphs, gname, n, fanout):
    ip, ix, g = graphs[gname]
    rng = np.random.RandomState(n * 7 + fanout)
    inp = rng.randint(0, ip.size - 1, n).astype(np.uint32)
    src, dst, num = ops.sample_khop0(g, dev(inp) if n else torch.zeros(0, dtype=torch.int32, device="cuda"), fanout)
    wsrc, wdst = oracle.sample_khop0(ip, ix, inp, fanout)
    m = int(num.item())
    assert m == wsrc.size
    np.testing.assert_array_equal(host_u32(src, m), wsrc)
    np.testing.assert_array_equal(host_u32(dst, m), wdst)


@pytest.mark.parametrize("gname,n,fanout", [("small", 1, 5), ("small", 300, 25), ("small", 0, 4), ("mid", 8000, 10),
                                            ("mid", 30000, 25), ("mid", 5000, 3), ("wide", 9000, 6),
                                            ("mid", 8192, 4), ("mid", 8193, 4), ("wide", 8192, 2),
                                            ("mid", 1_100_000, 1), ("ids23", 9000, 3), ("ids23", 5000, 3)])
def test_sample_khop1(ops, graphs, gname, n, fanout):
    """With replacement + stable sort by src + adjacent-duplicate drop (khop1.cu:42-127); 30000*25 > 512 K tasks
    exercises the grid-stride reuse of a stream.  The seed sort has three forms by size: one workgroup in LDS (up to
    8192 seeds), 11-bit digits (up to 2^20), 8-bit digits beyond."""
    ip, ix, g = graphs[gname]
    rng = np.random.RandomState(n * 11 + fanout)
    inp = rng.randint(0, ip.size - 1, n).astype(np.uint32)
    nstates = max(256, min(n * fanout, 512 * 1024))
    st_gpu = ops.random_states(nstates, 0x1001)
    st_orc = oracle.random_states(nstates, 0x1001)
    for rep in range(2):
        src, dst, num = ops.sample_khop1(g, dev(inp) if n else torch.zeros(0, dtype=torch.int32, device="cuda"),
                                         fanout, st_gpu)
        wsrc, wdst = oracle.sample_khop1(ip, ix, inp, fanout, st_orc)
        m = int(num.item())
        assert m == wsrc.size
        np.testing.assert_array_equal(host_u32(src, m), wsrc)
        np.testing.assert_array_equal(host_u32(dst, m), wdst)
        got_states = states_np(st_gpu)
        np.testing.assert_array_equal(got_states[:, 0], st_orc["d"])
        np.testing.assert_array_equal(got_states[:, 1:], st_orc["v"])


@pytest.mark.parametrize("gname,n,fanout", [("small", 1, 5), ("small", 300, 25), ("small", 0, 4), ("small", 1025, 3),
                                            ("mid", 8000, 10), ("mid", 20000, 25), ("mid", 4097, 64)])
def test_sample_khop2(ops, graphs, gname, n, fanout):
    """In-place partial Fisher-Yates (khop2.cu:46-95): edges, RNG states AND the permuted CSR must match."""
    ip, ix, _ = graphs[gname]
    ix_orc = ix.copy()
    t_ix = dev(ix)  # private copy: the sampler permutes it
    g = ops.DeviceGraph(dev(ip), t_ix)
    rng = np.random.RandomState(n * 3 + fanout)
    inp = rng.permutation(ip.size - 1)[:n].astype(np.uint32)  # distinct seeds (see include/ggms.h)
    nstates = max(256, (n + 1023) // 1024 * 256)
    st_gpu = ops.random_states(nstates, 0x5EED2)
    st_orc = oracle.random_states(nstates, 0x5EED2)
    for rep in range(3):  # the second and third call sample from the permuted lists
        src, dst, num = ops.sample_khop2(g, dev(inp) if n else torch.zeros(0, dtype=torch.int32, device="cuda"),
                                         fanout, st_gpu)
        wsrc, wdst = oracle.sample_khop2(ip, ix_orc, inp, fanout, st_orc)
        m = int(num.item())
        assert m == wsrc.size
        np.testing.assert_array_equal(host_u32(src, m), wsrc)
        np.testing.assert_array_equal(host_u32(dst, m), wdst)
        np.testing.assert_array_equal(host_u32(t_ix), ix_orc)
        got_states = states_np(st_gpu)
        np.testing.assert_array_equal(got_states[:, 0], st_orc["d"])
        np.testing.assert_array_equal(got_states[:, 1:], st_orc["v"])
    # a permutation of each list, never a change of its contents
    for v in inp[:50]:
        assert sorted(ix_orc[ip[v]:ip[v + 1]]) == sorted(ix[ip[v]:ip[v + 1]])


def test_sample_khop2_rejects_sharded_graph(ops):
    ip, ix = powerlaw_csr(1000, mean_deg=8, seed=5)
    from oracle import partition_graph
    parts = [partition_graph(ip, ix, r, 2, 500) for r in range(2)]
    pip = [dev(p[0]) for p in parts] + [dev(ip)]
    pix = [dev(p[1]) for p in parts] + [dev(ix)]
    g = ops.DeviceGraph(None, None, part_indptr=pip, part_indices=pix, num_cache_node=1000)
    with pytest.raises(RuntimeError):
        ops.sample_khop2(g, dev(np.arange(10, dtype=np.uint32)), 3, ops.random_states(256, 1))


@pytest.mark.parametrize("gname,n,fanout", [("small", 1, 5), ("small", 300, 25), ("small", 0, 4), ("mid", 8000, 10),
                                            ("mid", 20000, 25), ("mid", 5000, 3), ("mid", 40000, 15),
                                            ("wide", 9000, 6)])
def test_sample_weighted_khop(ops, graphs, gname, n, fanout):
    """Alias-method sampler incl. stable sort by src and adjacent-duplicate compaction; duplicated seeds too."""
    ip, ix, g = graphs[gname]
    N = ip.size - 1
    rng = np.random.RandomState(n * 3 + fanout)
    prob = rng.random_sample(ix.size).astype(np.float32)
    alias = rng.randint(0, N, ix.size).astype(np.uint32)
    inp = rng.randint(0, N, n).astype(np.uint32)  # with repeats: the stable sort must keep position order
    tasks = n * fanout
    nstates = max(256, min((min(tasks, 512 * 1024) + 255) // 256 * 256, max(tasks, 1)))
    st_gpu = ops.random_states(nstates, 4242)
    st_orc = oracle.random_states(nstates, 4242)
    t_prob, t_alias = dev(prob), dev(alias)
    for rep in range(2):
        src, dst, num = ops.sample_weighted_khop(g, t_prob, t_alias,
                                                 dev(inp) if n else torch.zeros(0, dtype=torch.int32, device="cuda"),
                                                 fanout, st_gpu)
        wsrc, wdst = oracle.sample_weighted_khop(ip, ix, prob, alias, inp, fanout, st_orc)
        m = int(num.item())
        assert m == wsrc.size
        np.testing.assert_array_equal(host_u32(src, m), wsrc)
        np.testing.assert_array_equal(host_u32(dst, m), wdst)
        got_states = states_np(st_gpu)
        np.testing.assert_array_equal(got_states[:, 0], st_orc["d"])
        np.testing.assert_array_equal(got_states[:, 1:], st_orc["v"])


@pytest.mark.parametrize("gname,n,fanout", [("small", 1, 5), ("small", 300, 25), ("small", 0, 4), ("mid", 8000, 10),
                                            ("mid", 30000, 25)])
def test_sample_weighted_khop_prefix(ops, graphs, gname, n, fanout):
    """Inverse-CDF draw over per-list prefix sums (weighted_khop_prefix.cu:41-91), f32 product + binary search."""
    from graphgen import prefix_sums
    ip, ix, g = graphs[gname]
    N = ip.size - 1
    rng = np.random.RandomState(n * 5 + fanout)
    pre = prefix_sums(ip, (rng.random_sample(ix.size) + 0.01).astype(np.float32))
    inp = rng.randint(0, N, n).astype(np.uint32)
    nstates = max(256, min(n * fanout, 512 * 1024))
    st_gpu = ops.random_states(nstates, 77)
    st_orc = oracle.random_states(nstates, 77)
    t_pre = dev(pre)
    for rep in range(2):
        src, dst, num = ops.sample_weighted_khop_prefix(
            g, t_pre, dev(inp) if n else torch.zeros(0, dtype=torch.int32, device="cuda"), fanout, st_gpu)
        wsrc, wdst = oracle.sample_weighted_khop_prefix(ip, ix, pre, inp, fanout, st_orc)
        m = int(num.item())
        assert m == wsrc.size
        np.testing.assert_array_equal(host_u32(src, m), wsrc)
        np.testing.assert_array_equal(host_u32(dst, m), wdst)
        got_states = states_np(st_gpu)
        np.testing.assert_array_equal(got_states[:, 0], st_orc["d"])
        np.testing.assert_array_equal(got_states[:, 1:], st_orc["v"])


@pytest.mark.parametrize("gname,n,fanout", [("small", 1, 5), ("small", 300, 25), ("small", 0, 4), ("small", 1025, 3),
                                            ("mid", 8000, 10), ("mid", 5000, 49)])
def test_sample_weighted_khop_hash_dedup(ops, graphs, gname, n, fanout):
    """Alias-method candidates until `fanout` distinct ids per seed; per-thread 50-slot table keyed by seed id
    (weighted_khop_hash_dedup.cu:41-117).  Repeated seeds included: a later copy sees the earlier copy's entries."""
    ip, ix, g = graphs[gname]
    N = ip.size - 1
    rng = np.random.RandomState(n * 13 + fanout)
    prob = rng.random_sample(ix.size).astype(np.float32)
    alias = rng.randint(0, N, ix.size).astype(np.uint32)
    inp = rng.randint(0, N, n).astype(np.uint32)
    if n >= 600:  # one thread meets the same high-degree seed three times: its 50-slot table fills up
        inp[[0, 256, 512]] = int(np.argmax(ip[1:] - ip[:-1]))
    nstates = max(256, (n + 1023) // 1024 * 256)
    st_gpu = ops.random_states(nstates, 99)
    st_orc = oracle.random_states(nstates, 99)
    t_prob, t_alias = dev(prob), dev(alias)
    for rep in range(2):
        src, dst, num = ops.sample_weighted_khop_hash_dedup(
            g, t_prob, t_alias, dev(inp) if n else torch.zeros(0, dtype=torch.int32, device="cuda"), fanout, st_gpu)
        wsrc, wdst = oracle.sample_weighted_khop_hash_dedup(ip, ix, prob, alias, inp, fanout, st_orc)
        m = int(num.item())
        assert m == wsrc.size
        np.testing.assert_array_equal(host_u32(src, m), wsrc)
        np.testing.assert_array_equal(host_u32(dst, m), wdst)
        got_states = states_np(st_gpu)
        np.testing.assert_array_equal(got_states[:, 0], st_orc["d"])
        np.testing.assert_array_equal(got_states[:, 1:], st_orc["v"])


@pytest.mark.parametrize("gname,n,wl,p,nw,K", [("small", 1, 3, 0.5, 4, 5), ("small", 300, 3, 0.5, 4, 5),
                                               ("small", 0, 3, 0.5, 4, 5), ("mid", 8000, 3, 0.5, 4, 5),
                                               ("mid", 5000, 4, 0.2, 5, 3), ("mid", 3000, 10, 0.1, 10, 20),
                                               ("mid", 1000, 2, 0.0, 1, 1), ("mid", 4000, 6, 0.1, 6, 10),
                                               ("mid", 2000, 8, 0.05, 16, 128), ("mid", 70000, 3, 0.5, 4, 5),
                                               ("mid", 600, 20, 0.02, 10, 7), ("mid", 150, 2, 0.1, 300, 400)])
def test_sample_random_walk(ops, graphs, gname, n, wl, p, nw, K):
    """PinSAGE neighbourhood: walks with restart + per-seed top-K by visit count (ties: first visit)."""
    ip, ix, g = graphs[gname]
    rng = np.random.RandomState(n + wl * 7 + nw)
    inp = rng.randint(0, ip.size - 1, n).astype(np.uint32)
    nstates = max(256, ops.lib().ggms_random_walk_num_states(n, nw))
    st_gpu = ops.random_states(nstates, 777)
    st_orc = oracle.random_states(nstates, 777)
    for rep in range(2):
        src, dst, data, num = ops.sample_random_walk(
            g, dev(inp) if n else torch.zeros(0, dtype=torch.int32, device="cuda"), wl, p, nw, K, st_gpu)
        wsrc, wdst, wdata = oracle.sample_random_walk(ip, ix, inp, wl, p, nw, K, st_orc)
        m = int(num.item())
        assert m == wsrc.size
        np.testing.assert_array_equal(host_u32(src, m), wsrc)
        np.testing.assert_array_equal(host_u32(dst, m), wdst)
        np.testing.assert_array_equal(host_u32(data, m), wdata)
        got_states = states_np(st_gpu)
        np.testing.assert_array_equal(got_states[:, 0], st_orc["d"])
        np.testing.assert_array_equal(got_states[:, 1:], st_orc["v"])


def test_khop3_properties_full_size(ops):
    """BASELINE-sized layer (88k seeds x fanout 25): distinctness / membership / counts."""
    ip, ix = powerlaw_csr(200_000, mean_deg=40, seed=5, zero_frac=0.01)
    g = ops.DeviceGraph(dev(ip), dev(ix))
    n, fanout = 88_000, 25
    inp = np.random.RandomState(0).permutation(200_000)[:n].astype(np.uint32)
    st = ops.random_states(n, 1)
    src, dst, num = ops.sample_khop3(g, dev(inp), fanout, st)
    m = int(num.item())
    src, dst = host_u32(src, m), host_u32(dst, m)
    deg = (ip[1:] - ip[:-1])[inp].astype(np.int64)
    cnt = np.minimum(deg, fanout)
    assert m == cnt.sum()
    np.testing.assert_array_equal(src, np.repeat(inp, cnt))
    off = np.concatenate([[0], np.cumsum(cnt)])
    for i in np.random.RandomState(1).randint(0, n, 300):
        nb = ix[ip[inp[i]]: ip[inp[i] + 1]]
        got = dst[off[i]: off[i + 1]]
        if deg[i] <= fanout:
            np.testing.assert_array_equal(got, nb)
        else:
            # `fanout` DISTINCT positions of the list: as multisets, got is a sub-multiset of nb
            vals, c_got = np.unique(got, return_counts=True)
            c_nb = np.array([(nb == v).sum() for v in vals])
            assert (c_got <= c_nb).all()


# ------------------------------------------------------------------ cache
@pytest.mark.parametrize("n,ratio", [(0, 0.5), (1, 0.5), (1023, 0.0), (1024, 1.0), (5000, 0.3), (100_000, 0.64)])
def test_get_miss_cache_index(ops, n, ratio):
    N = 50_000
    rng = np.random.RandomState(n + 1)
    rank = rng.permutation(N).astype(np.uint32)
    _, table = oracle.cache_build(rank, int(N * ratio), False)
    nodes = rng.randint(0, N, n).astype(np.uint32)
    t_nodes = dev(nodes) if n else torch.zeros(0, dtype=torch.int32, device="cuda")
    ms, md, nm, hs, hd, nh = ops.get_miss_cache_index(dev(table), t_nodes)
    wms, wmd, whs, whd = oracle.get_miss_cache_index(table, nodes)
    assert int(nm.item()) == wms.size and int(nh.item()) == whs.size
    np.testing.assert_array_equal(host_u32(ms, wms.size), wms)
    np.testing.assert_array_equal(host_u32(md, wms.size), wmd)
    np.testing.assert_array_equal(host_u32(hs, whs.size), whs)
    np.testing.assert_array_equal(host_u32(hd, whs.size), whd)


@pytest.mark.parametrize("P", [1, 2, 3, 8])
@pytest.mark.parametrize("dim", [100, 128])
def test_partition_cache_paths(ops, P, dim):
    """combine_cache_data_for_partition + miss path vs oracle, and the fused one-pass extract."""
    N, n, ratio = 20_000, 7000, 0.4
    rng = np.random.RandomState(P * 100 + dim)
    feat = exact_features(N, dim, np.float32)
    rank = rng.permutation(N).astype(np.uint32)
    num_cached = int(N * ratio)
    rank_s, table = oracle.cache_build(rank, num_cached, True)
    parts = [oracle.partition_feature(feat, rank_s, num_cached, p, P) for p in range(P)]
    nodes = rng.randint(0, N, n).astype(np.uint32)
    wms, wmd, whs, whd = oracle.get_miss_cache_index(table, nodes)
    want = np.zeros((n, dim), np.float32)
    oracle.gather_scatter(want, feat, wms, wmd)
    oracle.gather_scatter_partition(want, parts, whs, whd)
    np.testing.assert_array_equal(want, feat[nodes])  # oracle self-consistency

    t_parts = [dev(p) for p in parts]
    ptab = ops.part_pointer_table(t_parts, "cuda")
    t_feat = dev(feat)  # stands in for the pinned-host tier in this test
    ms, md, nm, hs, hd, nh = ops.get_miss_cache_index(dev(table), dev(nodes))
    out = torch.zeros((n, dim), dtype=torch.float32, device="cuda")
    ops.gather_scatter(out, t_feat, ms, md, num=n, num_dev=nm)
    ops.gather_scatter_partition(out, ptab, P, hs, hd, num=n, num_dev=nh)
    assert out.cpu().numpy().tobytes() == want.tobytes()

    out2 = torch.zeros((n, dim), dtype=torch.float32, device="cuda")
    nmiss = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.extract_cached(out2, dev(nodes), dev(table), ptab, P, t_feat, num_miss=nmiss)
    assert out2.cpu().numpy().tobytes() == want.tobytes()
    assert int(nmiss.item()) == wms.size


def test_tier_counters_over_many_tiles_per_workgroup(ops):
    """The gather counts rows per tier in registers per wave, combines them per workgroup in LDS and adds to the
    caller's counters once per workgroup: with 300 K rows on a 256-workgroup grid every wave sweeps several tiles.
    Counters are ADDED to (two calls = twice the rows); the rows themselves equal the table's."""
    N, dim, n, P, me, R = 50_000, 32, 300_000, 4, 1, 6_000
    rng = np.random.RandomState(3)
    feat = exact_features(N, dim, np.float32)
    rank = rng.permutation(N).astype(np.int64)
    num_cached = 30_000
    table = np.full(N, -1, np.int32)
    table[rank[:num_cached]] = np.arange(num_cached, dtype=np.int32)
    replica = dev(feat[rank[:R]])
    parts = [dev(feat[rank[R + p:num_cached:P]]) for p in range(P)]
    ptab = ops.part_pointer_table(parts, "cuda")
    nodes = rng.randint(0, N, n).astype(np.uint32)
    out = torch.zeros((n, dim), dtype=torch.float32, device="cuda")
    counters = torch.zeros(4, dtype=torch.int64, device="cuda")
    t_feat, t_table, t_nodes = dev(feat), dev(table), dev(nodes)
    for _ in range(2):
        ops.extract_tiered(out, t_nodes, t_table, replica, ptab, P, me, t_feat, tier_rows=counters)
    assert out.cpu().numpy().tobytes() == feat[nodes].tobytes()
    slots = table[nodes]
    shard = (slots - R) % P
    want = [int((slots < 0).sum()), int(((slots >= R) & (shard != me)).sum()), int(((slots >= R) & (shard == me)).sum()),
            int(((slots >= 0) & (slots < R)).sum())]
    assert counters.cpu().tolist() == [2 * w for w in want] and sum(want) == n
    # the plain cached gather's miss counter takes the same route
    nmiss = torch.zeros(1, dtype=torch.int64, device="cuda")
    full = dev(feat[rank[:num_cached]])
    ops.extract_cached(out, t_nodes, t_table, ops.part_pointer_table([full], "cuda"), 0, t_feat, num_miss=nmiss)
    assert int(nmiss.item()) == want[0] and out.cpu().numpy().tobytes() == feat[nodes].tobytes()


def test_fabric_probe_runs_and_updates_the_table(ops):
    """ggms_fabric_probe (bench.py's memory-side ceilings): the three request kinds launch, the atomics are real
    updates (a decreasing salt lowers the words), loads leave the table alone."""
    import ctypes as C
    from xgnn_amd import lib
    words, reqs = 1 << 20, 200_000
    tab = torch.full((words,), -1, dtype=torch.int64, device="cuda")
    sink = torch.zeros(1, dtype=torch.int32, device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib().ggms_fabric_probe(1, C.c_void_p(tab.data_ptr()), words, reqs, 0x7fffff00, C.c_void_p(sink.data_ptr()), s) == 0
    assert bool((tab == -1).all())
    assert lib().ggms_fabric_probe(0, C.c_void_p(tab.data_ptr()), words, reqs, 0x7fffff00, C.c_void_p(sink.data_ptr()), s) == 0
    touched = int((tab != -1).sum().item())
    assert 0.8 * reqs < touched <= reqs  # random keys: a few collisions
    before = tab.cpu().numpy().view(np.uint64).copy()
    assert lib().ggms_fabric_probe(2, C.c_void_p(tab.data_ptr()), words, reqs, 0x7ffffeff, C.c_void_p(sink.data_ptr()), s) == 0
    after = tab.cpu().numpy().view(np.uint64)
    assert (after <= before).all() and (after < before).any()
    assert lib().ggms_fabric_probe(7, C.c_void_p(tab.data_ptr()), words, reqs, 1, C.c_void_p(sink.data_ptr()), s) != 0  # unknown kind


# ------------------------------------------- per-list limits and long lists
@pytest.mark.parametrize("sampler,fanout", [("khop3", 127), ("khop3", 100), ("khop0", 2048), ("khop0", 100), ("khop0", 4000),
                                            ("khop0", 7), ("khop2", 300), ("khop1", 40)])
def test_samplers_on_hub_graph(ops, sampler, fanout):
    """Six 6000-neighbour lists among short ones: khop3 at its fanout limit (127, khop3.cu:85), khop0 at its LDS slot
    limit (2048) and through its heavy-list kernel, khop2 / khop1 with fanouts above most degrees."""
    ip, ix = hub_csr()
    n = 2500
    inp = np.random.RandomState(fanout).permutation(ip.size - 1)[:n].astype(np.uint32)
    inp[:6] = np.argsort(ip[1:] - ip[:-1])[-6:].astype(np.uint32)  # the hubs are in
    inp = np.unique(inp)  # khop2 wants distinct seeds
    n = inp.size
    ix_orc = ix.copy()
    t_ix = dev(ix)
    g = ops.DeviceGraph(dev(ip), t_ix)
    nstates = max(1024, n * fanout if sampler == "khop1" else 0)
    nstates = min(nstates, 512 * 1024)
    st_gpu = ops.random_states(nstates, 31)
    st_orc = oracle.random_states(nstates, 31)
    for rep in range(2):
        if sampler == "khop3":
            got = ops.sample_khop3(g, dev(inp), fanout, st_gpu)
            want = oracle.sample_khop3(ip, ix_orc, inp, fanout, st_orc)
        elif sampler == "khop0":
            got = ops.sample_khop0(g, dev(inp), fanout)
            want = oracle.sample_khop0(ip, ix_orc, inp, fanout)
        elif sampler == "khop2":
            got = ops.sample_khop2(g, dev(inp), fanout, st_gpu)
            want = oracle.sample_khop2(ip, ix_orc, inp, fanout, st_orc)
        else:
            got = ops.sample_khop1(g, dev(inp), fanout, st_gpu)
            want = oracle.sample_khop1(ip, ix_orc, inp, fanout, st_orc)
        m = int(got[2].item())
        assert m == want[0].size
        np.testing.assert_array_equal(host_u32(got[0], m), want[0])
        np.testing.assert_array_equal(host_u32(got[1], m), want[1])
    np.testing.assert_array_equal(host_u32(t_ix), ix_orc)


@pytest.mark.parametrize("fanout", [2049, 3000])
def test_khop0_fanout_beyond_the_lds_slots(ops, fanout):
    """khop0 with more reservoir slots than LDS holds (fanout > 2048; the reference has no bound): the slots live in
    the seed's slice of the output until the neighbours replace them.  Lists shorter than, just above (16-lane
    resolve) and far above (whole-block resolve) the fanout; leaf operator and batch (fused dedup insert)."""
    rng = np.random.RandomState(fanout)
    N = 1500
    deg = rng.choice([0, 9, 2300, 2600, 3100, 3600, 6000], size=N, p=[0.05, 0.55, 0.08, 0.08, 0.08, 0.08, 0.08])
    ip = np.zeros(N + 1, np.uint32)
    ip[1:] = np.cumsum(deg)
    ix = rng.randint(0, N, int(ip[-1])).astype(np.uint32)
    g = ops.DeviceGraph(dev(ip), dev(ix))
    inp = rng.permutation(N)[:400].astype(np.uint32)
    for rep in range(2):
        got = ops.sample_khop0(g, dev(inp), fanout)
        want = oracle.sample_khop0(ip, ix, inp, fanout)
        m = int(got[2].item())
        assert m == want[0].size
        np.testing.assert_array_equal(host_u32(got[0], m), want[0])
        np.testing.assert_array_equal(host_u32(got[1], m), want[1])
    for direct in (True, False):
        bs = ops.BatchSampler(g, [fanout], 40, sample_type=ops.KHOP0, seed=3, direct_table=direct)
        seeds = inp[:40]
        bs.sample(dev(seeds))
        r = bs.result()
        want = oracle.do_sample(oracle.KHOP0, ip, ix, seeds, [fanout], None)
        np.testing.assert_array_equal(host_u32(r["input_nodes"]), want["input_nodes"])
        np.testing.assert_array_equal(host_u32(r["layers"][0]["row"]), want["layers"][0]["row"])
        np.testing.assert_array_equal(host_u32(r["layers"][0]["col"]), want["layers"][0]["col"])


# ------------------------------------------------------- multi-layer batch
@pytest.mark.parametrize("direct", [True, False])
@pytest.mark.parametrize("stype", ["khop3", "khop0", "khop2", "khop1"])
@pytest.mark.parametrize("fanouts,nseed", [([25, 10], 1000), ([5, 10, 15], 300), ([3], 129), ([25, 10], 0)])
def test_sample_batch_vs_oracle(ops, stype, fanouts, nseed, direct):
    """DoGPUSample (dist_loops.cc:62-368): row/col/num_src/num_dst per layer + input nodes."""
    ip, ix = powerlaw_csr(20_000, mean_deg=30, seed=2)
    t_ix = dev(ix)
    g = ops.DeviceGraph(dev(ip), t_ix)
    rng = np.random.RandomState(len(fanouts) * 1000 + nseed)
    code = {"khop3": ops.KHOP3, "khop0": ops.KHOP0, "khop2": ops.KHOP2, "khop1": ops.KHOP1}[stype]
    ocode = {"khop3": oracle.KHOP3, "khop0": oracle.KHOP0, "khop2": oracle.KHOP2, "khop1": oracle.KHOP1}[stype]
    bs = ops.BatchSampler(g, fanouts, max(nseed, 1), sample_type=code, seed=77, direct_table=direct)
    orc_states = oracle.random_states(bs.states.shape[0], 77) if stype != "khop0" else None
    ix = ix.copy()  # khop2 permutes the oracle's CSR too
    for rep in range(3):
        seeds = rng.permutation(20_000)[:nseed].astype(np.uint32)
        if rep == 2 and nseed > 10 and stype != "khop2":  # khop2 needs distinct seeds (both sides race otherwise)
            seeds[5] = seeds[0]  # duplicated seed: local ids of raw seeds go through the table
        t_seeds = dev(seeds) if nseed else torch.zeros(0, dtype=torch.int32, device="cuda")
        bs.sample(t_seeds)
        got = bs.result()
        want = oracle.do_sample(ocode, ip, ix, seeds, fanouts, orc_states)
        np.testing.assert_array_equal(host_u32(got["input_nodes"]), want["input_nodes"])
        for i in range(len(fanouts)):
            gl, wl = got["layers"][i], want["layers"][i]
            assert (gl["num_src"], gl["num_dst"]) == (wl["num_src"], wl["num_dst"]), (rep, i)
            np.testing.assert_array_equal(host_u32(gl["row"]), wl["row"], err_msg=f"row layer {i} rep {rep}")
            np.testing.assert_array_equal(host_u32(gl["col"]), wl["col"], err_msg=f"col layer {i} rep {rep}")
    np.testing.assert_array_equal(host_u32(t_ix), ix)  # untouched, or permuted identically (khop2)


@pytest.mark.parametrize("stype", ["khop3", "khop2", "khop1", "khop0"])
def test_batches_in_flight_keep_batch_order(ops, stype):
    """Three batches in flight on three streams (own table + workspace each) must give exactly what the
    one-at-a-time loop gives: the RNG pool / khop2's CSR are consumed in batch order (rng_wait / rng_done)."""
    ip, ix = powerlaw_csr(20_000, mean_deg=30, seed=4)
    t_ix = dev(ix)
    g = ops.DeviceGraph(dev(ip), t_ix)
    fanouts, nseed, K, NB = [10, 5], 1500, 3, 7
    code = {"khop3": ops.KHOP3, "khop0": ops.KHOP0, "khop2": ops.KHOP2, "khop1": ops.KHOP1}[stype]
    ocode = {"khop3": oracle.KHOP3, "khop0": oracle.KHOP0, "khop2": oracle.KHOP2, "khop1": oracle.KHOP1}[stype]
    bs = ops.BatchSampler(g, fanouts, nseed, sample_type=code, seed=5, num_slots=NB, num_pipelines=K)
    orc_states = oracle.random_states(bs.states.shape[0], 5) if stype != "khop0" else None
    rng = np.random.RandomState(8)
    seeds = [rng.permutation(20_000)[:nseed].astype(np.uint32) for _ in range(NB)]
    t_seeds = [dev(x) for x in seeds]
    streams = [torch.cuda.Stream() for _ in range(K)]
    torch.cuda.synchronize()
    for b in range(NB):
        with torch.cuda.stream(streams[b % K]):
            bs.sample(t_seeds[b], slot=b, copy_input_nodes=True)
    torch.cuda.synchronize()
    ix = ix.copy()
    for b in range(NB):
        want = oracle.do_sample(ocode, ip, ix, seeds[b], fanouts, orc_states)
        c = bs.counts_slots[b].cpu().tolist()
        np.testing.assert_array_equal(host_u32(bs.input_nodes[b], c[3 * len(fanouts)]), want["input_nodes"])
        for i in range(len(fanouts)):
            wl = want["layers"][i]
            assert (c[3 * i], c[3 * i + 1], c[3 * i + 2]) == (wl["row"].size, wl["num_src"], wl["num_dst"]), (b, i)
            np.testing.assert_array_equal(host_u32(bs.rows[b][i], c[3 * i]), wl["row"], err_msg=f"row {b}/{i}")
            np.testing.assert_array_equal(host_u32(bs.cols[b][i], c[3 * i]), wl["col"], err_msg=f"col {b}/{i}")
    np.testing.assert_array_equal(host_u32(t_ix), ix)


@pytest.mark.parametrize("name", ["khop3", "khop0", "khop2", "khop1", "weighted_khop", "random_walk"])
@pytest.mark.parametrize("direct", [True, False])
def test_sampler_golden_vectors_hip(ops, name, direct):
    """The HIP batch sampler against the committed known-answer vectors (tests/golden/sampler_golden.npz), both table
    layouts: COO, input nodes and the RNG pool after the batch."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampler_golden.npz"))
    ip, ix, seeds = g["indptr"], g["indices"], g["seeds"]
    fanouts = [int(x) for x in g[f"{name}:fanouts"]]
    code = {"khop3": ops.KHOP3, "khop0": ops.KHOP0, "khop2": ops.KHOP2, "khop1": ops.KHOP1,
            "weighted_khop": ops.WEIGHTED_KHOP, "random_walk": ops.RANDOM_WALK}[name]
    kw = {}
    if name == "weighted_khop":
        kw = dict(prob_table=dev(g["prob"]), alias_table=dev(g["alias"]))
    if name == "random_walk":
        kw = dict(random_walk_length=3, random_walk_restart_prob=0.5, num_random_walk=4)
    graph = ops.DeviceGraph(dev(ip), dev(ix.copy()))
    s = seeds if name != "khop2" else np.unique(seeds)
    bs = ops.BatchSampler(graph, fanouts, s.size, sample_type=code, seed=int(g["rng_seed"]), direct_table=direct, **kw)
    bs.sample(dev(s))
    got = bs.result()
    assert np.array_equal(host_u32(got["input_nodes"]), g[f"{name}:input_nodes"])
    for i, l in enumerate(got["layers"]):
        assert np.array_equal(host_u32(l["row"]), g[f"{name}:row{i}"]) and np.array_equal(host_u32(l["col"]), g[f"{name}:col{i}"])
        assert [l["num_src"], l["num_dst"]] == [int(x) for x in g[f"{name}:num{i}"]]
        if l["data"] is not None:
            assert np.array_equal(host_u32(l["data"]), g[f"{name}:data{i}"])
    if bs.states is not None:  # the pool may be sized differently; stream t is curand_init(seed + t) either way
        st = states_np(bs.states)
        m = min(st.shape[0], g[f"{name}:states_d"].size)
        assert np.array_equal(st[:m, 0], g[f"{name}:states_d"][:m]) and np.array_equal(st[:m, 1:], g[f"{name}:states_v"][:m])


def test_fused_khop3_random_shapes(ops):
    """The one-launch khop3 layer (offset scan + positions + gather + dedup insert) over a sweep of seeded random shapes:
    fan-outs 1..127 (LDS > 48 KB above 96), frontiers of 1, 127, 128, 129 ... seeds, repeated seeds, graphs with empty
    lists and hubs, 1-3 layers -- leaf call and batch call, each equal to the oracle."""
    rng = np.random.RandomState(2024)
    graphs = [powerlaw_csr(4000, mean_deg=9, seed=1), powerlaw_csr(9000, mean_deg=40, seed=2), hub_csr(3000, 6, 6000, 9, 3)]
    t_graphs = [ops.DeviceGraph(dev(ip), dev(ix)) for ip, ix in graphs]
    for trial in range(36):
        gi = trial % 3
        ip, ix = graphs[gi]
        N = ip.size - 1
        L = int(rng.randint(1, 4))
        fan_hi = 127 if trial % 6 == 0 else 24
        fanouts = [int(rng.randint(1, fan_hi + 1)) for _ in range(L)]
        nseed = int(rng.choice([1, 2, 63, 127, 128, 129, 255, 256, 700, 1500]))
        if np.prod([f + 1 for f in fanouts]) * nseed > 3_000_000:
            fanouts = [min(f, 12) for f in fanouts]
        seeds = rng.randint(0, N, nseed).astype(np.uint32)  # with repeats
        # leaf: GPUSampleKHop3
        st = ops.random_states((nseed + 127) // 128 * 8, 100 + trial)
        src, dst, num = ops.sample_khop3(t_graphs[gi], dev(seeds), fanouts[-1], st)
        ost = oracle.random_states(st.shape[0], 100 + trial)
        wsrc, wdst = oracle.sample_khop3(ip, ix, seeds, fanouts[-1], ost)
        k = int(num.item())
        assert k == wsrc.size, (trial, fanouts, nseed)
        assert np.array_equal(host_u32(src, k), wsrc) and np.array_equal(host_u32(dst, k), wdst), (trial, fanouts, nseed)
        # batch: DoGPUSample
        bs = ops.BatchSampler(t_graphs[gi], fanouts, nseed, sample_type=ops.KHOP3, seed=7 + trial)
        bs.sample(dev(seeds))
        got = bs.result()
        want = oracle.do_sample(oracle.KHOP3, ip, ix, seeds, fanouts, oracle.random_states(bs.states.shape[0], 7 + trial))
        assert np.array_equal(host_u32(got["input_nodes"]), want["input_nodes"]), (trial, fanouts, nseed)
        for i in range(L):
            assert np.array_equal(host_u32(got["layers"][i]["row"]), want["layers"][i]["row"]), (trial, i, fanouts, nseed)
            assert np.array_equal(host_u32(got["layers"][i]["col"]), want["layers"][i]["col"]), (trial, i, fanouts, nseed)
        del bs
    assert ops.device_status() == 0


def test_all_samplers_random_shapes(ops):
    """Batch calls (DoGPUSample) of every other sample type over seeded random shapes, both table layouts: random walks
    from 1 to 300 visits per seed (LDS tiles of 256 / 128 / 64 seeds and the spilling variant), khop0 up to fanout 2500
    (LDS slots and slots in the output), khop2 (distinct seeds), khop1 / alias / prefix with frontiers around the seed
    sort's 8192-pair switch."""
    rng = np.random.RandomState(77)
    graphs = [powerlaw_csr(4000, mean_deg=9, seed=1), powerlaw_csr(30_000, mean_deg=25, seed=2), hub_csr(3000, 6, 6000, 9, 3)]
    kinds = ["random_walk", "khop0", "khop2", "khop1", "weighted", "prefix"]
    for trial in range(48):
        kind = kinds[trial % len(kinds)]
        gi = (trial // len(kinds)) % 3
        ip, ix = graphs[gi]
        N = ip.size - 1
        ix_orc = ix.copy()
        g = ops.DeviceGraph(dev(ip), dev(ix.copy()))
        L = int(rng.randint(1, 4))
        nseed = int(rng.choice([1, 63, 64, 65, 256, 257, 900, 2100]))
        direct = bool(trial % 2)
        kw, okw = {}, {}
        if kind == "random_walk":
            wl, nw = [(3, 4), (1, 1), (5, 6), (4, 12), (10, 12), (15, 9), (2, 150)][int(rng.randint(0, 7))]
            fanouts = [int(rng.randint(1, 9)) for _ in range(L)]
            if wl * nw > 100:
                nseed, fanouts = min(nseed, 257), fanouts[:2]
            p = float(rng.choice([0.0, 0.3, 0.5]))
            kw = dict(random_walk_length=wl, random_walk_restart_prob=p, num_random_walk=nw)
            okw = dict(walk_length=wl, restart_prob=p, num_walk=nw)
            code, ocode = ops.RANDOM_WALK, oracle.RANDOM_WALK
        elif kind == "khop0":
            fanouts = [int(rng.randint(1, 30)) for _ in range(L)]
            if trial % 12 == 1:
                fanouts, nseed = [int(rng.choice([2049, 2500]))], min(nseed, 65)
            code, ocode = ops.KHOP0, oracle.KHOP0
        elif kind == "khop2":
            fanouts = [int(rng.randint(1, 30)) for _ in range(L)]
            code, ocode = ops.KHOP2, oracle.KHOP2
        else:
            fanouts = [int(rng.randint(1, 12)) for _ in range(L)]
            if L >= 2 and gi == 1:
                nseed = 2100  # the deeper frontiers then straddle 8192 seeds
            code, ocode = {"khop1": (ops.KHOP1, oracle.KHOP1), "weighted": (ops.WEIGHTED_KHOP, oracle.WEIGHTED_KHOP),
                           "prefix": (ops.WEIGHTED_KHOP_PREFIX, oracle.WEIGHTED_KHOP_PREFIX)}[kind]
            if kind == "weighted":
                prob = rng.random_sample(ix.size).astype(np.float32)
                alias = rng.randint(0, N, ix.size).astype(np.uint32)
                kw, okw = dict(prob_table=dev(prob), alias_table=dev(alias)), dict(prob=prob, alias=alias)
            if kind == "prefix":
                from xgnn_amd import datagen
                prob = datagen.build_prob_prefix_table(ip, datagen.edge_weights({"indptr": ip, "indices": ix}, seed=trial))
                kw, okw = dict(prob_table=dev(prob)), dict(prob=prob)
        seeds = (rng.permutation(N)[:nseed] if kind == "khop2" else rng.randint(0, N, nseed)).astype(np.uint32)
        nseed = seeds.size
        bs = ops.BatchSampler(g, fanouts, nseed, sample_type=code, seed=300 + trial, direct_table=direct, **kw)
        states = oracle.random_states(bs.states.shape[0], 300 + trial) if bs.states is not None else None
        for rep in range(2):
            bs.sample(dev(seeds))
            got = bs.result()
            want = oracle.do_sample(ocode, ip, ix_orc, seeds, fanouts, states, **okw)
            tag = (trial, kind, fanouts, nseed, direct, kw.get("random_walk_length"), kw.get("num_random_walk"), rep)
            assert np.array_equal(host_u32(got["input_nodes"]), want["input_nodes"]), tag
            for i in range(len(fanouts)):
                gl, wl_ = got["layers"][i], want["layers"][i]
                assert (gl["num_src"], gl["num_dst"]) == (wl_["num_src"], wl_["num_dst"]), tag + (i,)
                assert np.array_equal(host_u32(gl["row"]), wl_["row"]) and np.array_equal(host_u32(gl["col"]), wl_["col"]), tag + (i,)
                if kind == "random_walk":
                    assert np.array_equal(host_u32(gl["data"]), wl_["data"]), tag + (i,)
        del bs
    assert ops.device_status() == 0


def test_sample_batch_refuses_degenerate_shapes(ops):
    """A layer with fanout 0 (khop0's resolver would divide by it) or a random walk with no walks is an argument
    error, not a crash or a spin."""
    ip, ix = powerlaw_csr(2000, mean_deg=9, seed=1)
    g = ops.DeviceGraph(dev(ip), dev(ix))
    seeds = dev(np.arange(50, dtype=np.uint32))
    for code in (ops.KHOP0, ops.KHOP3, ops.KHOP2, ops.KHOP1):
        with pytest.raises(Exception):
            bs = ops.BatchSampler(g, [5, 0], 50, sample_type=code, seed=1)
            bs.sample(seeds)
            bs.result()
    with pytest.raises(Exception):
        bs = ops.BatchSampler(g, [5, 5], 50, sample_type=ops.RANDOM_WALK, seed=1, random_walk_length=3,
                              random_walk_restart_prob=0.5, num_random_walk=0)
        bs.sample(seeds)
        bs.result()


def test_heavy_wait_changes_timing_not_results(ops):
    """ggms_sample_extra_t.heavy_wait: the last layer's sampler launch waits for an event recorded on another stream
    (a feature gather, in the pipeline).  Whatever the event, the batch is the oracle's."""
    ip, ix = powerlaw_csr(20_000, mean_deg=30, seed=6)
    g = ops.DeviceGraph(dev(ip), dev(ix))
    fanouts = [5, 10, 15]
    bs = ops.BatchSampler(g, fanouts, 400, sample_type=ops.KHOP3, seed=9)
    states = oracle.random_states(bs.states.shape[0], 9)
    other = torch.cuda.Stream()
    big = torch.empty(1 << 26, dtype=torch.float32, device="cuda")
    rng = np.random.RandomState(3)
    for rep in range(3):
        seeds = rng.permutation(20_000)[:400].astype(np.uint32)
        with torch.cuda.stream(other):
            big.add_(1.0)  # something for the event to sit behind
            ev = torch.cuda.Event()
            ev.record(other)
        bs.sample(dev(seeds), heavy_wait=ev if rep else None)
        got = bs.result()
        want = oracle.do_sample(oracle.KHOP3, ip, ix, seeds, fanouts, states)
        np.testing.assert_array_equal(host_u32(got["input_nodes"]), want["input_nodes"])
        for i in range(3):
            np.testing.assert_array_equal(host_u32(got["layers"][i]["row"]), want["layers"][i]["row"])
            np.testing.assert_array_equal(host_u32(got["layers"][i]["col"]), want["layers"][i]["col"])


@pytest.mark.parametrize("stype", ["weighted", "random_walk", "random_walk_long"])
def test_sample_batch_weighted_and_random_walk(ops, stype):
    """DoGPUSample with the weighted (alias) sampler and with PinSAGE random walks (row/col/data); _long: 135 visits
    per seed, beyond what the top-K kernel ranks in LDS (its spilling variant)."""
    ip, ix = powerlaw_csr(20_000, mean_deg=30, seed=2)
    N = ip.size - 1
    g = ops.DeviceGraph(dev(ip), dev(ix))
    rng = np.random.RandomState(11)
    if stype == "weighted":
        fanouts = [10, 5]
        prob = rng.random_sample(ix.size).astype(np.float32)
        alias = rng.randint(0, N, ix.size).astype(np.uint32)
        t_prob, t_alias = dev(prob), dev(alias)
        bs = ops.BatchSampler(g, fanouts, 500, sample_type=ops.WEIGHTED_KHOP, seed=5, prob_table=t_prob,
                              alias_table=t_alias)
        kw = dict(prob=prob, alias=alias)
        code = oracle.WEIGHTED_KHOP
    else:
        wl_, p_, nw_ = (3, 0.5, 4) if stype == "random_walk" else (15, 0.05, 9)
        fanouts = [5, 5, 5] if stype == "random_walk" else [6, 40]  # num_neighbor per layer
        bs = ops.BatchSampler(g, fanouts, 500, sample_type=ops.RANDOM_WALK, seed=5, random_walk_length=wl_,
                              random_walk_restart_prob=p_, num_random_walk=nw_)
        kw = dict(walk_length=wl_, restart_prob=p_, num_walk=nw_)
        code = oracle.RANDOM_WALK
    orc_states = oracle.random_states(bs.states.shape[0], 5)
    for rep in range(2):
        seeds = rng.permutation(N)[:500].astype(np.uint32)
        bs.sample(dev(seeds))
        got = bs.result()
        want = oracle.do_sample(code, ip, ix, seeds, fanouts, orc_states, **kw)
        np.testing.assert_array_equal(host_u32(got["input_nodes"]), want["input_nodes"])
        for i in range(len(fanouts)):
            gl, wl = got["layers"][i], want["layers"][i]
            assert (gl["num_src"], gl["num_dst"]) == (wl["num_src"], wl["num_dst"]), (rep, i)
            np.testing.assert_array_equal(host_u32(gl["row"]), wl["row"])
            np.testing.assert_array_equal(host_u32(gl["col"]), wl["col"])
            if stype != "weighted":
                np.testing.assert_array_equal(host_u32(gl["data"]), wl["data"])


def test_full_size_batch_properties(ops):
    """BASELINE configs[1] size (products-shaped, batch 8000, fanout [25,10]): size-independent properties
    instead of an oracle replay -- determinism, hashed == direct table, structural validity of the COO."""
    from xgnn_amd import datagen
    g = datagen.make_graph("products", seed=42)
    ip, ix = g["indptr"], g["indices"]
    N = ip.size - 1
    graph = ops.DeviceGraph(dev(ip), dev(ix))
    seeds = g["train_set"][:8000]
    results = []
    for direct in (True, False, True):
        bs = ops.BatchSampler(graph, [25, 10], 8000, sample_type=ops.KHOP3, seed=0x5EED, direct_table=direct)
        bs.sample(dev(seeds))
        r = bs.result()
        results.append(dict(inp=host_u32(r["input_nodes"]).copy(),
                            layers=[(host_u32(l["row"]).copy(), host_u32(l["col"]).copy(), l["num_src"], l["num_dst"])
                                    for l in r["layers"]]))
    a = results[0]
    for b in results[1:]:  # same seeds + same RNG seed: bit-identical, whatever the table layout
        np.testing.assert_array_equal(a["inp"], b["inp"])
        for la, lb in zip(a["layers"], b["layers"]):
            np.testing.assert_array_equal(la[0], lb[0])
            np.testing.assert_array_equal(la[1], lb[1])
            assert la[2:] == lb[2:]
    inp = a["inp"]
    assert np.unique(inp).size == inp.size                      # dedup: no node twice
    np.testing.assert_array_equal(inp[:8000], seeds)            # seeds keep local ids 0..7999 (prefix stability)
    deg = (ip[1:].astype(np.int64) - ip[:-1].astype(np.int64))
    (row1, col1, nsrc1, ndst1), (row0, col0, nsrc0, ndst0) = a["layers"][1], a["layers"][0]
    assert ndst1 == 8000 and ndst0 == nsrc1 and nsrc0 == inp.size and nsrc1 <= nsrc0
    for row, col, nsrc, ndst in a["layers"]:
        assert row.max() < nsrc and col.max() < ndst
        # per seed exactly min(deg, fanout) edges, seeds in order
        assert (np.diff(col.astype(np.int64)) >= 0).all()
    cnt1 = np.bincount(col1, minlength=ndst1)
    np.testing.assert_array_equal(cnt1, np.minimum(deg[inp[:ndst1]], 10))
    cnt0 = np.bincount(col0, minlength=ndst0)
    np.testing.assert_array_equal(cnt0, np.minimum(deg[inp[:ndst0]], 25))
    # every sampled edge exists in the CSR: neighbour inp[row] is in the list of seed inp[col] (sampled check)
    rng = np.random.RandomState(0)
    for e in rng.randint(0, row0.size, 2000):
        s, d = inp[col0[e]], inp[row0[e]]
        assert d in ix[ip[s]:ip[s + 1]]
    # first occurrence order: local ids of new nodes increase with the position of their first edge
    first_pos = np.full(nsrc0, row0.size, np.int64)
    np.minimum.at(first_pos, row0, np.arange(row0.size))
    new = np.arange(ndst0, nsrc0)
    assert (np.diff(first_pos[new]) > 0).all()
