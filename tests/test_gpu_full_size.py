"""BASELINE.json configs at their REAL sizes on the GPU box (the toy-size twins live in test_gpu_parity.py).

configs[2]  papers100M-shaped CSR (N 111,059,956, E 1.62e9), 3-hop GCN fanout [5,10,15], batch 8000, every feature
            row (dim 128 f32, 56.9 GB) in pinned host DRAM, gathered zero-copy (cache_ratio 0) -- and the same batch
            from the HBM tier (cache_ratio 1.0, rows in node order);
configs[4]  Friendster-scale CSR (N 65,608,366, E 1.8e9, dim 256 f32), PinSAGE random walk (length 3, restart 0.5,
            4 walks, top-5, 3 layers), hybrid store: the hotter half of the rows (degree rank) in HBM, all of them in
            pinned host DRAM.

Each batch is compared with the CPU oracle (the C restatement finishes a batch in seconds even at this size) AND held to
the size-independent properties of the domain: determinism, hashed == direct dedup table, COO validity, min(deg, fanout)
edges per seed, sampled edges exist in the CSR, first-occurrence numbering, gathered rows equal the generator's closed
form, hits + misses = rows.
"""
import numpy as np
import pytest

import oracle

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

BATCH = 8000


def dev(a):
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda()


def u32(t, n=None):
    a = t.cpu().numpy()
    a = a[:n] if n is not None else a
    return a.view(np.uint32) if a.dtype == np.int32 else a


def feat_rows(node_ids, out, dim):
    """feat[i, j] = float((i*dim + j) & 0xFFFF): exactly representable, so gathers compare bit for bit."""
    cols = torch.arange(dim, dtype=torch.int64, device=out.device)
    step = 1 << 21
    for lo in range(0, node_ids.numel(), step):
        ids = node_ids[lo:lo + step].to(out.device, torch.int64)
        out[lo:lo + step] = ((ids[:, None] * dim + cols[None, :]) & 0xFFFF).to(torch.float32)


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (run with -m gpu on an MI355X box)")
    from xgnn_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def papers():
    """The papers100M-shaped CSR (N 111,059,956, E 1.62e9), generated once for the module."""
    from xgnn_amd import datagen
    return datagen.make_graph("papers100M", seed=42)


_ORACLE = {}


def papers_oracle(papers, num_states):
    """The oracle's GCN [5,10,15] batch of the first 8000 train nodes on the plain CSR (cached: a few seconds of CPU)."""
    if num_states not in _ORACLE:
        st = oracle.random_states(num_states, 0x5EED)
        _ORACLE[num_states] = (oracle.do_sample(oracle.KHOP3, papers["indptr"], papers["indices"],
                                                papers["train_set"][:BATCH], [5, 10, 15], st), st)
    return _ORACLE[num_states]


def coo_properties(ip, ix, seeds, fanouts, inp, layers):
    """layers[i] = (row, col, num_src, num_dst) as host arrays; khop samplers (exactly min(deg, fanout) edges per seed)."""
    L = len(fanouts)
    deg = ip[1:].astype(np.int64) - ip[:-1].astype(np.int64)
    assert np.unique(inp).size == inp.size and np.array_equal(inp[:seeds.size], seeds)
    assert layers[L - 1][3] == seeds.size and layers[0][2] == inp.size
    for i in range(L):
        row, col, nsrc, ndst = layers[i]
        assert row.max() < nsrc and col.max() < ndst and (np.diff(col.astype(np.int64)) >= 0).all()
        if i + 1 < L:
            assert layers[i + 1][2] == ndst
        assert np.array_equal(np.bincount(col, minlength=ndst), np.minimum(deg[inp[:ndst]], fanouts[i]))
        for e in np.random.RandomState(i).randint(0, row.size, 1000):  # sampled edges exist in the CSR
            s, d = inp[col[e]], inp[row[e]]
            assert d in ix[ip[s]:ip[s + 1]]
        first_pos = np.full(nsrc, row.size, np.int64)  # new ids are numbered in first-occurrence order
        np.minimum.at(first_pos, row, np.arange(row.size))
        assert (np.diff(first_pos[np.arange(ndst, nsrc)]) > 0).all()


def test_papers100m_hub_skewed_neighbours_oracle_exact(ops):
    """north_star's "degree-skewed neighbour selection": the papers100M-shaped CSR with every neighbour drawn with
    probability proportional to its degree (datagen.make_graph neighbour_skew = 1.0 -- hubs turn up in thousands of
    lists and hundreds of times in one frontier, so the same dedup words are hit by many lanes at once).  One full
    batch, GCN [5,10,15], both table layouts: COO, input nodes and RNG pool equal to the oracle's, bit for bit."""
    from xgnn_amd import datagen
    g = datagen.make_graph("papers100M", seed=42, neighbour_skew=1.0)
    ip, ix = g["indptr"], g["indices"]
    hits = np.bincount(ix[: 50_000_000], minlength=ip.size - 1)
    assert hits.max() > 2000  # a hub: ~0.02 % of all edge slots point at the heaviest node (uniform: a handful)
    graph = ops.DeviceGraph(dev(ip), dev(ix))
    fanouts, L = [5, 10, 15], 3
    seeds = g["train_set"][:BATCH]
    runs = []
    for direct in (True, False):
        bs = ops.BatchSampler(graph, fanouts, BATCH, sample_type=ops.KHOP3, seed=0x5EED, direct_table=direct)
        bs.sample(dev(seeds))
        r = bs.result()
        runs.append(dict(inp=u32(r["input_nodes"]).copy(),
                         layers=[(u32(l["row"]).copy(), u32(l["col"]).copy(), l["num_src"], l["num_dst"]) for l in r["layers"]],
                         states=bs.states.cpu().numpy().copy()))
        num_states = bs.states.shape[0]
        del bs, r
        torch.cuda.empty_cache()
    a, b = runs
    assert np.array_equal(a["inp"], b["inp"]) and np.array_equal(a["states"], b["states"])
    orc_states = oracle.random_states(num_states, 0x5EED)
    want = oracle.do_sample(oracle.KHOP3, ip, ix, seeds, fanouts, orc_states)
    assert np.array_equal(a["inp"], want["input_nodes"])
    edges = 0
    for i in range(L):
        for run in runs:
            assert np.array_equal(run["layers"][i][0], want["layers"][i]["row"])
            assert np.array_equal(run["layers"][i][1], want["layers"][i]["col"])
            assert run["layers"][i][2:] == (want["layers"][i]["num_src"], want["layers"][i]["num_dst"])
        edges += want["layers"][i]["row"].size
    got_states = a["states"].view(np.uint32)
    assert np.array_equal(got_states[:, 0], orc_states["d"]) and np.array_equal(got_states[:, 1:], orc_states["v"])
    # the skew shows: far more duplicates than the 12 % of the uniform graph
    assert a["inp"].size < 0.85 * edges  # (uniform neighbours: 0.88)
    coo_properties(ip, ix, seeds, fanouts, a["inp"], a["layers"])


@pytest.mark.parametrize("P,fraction", [(2, 0.5), (8, 0.5), (8, 1.0)])
def test_papers100m_sharded_topology_oracle_exact(ops, papers, P, fraction):
    """XGNN mode's graph (arch6 + use_dist_graph, /root/reference README.md:184; DeviceDistGraph, cuda/dist_graph.h:114-158)
    at papers100M size: the leading nodes that hold `fraction` of the edges live in P topology shards (node v in shard
    v % P at row v / P -- P LOGICAL shards in this one process, every one of them in HBM), every other node is read from
    the whole CSR in hipHostRegister'ed host memory (the last slot, over PCIe).  One GCN [5,10,15] batch sampled through
    that view is the oracle's batch on the plain CSR: COO, input nodes, RNG pool, bit for bit."""
    from xgnn_amd import ggms_store
    ip, ix = papers["indptr"], papers["indices"]
    N = ip.size - 1
    ncn = ggms_store.num_cache_node_for(ip, fraction)
    assert (ncn == N) if fraction >= 1.0 else (0.4 * N < ncn < 0.6 * N)  # the host slot is hit by about half the seeds
    t_ip, t_ix = dev(ip), dev(ix)
    pip, pix = ggms_store.topology_shards(t_ip, t_ix, P, ncn)
    del t_ip, t_ix
    torch.cuda.empty_cache()
    # shard p, row r == node p + r P: spot-check the builder against the CSR itself
    for p_ in (0, P - 1):
        sp, sx = u32(pip[p_]), None
        for r in (0, 1, (ncn - 1 - p_) // P):
            v = p_ + r * P
            lo, hi = int(sp[r]), int(sp[r + 1])
            assert np.array_equal(u32(pix[p_][lo:hi]), ix[ip[v]:ip[v + 1]])
    host_ip, host_ix = ops.RegisteredHost(ip), ops.RegisteredHost(ix)  # slot P: the whole CSR, host memory, zero-copy
    try:
        # P = 2: the reference's layout (list heads AND lists of slot P on the host); P = 8: the engine's (slot P's
        # indptr in HBM, its neighbour lists on the host) -- the layout of slot P is the caller's choice
        slot_ip = host_ip.tensor if P == 2 else dev(ip)
        graph = ops.DeviceGraph(None, None, part_indptr=pip + [slot_ip], part_indices=pix + [host_ix.tensor],
                                num_cache_node=ncn)
        fanouts, L = [5, 10, 15], 3
        seeds = papers["train_set"][:BATCH]
        bs = ops.BatchSampler(graph, fanouts, BATCH, sample_type=ops.KHOP3, seed=0x5EED)
        bs.sample(dev(seeds), distinct=True)
        r = bs.result()
        want, orc_states = papers_oracle(papers, bs.states.shape[0])
        assert np.array_equal(u32(r["input_nodes"]), want["input_nodes"])
        for i in range(L):
            assert np.array_equal(u32(r["layers"][i]["row"]), want["layers"][i]["row"])
            assert np.array_equal(u32(r["layers"][i]["col"]), want["layers"][i]["col"])
            assert (r["layers"][i]["num_src"], r["layers"][i]["num_dst"]) == (want["layers"][i]["num_src"], want["layers"][i]["num_dst"])
        got_states = bs.states.cpu().numpy().view(np.uint32)
        assert np.array_equal(got_states[:, 0], orc_states["d"]) and np.array_equal(got_states[:, 1:], orc_states["v"])
        # khop0 reads whole neighbour lists through the same view (leaf call, one layer of the batch's frontier)
        front = want["input_nodes"][:20_000]
        src, dst, num = ops.sample_khop0(graph, dev(front), 10)
        ws, wd = oracle.sample_khop0(ip, ix, front, 10)
        n = int(num.item())
        assert n == ws.size and np.array_equal(u32(src, n), ws) and np.array_equal(u32(dst, n), wd)
        assert ops.device_status() == 0
        del bs, graph
    finally:
        torch.cuda.synchronize()
        host_ip.close()
        host_ix.close()


def test_papers100m_gcn_host_tier_and_hbm_tier(ops, papers):
    g = papers
    ip, ix, meta = g["indptr"], g["indices"], g["meta"]
    N, dim = meta["num_node"], meta["feat_dim"]
    assert N == 111_059_956 and dim == 128
    graph = ops.DeviceGraph(dev(ip), dev(ix))
    fanouts, L = [5, 10, 15], 3
    seeds = g["train_set"][:BATCH]
    runs = []
    for direct, distinct in ((True, False), (False, False), (True, True)):  # the third: seeds promised distinct
        bs = ops.BatchSampler(graph, fanouts, BATCH, sample_type=ops.KHOP3, seed=0x5EED, direct_table=direct)
        bs.sample(dev(seeds), distinct=distinct)
        r = bs.result()
        runs.append(dict(inp=u32(r["input_nodes"]).copy(),
                         layers=[(u32(l["row"]).copy(), u32(l["col"]).copy(), l["num_src"], l["num_dst"]) for l in r["layers"]],
                         states=bs.states.cpu().numpy().copy()))
        num_states = bs.states.shape[0]
        del bs, r
        torch.cuda.empty_cache()
    a = runs[0]
    for b in runs[1:]:  # deterministic; the hashed (reference-sized) table numbers exactly like the direct one
        assert np.array_equal(a["inp"], b["inp"]) and np.array_equal(a["states"], b["states"])
        for la, lb in zip(a["layers"], b["layers"]):
            assert np.array_equal(la[0], lb[0]) and np.array_equal(la[1], lb[1]) and la[2:] == lb[2:]
    coo_properties(ip, ix, seeds, fanouts, a["inp"], a["layers"])
    # the oracle on the same batch: bit-exact COO, input nodes and RNG pool
    want, orc_states = papers_oracle(papers, num_states)
    assert np.array_equal(a["inp"], want["input_nodes"])
    for i in range(L):
        assert np.array_equal(a["layers"][i][0], want["layers"][i]["row"])
        assert np.array_equal(a["layers"][i][1], want["layers"][i]["col"])
        assert a["layers"][i][2:] == (want["layers"][i]["num_src"], want["layers"][i]["num_dst"])
    got_states = a["states"].view(np.uint32)
    assert np.array_equal(got_states[:, 0], orc_states["d"]) and np.array_equal(got_states[:, 1:], orc_states["v"])

    # ---- features: HBM tier (cache_ratio 1.0, node order, no table) and host tier (cache_ratio 0, pinned, zero-copy)
    inp = dev(a["inp"])
    n = inp.numel()
    cache = torch.empty((N, dim), dtype=torch.float32, device="cuda")  # 56.9 GB of the GPU's 288
    feat_rows(torch.arange(N, dtype=torch.int64, device="cuda"), cache, dim)
    want_rows = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    feat_rows(inp.to(torch.int64), want_rows, dim)
    out = torch.zeros((n + 8, dim), dtype=torch.float32, device="cuda")
    ptab = ops.part_pointer_table([cache], torch.device("cuda"))
    miss = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.extract_cached(out, inp, None, ptab, 0, None, num=n, num_miss=miss)
    assert torch.equal(out[:n], want_rows) and not bool(out[n:].any()) and int(miss.item()) == 0
    host = torch.empty((N, dim), dtype=torch.float32, pin_memory=True)  # BASELINE configs[2]: 56.9 GB pinned
    host.copy_(cache)
    del cache
    torch.cuda.empty_cache()
    out.zero_()
    ops.gather_scatter(out, host, inp, None, num=n)  # DoGPUFeatureExtract from host memory, dist_loops.cc:585-634
    assert torch.equal(out[:n], want_rows) and not bool(out[n:].any())
    # idempotence + a checksum of checksums against the CPU extract of the same pinned table
    again = torch.empty_like(out)
    ops.gather_scatter(again, host, inp, None, num=n)
    assert torch.equal(again[:n], out[:n])
    sub = a["inp"][:: max(1, n // 20000)]
    assert ops.extract(host, dev(sub)).cpu().numpy().tobytes() == oracle.extract(host.numpy(), sub).tobytes()


def test_friendster_pinsage_hybrid_store(ops):
    from xgnn_amd import datagen
    g = datagen.make_graph("friendster", seed=42)
    ip, ix, meta = g["indptr"], g["indices"], g["meta"]
    N, dim = meta["num_node"], meta["feat_dim"]
    assert N == 65_608_366 and dim == 256
    graph = ops.DeviceGraph(dev(ip), dev(ix))
    fanouts, L = [5, 5, 5], 3  # num_neighbor per layer (sgnn/train_pinsage.py:138-142)
    kw = dict(random_walk_length=3, random_walk_restart_prob=0.5, num_random_walk=4)
    seeds = g["train_set"][:BATCH]
    runs = []
    for direct in (True, False):
        bs = ops.BatchSampler(graph, fanouts, BATCH, sample_type=ops.RANDOM_WALK, seed=0x5EED, direct_table=direct, **kw)
        bs.sample(dev(seeds))
        r = bs.result()
        runs.append(dict(inp=u32(r["input_nodes"]).copy(),
                         layers=[(u32(l["row"]).copy(), u32(l["col"]).copy(), u32(l["data"]).copy(), l["num_src"], l["num_dst"])
                                 for l in r["layers"]]))
        num_states = bs.states.shape[0]
        del bs, r
        torch.cuda.empty_cache()
    a, b = runs
    assert np.array_equal(a["inp"], b["inp"])
    for la, lb in zip(a["layers"], b["layers"]):
        assert all(np.array_equal(x, y) for x, y in zip(la[:3], lb[:3])) and la[3:] == lb[3:]
    inp = a["inp"]
    assert np.unique(inp).size == inp.size and np.array_equal(inp[:BATCH], seeds)
    for i in range(L):  # top-K structure: <= K per seed, seeds ascending, counts descending within a seed, 1 <= count <= walks*len
        row, col, data, nsrc, ndst = a["layers"][i]
        assert row.max() < nsrc and col.max() < ndst and (np.diff(col.astype(np.int64)) >= 0).all()
        assert np.bincount(col, minlength=ndst).max() <= fanouts[i]
        assert data.min() >= 1 and data.max() <= 12
        same = col[1:] == col[:-1]
        assert (data[1:][same] <= data[:-1][same]).all()
    want = oracle.do_sample(oracle.RANDOM_WALK, ip, ix, seeds, fanouts, oracle.random_states(num_states, 0x5EED),
                            walk_length=3, restart_prob=0.5, num_walk=4)
    assert np.array_equal(inp, want["input_nodes"])
    for i in range(L):
        for k, name in enumerate(("row", "col", "data")):
            assert np.array_equal(a["layers"][i][k], want["layers"][i][name]), (i, name)

    # ---- hybrid store: the hotter half (degree rank) in HBM, every row in pinned host DRAM; one fused gather
    rank = datagen.degree_rank(ip)
    num_cached = N // 2
    t_rank = dev(rank)
    cache = torch.empty((num_cached, dim), dtype=torch.float32, device="cuda")  # 33.6 GB
    feat_rows(t_rank[:num_cached], cache, dim)
    table = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    table[t_rank[:num_cached].long()] = torch.arange(num_cached, dtype=torch.int32, device="cuda")
    host = torch.empty((N, dim), dtype=torch.float32, pin_memory=True)  # 67.2 GB pinned
    step = 1 << 22
    tmp = torch.empty((step, dim), dtype=torch.float32, device="cuda")
    for lo in range(0, N, step):
        m = min(step, N - lo)
        feat_rows(torch.arange(lo, lo + m, dtype=torch.int64, device="cuda"), tmp[:m], dim)
        host[lo:lo + m].copy_(tmp[:m])
    del tmp
    t_inp = dev(inp)
    n = t_inp.numel()
    want_rows = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    feat_rows(t_inp.to(torch.int64), want_rows, dim)
    out = torch.zeros((n + 8, dim), dtype=torch.float32, device="cuda")
    miss = torch.zeros(1, dtype=torch.int64, device="cuda")
    ptab = ops.part_pointer_table([cache], torch.device("cuda"))
    ops.extract_cached(out, t_inp, table, ptab, 0, host, num=n, num_miss=miss)
    assert torch.equal(out[:n], want_rows) and not bool(out[n:].any())
    n_miss = int((table[t_inp.long()] == -1).sum().item())
    assert int(miss.item()) == n_miss and 0 < n_miss < n
    # the split form (GetMissCacheIndex) agrees with the fused one: hits + misses = rows, stable order
    ms, md, nm, cs, cd, nc = ops.get_miss_cache_index(table, t_inp)
    assert int(nm.item()) == n_miss and int(nm.item()) + int(nc.item()) == n
