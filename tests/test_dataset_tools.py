"""Dataset tools on the host (no GPU): the weighted samplers' per-edge tables and the OpenMP hash table of the CPU leg.

* xgnn_amd.datagen.build_alias_tables / build_prob_prefix_table (native host entry points of the library) against the
  oracle's restatement of utility/data-process/toolkit/weight/create_alias_table.cc:105-170 and
  create_prob_prefix_table.cc:94-123, bit for bit, and against what an alias table MEANS: the probability it gives each
  neighbour equals the normalised weight.
* oracle.CpuHashTable2 (cpu/cpu_hashtable2.cc:35-191 with its OpenMP loops): one thread == the serial table; several
  threads keep the set, the prefix stability and the id <-> position contract (the reference's own gtest properties,
  samgraph/unittest/test_hashmap.cc:84-276).
"""
import os

import numpy as np
import pytest

import oracle
from graphgen import powerlaw_csr
from xgnn_amd import datagen


@pytest.mark.parametrize("policy", ["default", "inverse_src_degree", "src_suffix"])
def test_alias_and_prefix_tables_match_the_oracle(policy):
    g = datagen.make_graph("tiny", seed=3)
    ip, ix = g["indptr"], g["indices"]
    w = datagen.edge_weights(g, policy, seed=1)
    prob, alias = datagen.build_alias_tables(ip, ix, w, num_threads=3)
    prob_o, alias_o = oracle.create_alias_table(ip, ix, w)
    assert prob.tobytes() == prob_o.tobytes() and np.array_equal(alias, alias_o)
    assert np.array_equal(datagen.build_alias_tables(ip, ix, w, num_threads=1)[0], prob)  # thread count does not matter
    assert datagen.build_prob_prefix_table(ip, w).tobytes() == oracle.create_prob_prefix_table(ip, w).tobytes()
    assert prob.min() > 0.0 and prob.max() <= 1.0


def test_alias_table_reproduces_the_weights():
    """P(neighbour i of a list) under the alias method = (prob[i] + sum over slots j that alias to i of (1 - prob[j])) / len
    must equal w_i / sum(w).  Lists with distinct neighbours, so that a node id names one slot."""
    rng = np.random.RandomState(5)
    lens = rng.randint(0, 40, 500)
    ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    ix = np.concatenate([rng.permutation(10_000)[:n] for n in lens] + [np.zeros(0, np.int64)]).astype(np.uint32)
    w = (rng.random_sample(ix.size) * 9 + 0.5).astype(np.float32)
    prob, alias = datagen.build_alias_tables(ip, ix, w)
    for v in range(lens.size):
        lo, hi = int(ip[v]), int(ip[v + 1])
        if hi == lo:
            continue
        p = prob[lo:hi].astype(np.float64).copy()
        got = p.copy()
        pos = {int(n): k for k, n in enumerate(ix[lo:hi])}
        for j in range(hi - lo):
            if p[j] < 1.0:
                got[pos[int(alias[lo + j])]] += 1.0 - p[j]
        want = w[lo:hi].astype(np.float64) / w[lo:hi].astype(np.float64).sum()
        np.testing.assert_allclose(got / (hi - lo), want, atol=2e-6)
    # inverse-CDF table: the last entry of a list is its weight sum, entries never decrease
    pre = datagen.build_prob_prefix_table(ip, w)
    for v in range(lens.size):
        lo, hi = int(ip[v]), int(ip[v + 1])
        if hi > lo:
            assert (np.diff(pre[lo:hi]) > 0).all() and abs(pre[hi - 1] - w[lo:hi].sum(dtype=np.float64)) < 1e-3


def test_write_dataset_writes_the_weight_tables(tmp_path):
    g = datagen.make_graph(dict(num_node=3000, mean_deg=8.0, alpha=0.7, dmax=200, feat_dim=4, num_class=3, num_train=100))
    w = datagen.edge_weights(g, "default", seed=2)
    d = datagen.write_dataset(str(tmp_path / "ds"), g, weights=w)
    E = g["indices"].size
    prob = np.fromfile(os.path.join(d, "prob_table.bin"), np.float32)
    alias = np.fromfile(os.path.join(d, "alias_table.bin"), np.uint32)
    pre = np.fromfile(os.path.join(d, "prob_prefix_table.bin"), np.float32)
    assert prob.size == alias.size == pre.size == E
    po, ao = oracle.create_alias_table(g["indptr"], g["indices"], w)
    assert prob.tobytes() == po.tobytes() and np.array_equal(alias, ao)


@pytest.mark.parametrize("threads", [1, 4])
def test_cpu_hashtable2_openmp_port(threads):
    rng = np.random.RandomState(7)
    N = 200_000
    ref = oracle.HashTable(N, 700_000)
    ht = oracle.CpuHashTable2(N, threads)
    fills = [rng.randint(0, N, n).astype(np.uint32) for n in (30_000, 200_000, 5, 300_000)]
    prev = np.zeros(0, np.uint32)
    for f in fills:
        n = ht.fill_with_duplicates(f)
        assert n == ref.fill_with_duplicates(f)
        u = ht.unique()
        assert np.unique(u).size == u.size and set(u.tolist()) == set(ref.unique().tolist())
        assert np.array_equal(u[:prev.size], prev)  # prefix stability across fills (test_hashmap.cc:113-114)
        if threads == 1:
            assert np.array_equal(u, ref.unique())  # one thread: first occurrence wins, exactly the serial table
        ns, nd = ht.map_edges(f, f[::-1].copy())
        assert np.array_equal(u[ns], f) and np.array_equal(u[nd], f[::-1])  # SearchO2N(unique[i]).local == i
        prev = u
    ht.reset()
    assert ht.fill_with_duplicates(fills[0]) == np.unique(fills[0]).size


def test_graph_generator_is_thread_independent_and_skew_is_degree_proportional():
    """make_graph fills its edge chunks on a thread pool: the result must not depend on the number of threads (every
    chunk has its own seed), the default (uniform neighbour ids) must not change when the skewed variant
    is added, and neighbour_skew = 1 must draw neighbours in proportion to their degree."""
    from xgnn_amd import datagen
    preset = dict(num_node=60_000, mean_deg=20.0, alpha=0.75, dmax=3_000, feat_dim=4, num_class=3, num_train=100)
    a = datagen.make_graph(preset, seed=42, chunk=1 << 16, threads=1)
    b = datagen.make_graph(preset, seed=42, chunk=1 << 16, threads=4)
    assert np.array_equal(a["indices"], b["indices"]) and np.array_equal(a["indptr"], b["indptr"])
    s1 = datagen.make_graph(preset, seed=42, chunk=1 << 16, threads=1, neighbour_skew=1.0)
    s4 = datagen.make_graph(preset, seed=42, chunk=1 << 16, threads=4, neighbour_skew=1.0)
    assert np.array_equal(s1["indices"], s4["indices"]) and np.array_equal(s1["indptr"], a["indptr"])
    deg = np.diff(a["indptr"].astype(np.int64))
    hits_uniform = np.bincount(a["indices"], minlength=deg.size)
    hits_skewed = np.bincount(s1["indices"], minlength=deg.size)
    assert np.corrcoef(hits_skewed, deg)[0, 1] > 0.95 and abs(np.corrcoef(hits_uniform, deg)[0, 1]) < 0.05
    half = datagen.make_graph(preset, seed=42, chunk=1 << 16, neighbour_skew=0.5)
    changed = (half["indices"] != a["indices"]).mean()
    assert 0.4 < changed < 0.6  # every neighbour is a degree-proportional pick with probability p, else the uniform one


def test_num_cache_node_rule_matches_the_oracle():
    """use_dist_graph = fraction of the EDGES on the GPUs (dist_engine.cc:225, dist_graph.cu:318-325): the host-side
    rule of xgnn_amd.ggms_store against the oracle's restatement, edge cases included."""
    import oracle
    from graphgen import powerlaw_csr
    from xgnn_amd import ggms_store
    ip, _ = powerlaw_csr(5000, mean_deg=12, seed=3)
    for f in (0.0, 1e-9, 0.1, 0.5, 0.64, 0.999, 1.0):
        assert ggms_store.num_cache_node_for(ip, f) == oracle.num_cache_node(ip, f), f


def test_launch_constant_divisor_arithmetic():
    """ggms_device.h Divisor: v / d and v % d as one multiply-high by floor(2^32 / d) (d = 1: 2^32 - 1) and ONE fix-up.
    The same arithmetic in exact integers, for every shard count the kernels carry and a spread of others, on the edges
    of the 32-bit range and on random ids: the claim "q' is q or q - 1" that the kernels rely on."""
    import numpy as np
    rng = np.random.RandomState(0)
    v = np.concatenate([np.arange(0, 4096, dtype=np.uint64), (1 << 32) - 1 - np.arange(0, 4096, dtype=np.uint64),
                        rng.randint(0, 1 << 32, size=2_000_000, dtype=np.uint64)])
    for d in list(range(1, 65)) + [100, 1000, 65535, 65536, 65537, (1 << 31) - 1, 1 << 31, (1 << 32) - 1]:
        magic = np.uint64(0xFFFFFFFF if d <= 1 else (1 << 32) // d)
        q = (v * magic) >> np.uint64(32)
        r = v - q * np.uint64(d)
        assert (r < np.uint64(2 * d)).all(), d          # one fix-up is enough
        up = r >= np.uint64(d)
        q, r = q + up.astype(np.uint64), r - up.astype(np.uint64) * np.uint64(d)
        assert np.array_equal(q, v // np.uint64(d)) and np.array_equal(r, v % np.uint64(d)), d
