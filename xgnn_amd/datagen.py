"""Seeded synthetic datasets shaped like the reference's.

There is no network, so ogbn-products / papers100M cannot be downloaded; these
generators produce power-law in-neighbour CSRs with the same node count, mean
degree, feature width and train-set size, plus exactly representable features
so gathers can be checked bit for bit.  numpy only (host side).
"""
import numpy as np

PRESETS = {
    # name: num_node, mean_deg, alpha, dmax, feat_dim, num_class, num_train
    "products": dict(num_node=2_449_029, mean_deg=50.5, alpha=0.75, dmax=17_000, feat_dim=100, num_class=47,
                     num_train=196_615),
    "papers100M": dict(num_node=111_059_956, mean_deg=14.55, alpha=0.7, dmax=300_000, feat_dim=128, num_class=172,
                       num_train=1_207_179),
    # com-friendster as the reference generates it (datagen/friendster.py:68-69: 65.6 M nodes, 1.8 G directed edges;
    # synthetic 256-dim f32 features per BASELINE configs[4], 1 % of the nodes as the train set)
    "friendster": dict(num_node=65_608_366, mean_deg=27.53, alpha=0.7, dmax=6_000, feat_dim=256, num_class=100,
                       num_train=656_083),
    "tiny": dict(num_node=20_000, mean_deg=30.0, alpha=0.75, dmax=2_000, feat_dim=100, num_class=47,
                 num_train=4_000),
}


def powerlaw_degrees(num_node, mean_deg, alpha, dmax, rng):
    """d_v = min(dmax, floor(c * u^-alpha)), c tuned so that the mean is mean_deg."""
    u = rng.random_sample(num_node)
    base = u ** (-alpha)
    calib = base if num_node <= (1 << 22) else base[:: num_node // (1 << 22)]  # calibrate c on a subsample
    lo, hi = 1e-3, 1e6
    for _ in range(60):  # bisection on c
        c = 0.5 * (lo + hi)
        m = np.minimum(dmax, np.floor(c * calib)).mean()
        if m < mean_deg:
            lo = c
        else:
            hi = c
    return np.minimum(dmax, np.floor(hi * base)).astype(np.int64)


def _host_threads():
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(16, n))


def make_graph(preset="products", seed=42, chunk=1 << 24, neighbour_skew=0.0, threads=None):
    """Returns dict(indptr uint32[N+1], indices uint32[E], train_set uint32[T], meta).

    neighbour_skew = 0 (the default everywhere): neighbour ids uniform over the nodes.
    neighbour_skew = p in (0, 1]: each neighbour is, with probability p, the OWNER OF A UNIFORMLY RANDOM EDGE SLOT
    (probability proportional to the node's degree -- the hubs of the power law turn up in many lists, as in a real
    symmetrised graph) and uniform otherwise.  Chunks are seeded one by one, so the result does not depend on the
    number of host threads that fill them."""
    p = dict(PRESETS[preset]) if isinstance(preset, str) else dict(preset)
    rng = np.random.RandomState(seed)
    n = p["num_node"]
    deg = powerlaw_degrees(n, p["mean_deg"], p["alpha"], p["dmax"], rng)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    num_edge = int(indptr[-1])
    assert num_edge < 2 ** 32, "IdType is uint32 (constant.h:28): num_edge must stay below 2^32"
    indices = np.empty(num_edge, dtype=np.uint32)
    owner = None
    if neighbour_skew > 0:  # owner[e] = the node whose list holds edge slot e
        owner = np.repeat(np.arange(n, dtype=np.uint32), deg)

    def fill(s):  # chunked: bounded temporaries; numpy releases the GIL in randint / take
        e = min(num_edge, s + chunk)
        ids = np.random.RandomState(1234 + s // chunk).randint(0, n, size=e - s, dtype=np.int64)
        if owner is not None:
            r2 = np.random.RandomState(991234 + s // chunk)
            slot = r2.randint(0, num_edge, size=e - s, dtype=np.int64)
            hub = r2.random_sample(e - s) < neighbour_skew
            ids = np.where(hub, owner.take(slot), ids)
        indices[s:e] = ids

    starts = range(0, num_edge, chunk)
    nt = threads or _host_threads()
    if nt > 1 and len(starts) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(nt) as ex:
            list(ex.map(fill, starts))
    else:
        for s in starts:
            fill(s)
    del owner
    train = np.random.RandomState(seed + 1).permutation(n)[: p["num_train"]].astype(np.uint32)
    p["num_edge"] = num_edge
    p["neighbour_skew"] = float(neighbour_skew)
    return dict(indptr=indptr.astype(np.uint32), indices=indices, train_set=train, meta=p)


def degree_rank(indptr):
    """cache_by_degree.bin equivalent: node ids by descending in-degree (stable)."""
    deg = (indptr[1:].astype(np.int64) - indptr[:-1].astype(np.int64))
    return np.argsort(-deg, kind="stable").astype(np.uint32)


def edge_weights(graph, policy="default", seed=0):
    """Per-edge weights in the spirit of the reference's weight tool (create_alias_table.cc:36-60,75-92), seeded:
    'default' = integers 1..10, 'inverse_src_degree' = 1 / out-degree of the neighbour, 'src_suffix' = 100 for
    neighbours with out-degree < 10 else 1."""
    ip, ix = graph["indptr"], graph["indices"]
    if policy == "default":
        return np.random.RandomState(seed).randint(1, 11, size=ix.size).astype(np.float32)
    out_deg = np.bincount(ix, minlength=ip.size - 1).astype(np.int64)
    if policy == "inverse_src_degree":
        return (1.0 / out_deg[ix]).astype(np.float32)
    if policy == "src_suffix":
        return np.where(out_deg[ix] < 10, 100.0, 1.0).astype(np.float32)
    raise ValueError(policy)


def build_alias_tables(indptr, indices, weights, num_threads=8):
    """prob_table / alias_table of the alias-method samplers from per-edge weights (create_alias_table.cc:105-170);
    the alias slot holds the GLOBAL node id of the donor neighbour.  Host-only entry point of the library."""
    import ctypes as C
    from ._lib import check, lib
    ip = np.ascontiguousarray(indptr, np.uint32)
    ix = np.ascontiguousarray(indices, np.uint32)
    w = np.ascontiguousarray(weights, np.float32)
    assert w.size == ix.size
    prob, alias = np.empty(ix.size, np.float32), np.empty(ix.size, np.uint32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    check(lib().ggms_build_alias_table_host(vp(ip), vp(ix), ip.size - 1, vp(w), vp(prob), vp(alias), num_threads),
          "ggms_build_alias_table_host")
    return prob, alias


def build_prob_prefix_table(indptr, weights, num_threads=8):
    """prob_prefix_table of the inverse-CDF sampler: running float sum per neighbour list
    (create_prob_prefix_table.cc:94-123)."""
    import ctypes as C
    from ._lib import check, lib
    ip = np.ascontiguousarray(indptr, np.uint32)
    w = np.ascontiguousarray(weights, np.float32)
    out = np.empty(w.size, np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    check(lib().ggms_build_prob_prefix_table_host(vp(ip), ip.size - 1, vp(w), vp(out), num_threads),
          "ggms_build_prob_prefix_table_host")
    return out


def write_dataset(path, graph, feat=None, label=None, valid_frac=0.02, test_frac=0.05, feat_dtype="F32", weights=None,
                  minimal=False):
    """Write a dataset directory in the reference's on-disk format (datagen/README.md:37-51,
    samgraph/common/constant.cc:23-51, engine.cc:109-443): meta.txt (tab separated) + raw little-endian
    arrays: indptr/indices/train_set/test_set/valid_set/cache_by_* uint32, feat row-major, label int64;
    weights (one float per edge) adds prob_table.bin / alias_table.bin / prob_prefix_table.bin.
    minimal: what a sampling + extract run needs and nothing that costs minutes at papers100M size -- one-node
    valid / test sets instead of random slices of the complement, no cache_by_random.bin."""
    import os
    os.makedirs(path, exist_ok=True)
    ip, ix, train, meta = graph["indptr"], graph["indices"], graph["train_set"], graph["meta"]
    n = ip.size - 1
    if minimal:
        valid = test = np.zeros(1, np.uint32)
    else:
        rng = np.random.RandomState(7)
        rest = np.setdiff1d(np.arange(n, dtype=np.uint32), train, assume_unique=False)
        rng.shuffle(rest)
        nv, nt = int(n * valid_frac), int(n * test_frac)
        valid, test = rest[:nv].astype(np.uint32), rest[nv:nv + nt].astype(np.uint32)
    ip.astype(np.uint32).tofile(os.path.join(path, "indptr.bin"))
    ix.astype(np.uint32).tofile(os.path.join(path, "indices.bin"))
    train.astype(np.uint32).tofile(os.path.join(path, "train_set.bin"))
    valid.tofile(os.path.join(path, "valid_set.bin"))
    test.tofile(os.path.join(path, "test_set.bin"))
    if feat is not None:
        np.ascontiguousarray(feat).tofile(os.path.join(path, "feat.bin"))
    if label is not None:
        np.ascontiguousarray(label, dtype=np.int64).tofile(os.path.join(path, "label.bin"))
    if weights is not None:  # tables of the weighted samplers (engine.cc:372-384 loads them by these names)
        prob, alias = build_alias_tables(ip, ix, weights)
        prob.tofile(os.path.join(path, "prob_table.bin"))
        alias.tofile(os.path.join(path, "alias_table.bin"))
        build_prob_prefix_table(ip, weights).tofile(os.path.join(path, "prob_prefix_table.bin"))
    degree_rank(ip).tofile(os.path.join(path, "cache_by_degree.bin"))
    if not minimal:
        np.random.RandomState(11).permutation(n).astype(np.uint32).tofile(os.path.join(path, "cache_by_random.bin"))
    with open(os.path.join(path, "meta.txt"), "w") as f:
        for k, v in [("NUM_NODE", n), ("NUM_EDGE", ix.size), ("FEAT_DIM", meta["feat_dim"]),
                     ("NUM_CLASS", meta["num_class"]), ("NUM_TRAIN_SET", train.size), ("NUM_TEST_SET", test.size),
                     ("NUM_VALID_SET", valid.size)]:
            f.write(f"{k}\t{v}\n")
        if feat_dtype != "F32":
            f.write(f"FEAT_DATA_TYPE\t{feat_dtype}\n")
    return path
