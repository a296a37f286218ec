"""Seeded synthetic CSR graphs / features for the tests (numpy only)."""
import numpy as np


def powerlaw_csr(num_node, mean_deg=8.0, alpha=0.8, dmax=None, seed=0, zero_frac=0.05):
    """In-neighbour CSR with power-law degrees, uniform random neighbours.

    Some nodes get degree 0 (zero_frac) so the len==0 / len<=fanout branches of
    every sampler are exercised.
    """
    rng = np.random.RandomState(seed)
    u = rng.random_sample(num_node)
    deg = np.floor(u ** (-alpha)).astype(np.int64)
    deg = np.maximum(1, (deg * (mean_deg / max(deg.mean(), 1e-9))).astype(np.int64))
    if dmax is None:
        dmax = max(4, num_node // 2)
    deg = np.minimum(deg, dmax)
    deg[rng.random_sample(num_node) < zero_frac] = 0
    indptr = np.zeros(num_node + 1, dtype=np.uint32)
    indptr[1:] = np.cumsum(deg)
    indices = rng.randint(0, num_node, size=int(indptr[-1])).astype(np.uint32)
    return indptr, indices


def hub_csr(num_node=3000, num_hub=6, hub_deg=6000, base_deg=9, seed=0):
    """A few very long neighbour lists among short ones: the samplers' per-list limits (khop3 fanout 127, khop0's
    heavy-list path and its LDS slot budget) and tail behaviour."""
    rng = np.random.RandomState(seed)
    deg = np.full(num_node, base_deg, np.int64)
    deg[rng.permutation(num_node)[:num_hub]] = hub_deg
    deg[rng.random_sample(num_node) < 0.03] = 0
    indptr = np.zeros(num_node + 1, dtype=np.uint32)
    indptr[1:] = np.cumsum(deg)
    indices = rng.randint(0, num_node, size=int(indptr[-1])).astype(np.uint32)
    return indptr, indices


def exact_features(num_node, dim, dtype=np.float32):
    """feat[i, j] = (i * dim + j) & 0xFFFF -- exactly representable everywhere."""
    v = (np.arange(num_node * dim, dtype=np.int64) & 0xFFFF).reshape(num_node, dim)
    if np.dtype(dtype) == np.uint8:
        v = v & 0xFF
    return v.astype(dtype)


def prefix_sums(indptr, weights):
    """Per-neighbour-list inclusive prefix sums in float32 (the prob_prefix_table.bin layout)."""
    import numpy as np
    out = np.zeros(weights.size, np.float32)
    for v in range(indptr.size - 1):
        a, b = int(indptr[v]), int(indptr[v + 1])
        if b > a:
            out[a:b] = np.cumsum(weights[a:b], dtype=np.float32)
    return out


# The literal inputs of the reference's own hash-table unit test (samgraph/unittest/test_hashmap.cc:98-126,166-198):
# successive FillWithDuplicates calls on one table.  What that test asserts -- and ours with it -- is set equality
# with the inputs so far, no duplicates in the unique list, and that each fill keeps the previous list as a prefix.
REF_HASHMAP_VECTORS = {
    "DupRevised_Ref": [[1, 2, 3, 4, 5, 2, 5, 1, 10, 233],
                       [1, 2, 3, 6, 9, 2, 8, 1, 3, 7, 1023],
                       [1, 2, 3, 4, 5, 2, 5, 1, 10, 233, 1, 2, 3, 6, 9, 2, 8, 1, 3, 7, 1023]],
    "MixedFillMethod": [[1, 2, 3, 4, 5, 2, 5, 1, 10, 256, 6, 9, 2, 13, 5, 64, 512, 1021],
                        list(range(1, 513))],
}
REF_HASHMAP_EXPECT = {"MixedFillMethod": [13, 513]}  # EXPECT_EQ(h_set.size(), 13) / EXPECT_EQ(output_uniq.size(), 513)
