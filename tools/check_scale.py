"""Size-independent properties of one sampled batch at a large preset (run on the GPU box; the products-size twin is
tests/test_gpu_parity.py::test_full_size_batch_properties):  determinism, hashed == direct table, COO validity,
min(deg, fanout) edges per seed, sampled edges exist in the CSR, first-occurrence numbering, gathered rows.

    python tools/check_scale.py papers100M 5,10,15
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xgnn_amd import datagen, ops  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "papers100M"
fanouts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "5,10,15").split(",")]
batch = 8000
dev = torch.device("cuda", 0)
g = datagen.make_graph(preset, seed=42)
ip, ix = g["indptr"], g["indices"]
to_dev = lambda a: torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(dev)  # noqa: E731
u32 = lambda t: t.cpu().numpy().view(np.uint32)  # noqa: E731
graph = ops.DeviceGraph(to_dev(ip), to_dev(ix))
seeds = g["train_set"][:batch]
L = len(fanouts)
results = []
for direct in (True, False, True):
    bs = ops.BatchSampler(graph, fanouts, batch, sample_type=ops.KHOP3, seed=0x5EED, direct_table=direct)
    bs.sample(to_dev(seeds))
    r = bs.result()
    results.append(dict(inp=u32(r["input_nodes"]).copy(),
                        layers=[(u32(l["row"]).copy(), u32(l["col"]).copy(), l["num_src"], l["num_dst"]) for l in r["layers"]]))
    del bs
    torch.cuda.empty_cache()
a = results[0]
for b in results[1:]:
    assert np.array_equal(a["inp"], b["inp"])
    for la, lb in zip(a["layers"], b["layers"]):
        assert np.array_equal(la[0], lb[0]) and np.array_equal(la[1], lb[1]) and la[2:] == lb[2:]
print("deterministic, hashed == direct")
inp = a["inp"]
assert np.unique(inp).size == inp.size and np.array_equal(inp[:batch], seeds)
deg = ip[1:].astype(np.int64) - ip[:-1].astype(np.int64)
assert a["layers"][L - 1][3] == batch and a["layers"][0][2] == inp.size
for i in range(L):
    row, col, nsrc, ndst = a["layers"][i]
    assert row.max() < nsrc and col.max() < ndst and (np.diff(col.astype(np.int64)) >= 0).all()
    if i + 1 < L:
        assert a["layers"][i + 1][2] == ndst
    assert np.array_equal(np.bincount(col, minlength=ndst), np.minimum(deg[inp[:ndst]], fanouts[i]))
    rng = np.random.RandomState(i)
    for e in rng.randint(0, row.size, 1000):
        s, d = inp[col[e]], inp[row[e]]
        assert d in ix[ip[s]:ip[s + 1]]
    first_pos = np.full(nsrc, row.size, np.int64)
    np.minimum.at(first_pos, row, np.arange(row.size))
    assert (np.diff(first_pos[np.arange(ndst, nsrc)]) > 0).all()
print(f"{preset} {fanouts}: {sum(l[0].size for l in a['layers'])} edges, {inp.size} input nodes -- all properties hold")
