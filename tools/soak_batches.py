"""Soak: full-size batches (products-shaped graph, batch 8000) sampled back to back on two pipelines with a gather
running beside them, every batch compared with the oracle -- hunts for rare inter-workgroup races in the ticket /
look-back / dedup protocols that the small parity cases cannot provoke.

    python tools/soak_batches.py [num_batches] [sample_type]
    GGMS_TEST_SCAN_PATIENCE=0 python tools/soak_batches.py ...      the scans' look-backs never wait (self-serve path)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from xgnn_amd import datagen, ops  # noqa: E402

if os.environ.get("GGMS_TEST_SCAN_PATIENCE") is not None:  # 0: every look-back that finds a word missing serves itself
    from xgnn_amd import lib  # noqa: E402
    lib().ggms_debug_set_scan_patience(int(os.environ["GGMS_TEST_SCAN_PATIENCE"]))
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 40
stype = sys.argv[2] if len(sys.argv) > 2 else "khop3"
code = {"khop3": ops.KHOP3, "khop0": ops.KHOP0, "khop1": ops.KHOP1, "khop2": ops.KHOP2,
        "random_walk": ops.RANDOM_WALK}[stype]
ocode = {"khop3": oracle.KHOP3, "khop0": oracle.KHOP0, "khop1": oracle.KHOP1, "khop2": oracle.KHOP2,
         "random_walk": oracle.RANDOM_WALK}[stype]
walk = stype == "random_walk"  # PinSAGE defaults (sgnn/train_pinsage.py:138-142)
kw = dict(random_walk_length=3, random_walk_restart_prob=0.5, num_random_walk=4) if walk else {}
okw = dict(walk_length=3, restart_prob=0.5, num_walk=4) if walk else {}
dev = torch.device("cuda", 0)
g = datagen.make_graph("products", seed=42)
ip, ix = g["indptr"], g["indices"]
to_dev = lambda a: torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(dev)  # noqa: E731
graph = ops.DeviceGraph(to_dev(ip), to_dev(ix))  # khop2 permutes this copy; the oracle permutes `ix` the same way
if stype == "khop2":
    ix = ix.copy()
u32 = lambda t, n: t[:n].cpu().numpy().view(np.uint32)  # noqa: E731
noise_src = torch.empty((1 << 28,), dtype=torch.float32, device=dev)
noise_dst = torch.empty_like(noise_src)
for fanouts in ([5, 5, 5], [10, 5]) if walk else ([25, 10], [5, 10, 15]):
    K, L = 2, len(fanouts)
    bs = ops.BatchSampler(graph, fanouts, 8000, sample_type=code, seed=123, num_slots=nb, num_pipelines=K, **kw)
    states = oracle.random_states(bs.states.shape[0], 123) if stype != "khop0" else None
    rng = np.random.RandomState(1)
    seeds = [g["train_set"][rng.permutation(g["train_set"].size)[:8000]] for _ in range(nb)]
    t_seeds = [to_dev(s) for s in seeds]
    streams = [torch.cuda.Stream() for _ in range(K)]
    noise = torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.time()
    for b in range(nb):
        with torch.cuda.stream(streams[b % K]):
            # every other batch with the distinct-seed promise (its seeds are a slice of a permutation): the fused /
            # one-launch seed paths and the general insert + scan + look-up path interleave on the same tables
            bs.sample(t_seeds[b], slot=b, copy_input_nodes=True, distinct=(b % 2 == 0))
        with torch.cuda.stream(noise):  # an HBM stream beside the samplers, like the feature gather
            noise_dst.copy_(noise_src)
    torch.cuda.synchronize()
    assert ops.device_status() == 0
    for b in range(nb):
        want = oracle.do_sample(ocode, ip, ix, seeds[b], fanouts, states, **okw)
        c = bs.counts_slots[b].cpu().tolist()
        assert c[3 * L + 1] == 0, (fanouts, b, "batch status word", c[3 * L + 1])
        assert np.array_equal(u32(bs.input_nodes[b], c[3 * L]), want["input_nodes"]), (fanouts, b, "input_nodes")
        for i in range(L):
            wl = want["layers"][i]
            assert (c[3 * i], c[3 * i + 1], c[3 * i + 2]) == (wl["row"].size, wl["num_src"], wl["num_dst"]), (fanouts, b, i)
            assert np.array_equal(u32(bs.rows[b][i], c[3 * i]), wl["row"]), (fanouts, b, i, "row")
            assert np.array_equal(u32(bs.cols[b][i], c[3 * i]), wl["col"]), (fanouts, b, i, "col")
            if walk:
                assert np.array_equal(u32(bs.datas[b][i], c[3 * i]), wl["data"]), (fanouts, b, i, "data")
    print(f"soak ok: {stype} {fanouts}: {nb} batches of 8000 seeds on {K} pipelines, all equal to the oracle "
          f"({time.time() - t0:.1f} s)", flush=True)
    del bs
    torch.cuda.empty_cache()
