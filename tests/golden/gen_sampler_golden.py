"""Writes tests/golden/sampler_golden.npz: small known-answer vectors of the GPU-side sampler semantics as THIS build
defines them (lock-step khop3, highest-j-wins khop0, first-occurrence dedup, cuRAND XORWOW
constants of tests/golden/xorwow_constants.json) -- inputs, RNG seed and every output of DoGPUSample for each sample
type, produced by the oracle.

    python tests/golden/gen_sampler_golden.py

The reference holds no fixtures for these paths and its CUDA engine cannot be built here, so these vectors are a
REGRESSION pin (oracle and HIP kernels must keep producing them), not a pin against the CUDA engine.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

import oracle  # noqa: E402
from graphgen import powerlaw_csr  # noqa: E402

CASES = {  # name -> (oracle code, fanouts, extra)
    "khop3": (oracle.KHOP3, [6, 4], {}),
    "khop0": (oracle.KHOP0, [6, 4], {}),
    "khop2": (oracle.KHOP2, [6, 4], {}),
    "khop1": (oracle.KHOP1, [6, 4], {}),
    "weighted_khop": (oracle.WEIGHTED_KHOP, [6, 4], {"weighted": True}),
    "random_walk": (oracle.RANDOM_WALK, [5, 5], {"walk_length": 3, "restart_prob": 0.5, "num_walk": 4}),
}
SEED, NUM_STATES, NUM_SEEDS = 0x5EED, 16384, 300


def main():
    ip, ix = powerlaw_csr(2500, mean_deg=12, seed=11)
    rng = np.random.RandomState(5)
    seeds = rng.permutation(2500)[:NUM_SEEDS].astype(np.uint32)
    seeds[7] = seeds[3]  # a repeated seed: local ids of raw seeds go through the table
    weights = rng.randint(1, 11, ix.size).astype(np.float32)
    prob, alias = oracle.create_alias_table(ip, ix, weights)
    out = dict(indptr=ip, indices=ix, seeds=seeds, prob=prob, alias=alias, rng_seed=np.uint64(SEED),
               num_states=np.uint64(NUM_STATES))
    for name, (code, fanouts, extra) in CASES.items():
        states = oracle.random_states(NUM_STATES, SEED)
        kw = dict(extra)
        if kw.pop("weighted", False):
            kw.update(prob=prob, alias=alias)
        s = seeds if name != "khop2" else np.unique(seeds)  # khop2 needs distinct seeds
        res = oracle.do_sample(code, ip, ix.copy(), s, fanouts, states, **kw)
        out[f"{name}:fanouts"] = np.array(fanouts, np.uint32)
        out[f"{name}:input_nodes"] = res["input_nodes"]
        for i, l in enumerate(res["layers"]):
            out[f"{name}:row{i}"], out[f"{name}:col{i}"] = l["row"], l["col"]
            out[f"{name}:num{i}"] = np.array([l["num_src"], l["num_dst"]], np.uint64)
            if l["data"] is not None:
                out[f"{name}:data{i}"] = l["data"]
        # the RNG pool after the batch: the first 2048 streams (the live parity tests compare the whole pool)
        out[f"{name}:states_d"], out[f"{name}:states_v"] = states["d"][:2048].copy(), states["v"][:2048].copy()
    np.savez_compressed(os.path.join(HERE, "sampler_golden.npz"), **out)
    print("wrote sampler_golden.npz", os.path.getsize(os.path.join(HERE, "sampler_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
