"""Experiment (diagnostic, not the bench): how much sampling throughput is there to gain from keeping K batches
in flight on K streams?  The sampling kernels are latency-bound (few long serial waves), so independent batches
should overlap.  Here the K samplers are fully independent (own table, own RNG pool); the real engine has to
order the RNG pool across batches, which this upper bound ignores.

    python tools/exp_pipelined_samplers.py [--preset products] [--fanout 25,10] [--batch 8000] [--type khop3]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xgnn_amd import datagen, ops, parallel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="products")
    ap.add_argument("--fanout", default="25,10")
    ap.add_argument("--batch", type=int, default=8000)
    ap.add_argument("--type", default="khop3")
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    fanouts = [int(x) for x in args.fanout.split(",")]
    graph = datagen.make_graph(args.preset, seed=42)

    def to_dev(a):
        return torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(dev)

    g = ops.DeviceGraph(to_dev(graph["indptr"]), to_dev(graph["indices"]))
    train = graph["train_set"]
    seeds_all = to_dev(parallel.rank_slice(train, 1, 0, 0))
    nb = len(train) // args.batch
    code = {"khop3": ops.KHOP3, "khop0": ops.KHOP0, "khop2": ops.KHOP2, "khop1": ops.KHOP1}[args.type]
    for K in (1, 2, 3, 4):
        samplers = [ops.BatchSampler(g, fanouts, args.batch, sample_type=code, seed=1 + i, device=dev) for i in range(K)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(K)]

        def run(n):
            for s in range(n):
                k = s % K
                seeds = seeds_all[(s % nb) * args.batch:(s % nb + 1) * args.batch]
                with torch.cuda.stream(streams[k]):
                    samplers[k].sample(seeds)

        run(2 * K)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print(f"K={K}: {dt * 1e3:.3f} ms per batch", flush=True)
        del samplers


if __name__ == "__main__":
    main()
