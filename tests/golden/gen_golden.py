"""Generate golden vectors from the REFERENCE's own CPU leaf objects.

Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/gen_golden.py

It drives oracle/_ref/libref_cpu.so -- cpu/cpu_sampling_khop0.cc, cpu/cpu_sampling_khop2.cc,
cpu/cpu_random.cc, cpu/cpu_extraction.cc compiled in place from
/root/reference/samgraph/common -- on seeded inputs and stores inputs +
outputs under tests/golden/.  Only data is stored, no reference text.

RandomID's generator is a process-wide default-seeded thread_local mt19937
(cpu_random.cc:27), so the sampling vectors are produced by ONE fresh process
at omp_thread_num = 1 in the call order recorded in `order`; a checker replays
the same order from a freshly reset generator.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

import oracle  # noqa: E402
from graphgen import powerlaw_csr  # noqa: E402


def main():
    assert oracle.ref_lib() is not None, "build oracle/_ref first (make -C oracle ref)"
    out = {}

    # ---- RandomID stream (std::mt19937 default seed + uniform_int_distribution)
    r = oracle.ref_lib()
    bounds = [(0, 1), (0, 2), (0, 9), (0, 25), (0, 26), (3, 3), (5, 100000), (0, 0xFFFFFFFE),
              (0, 0xFFFFFFFF), (7, 0x80000000)]
    seq = []
    for rep in range(40):
        for lo, hi in bounds:
            seq.append(r.ref_random_id(np.uint32(lo).item(), np.uint32(hi).item()))
    out["randid_bounds"] = np.array(bounds, dtype=np.uint64)
    out["randid_values"] = np.array(seq, dtype=np.uint32)

    # ---- CPUSampleKHop0, sequential calls in one process
    order = []
    # toy graph of SURVEY 8c plus power-law graphs
    toy_indptr = np.array([0, 2, 5, 5, 9], dtype=np.uint32)
    toy_indices = np.array([1, 2, 0, 2, 3, 0, 1, 2, 3], dtype=np.uint32)
    graphs = {
        "toy": (toy_indptr, toy_indices),
        "pl200": powerlaw_csr(200, mean_deg=12, seed=1),
        "pl2k": powerlaw_csr(2000, mean_deg=14, seed=2),
    }
    for gname, (ip, ix) in graphs.items():
        out[f"g_{gname}_indptr"] = ip
        out[f"g_{gname}_indices"] = ix
    rng = np.random.RandomState(7)
    calls = [
        ("toy", np.array([0, 3, 1, 2], np.uint32), 2),
        ("toy", np.array([3, 3, 1], np.uint32), 3),
        ("pl200", rng.randint(0, 200, 64).astype(np.uint32), 5),
        ("pl200", np.arange(200, dtype=np.uint32), 25),
        ("pl200", np.zeros(0, np.uint32), 4),
        ("pl2k", rng.permutation(2000)[:800].astype(np.uint32), 10),
        ("pl2k", rng.randint(0, 2000, 1500).astype(np.uint32), 25),
        ("pl2k", rng.randint(0, 2000, 100).astype(np.uint32), 1),
    ]
    for ci, (gname, inp, fanout) in enumerate(calls):
        ip, ix = graphs[gname]
        src, dst = oracle.ref_cpu_sample_khop0(ip, ix, inp, fanout, num_threads=1)
        out[f"khop0_{ci}_input"] = inp
        out[f"khop0_{ci}_src"] = src
        out[f"khop0_{ci}_dst"] = dst
        order.append(f"{gname}:{fanout}")
    out["khop0_order"] = np.array(order)

    # ---- CPUSampleKHop2 (in-place Fisher-Yates), continuing on the SAME RandomID stream; each graph's
    # `indices` is permuted by the calls, so the calls on one graph chain and the final lists are stored
    order2 = []
    mutable = {k: (ip, ix.copy()) for k, (ip, ix) in graphs.items()}
    calls2 = [
        ("toy", np.array([0, 3, 1, 2], np.uint32), 2),
        ("toy", np.array([3, 1], np.uint32), 3),
        ("pl200", rng.permutation(200)[:64].astype(np.uint32), 5),
        ("pl200", np.arange(200, dtype=np.uint32), 25),
        ("pl200", np.zeros(0, np.uint32), 4),
        ("pl2k", rng.permutation(2000)[:800].astype(np.uint32), 10),
        ("pl2k", rng.permutation(2000)[:1500].astype(np.uint32), 25),
        ("pl2k", rng.randint(0, 2000, 100).astype(np.uint32), 1),  # repeated seeds: sequential at 1 thread
    ]
    for ci, (gname, inp, fanout) in enumerate(calls2):
        ip, ix = mutable[gname]
        src, dst = oracle.ref_cpu_sample_khop2(ip, ix, inp, fanout, num_threads=1)
        out[f"khop2_{ci}_input"] = inp
        out[f"khop2_{ci}_src"] = src
        out[f"khop2_{ci}_dst"] = dst
        out[f"khop2_{ci}_indices_after"] = ix.copy()
        order2.append(f"{gname}:{fanout}")
    out["khop2_order"] = np.array(order2)

    # ---- CPUExtract over the dtypes the reference dispatches (cpu_extraction.cc:66-90)
    rng = np.random.RandomState(11)
    cases = [("f32", np.float32, 100), ("f64", np.float64, 5), ("f16", np.int16, 7),
             ("u8", np.uint8, 3), ("i32", np.int32, 32), ("i64", np.int64, 1)]
    for name, dt, dim in cases:
        n = 300
        if np.issubdtype(dt, np.floating):
            table = rng.standard_normal((n, dim)).astype(dt)
        else:
            info = np.iinfo(dt)
            table = rng.randint(info.min, info.max, size=(n, dim), dtype=dt)
        idx = rng.randint(0, n, 157).astype(np.uint32)
        got = oracle.ref_cpu_extract(table, idx, num_threads=1)
        out[f"extract_{name}_table"] = table
        out[f"extract_{name}_index"] = idx
        out[f"extract_{name}_out"] = got
    np.savez_compressed(os.path.join(HERE, "ref_cpu_leaves.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_cpu_leaves.npz"),
          os.path.getsize(os.path.join(HERE, "ref_cpu_leaves.npz")), "bytes")


if __name__ == "__main__":
    main()
