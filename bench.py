#!/usr/bin/env python3
"""bench.py -- GGMS hot path on MI355X: sampled edges/s + feature-extract GB/s.

One "step" = one mini-batch of the hot path on ONE GPU, inputs resident in HBM:
    seeds (8000 train nodes) -> 2-layer khop3 neighbour sampling with ordered
    dedup/remap (DoGPUSample) -> feature rows of the batch's input nodes gathered
    through the cache table (cache_ratio = 1.0) + label gather.
Workload (BASELINE.json configs[1]): products-shaped synthetic power-law CSR
(N 2,449,029, E ~1.24e8, f32 dim 100), GraphSAGE fanout [25,10], batch 8000.

N > 1 (driver: torch.distributed.run, one rank per GPU): data parallel over seed
mini-batches, every rank samples its own slice of the shuffled train set from
its own replica of graph + features -- no data-path collective (weak scaling).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preset", default="products")
    ap.add_argument("--batch", type=int, default=8000)
    ap.add_argument("--fanout", default="25,10")
    ap.add_argument("--sample-type", default="khop3",
                    choices=["khop3", "khop0", "khop2", "khop1", "weighted_khop", "weighted_khop_hash_dedup",
                             "random_walk"],
                    help="random_walk: PinSAGE defaults (walk length 3, restart 0.5, 4 walks); --fanout gives the "
                         "top-K per layer, e.g. 5,5,5.  weighted_khop: synthetic alias tables")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline: keep sampling mini-batches this long")
    ap.add_argument("--host-profile", action="store_true", help="print host enqueue time per section to stderr")
    ap.add_argument("--pipelines", type=int, default=1,
                    help="sampling batches in flight (each on its own stream with its own dedup table).  1: the gather "
                         "keeps ~93 %% of its standalone rate beside the sampler; 2: +18 %% edges/s (the engine's "
                         "default) while the gather, sharing HBM with two sampled batches, drops to ~0.40 of peak")
    ap.add_argument("--no-overlap", action="store_true",
                    help="one stream: extract of batch k and sampling of batch k+1 run back to back "
                         "(default: two streams, the HBM-bound gather overlaps the latency-bound sampler)")
    ap.add_argument("--store", default="replica", choices=["replica", "peer", "a2a"],
                    help="N > 1: 'replica' = every GPU holds the cached features (DP over seeds only); 'peer' = GGMS "
                         "feature shards (slot %% N), rows read from the owner's HBM inside the gather kernel over xGMI "
                         "(hipIpc); 'a2a' = same shards, rows exchanged with RCCL all-to-all")
    ap.add_argument("--cache-ratio", type=float, default=1.0,
                    help="fraction of feature rows (by degree rank) resident in HBM; the rest is gathered from "
                         "pinned host memory by the same kernel (GGMS host tier). 1.0 = BASELINE configs[1]")
    return ap.parse_args()


def usable_cores():
    """Threads this process may really use: cgroup CPU quota if set, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def measured_traffic():
    """HBM traffic of the extract kernel from the committed PMC profile (profiles/*_extract_traffic.json):
    collected with separate `rocprofv3 --pmc` passes of this same command, FETCH_SIZE corrected x2."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_extract_traffic.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    d["source"] = os.path.relpath(files[-1], ROOT)
    return d


def cpu_baseline(graph, fanouts, batch, feat_dim, seconds):
    """CPU leg: the oracle (port) and, when shipped, the reference's own CPU leaves (oracle/_ref).
    Bounded sample: mini-batches of the same workload on the usable host cores for about `seconds` of CPU work."""
    import oracle
    cores = usable_cores()
    ip, ix, train = graph["indptr"], graph["indices"], graph["train_set"]
    n_node = ip.size - 1
    feat = (np.arange(n_node * feat_dim, dtype=np.int64) & 0xFFFF).astype(np.float32).reshape(n_node, feat_dim)
    have_ref = oracle.ref_lib() is not None
    sample = oracle.ref_cpu_sample_khop0 if have_ref else oracle.cpu_sample_khop0
    extract = oracle.ref_cpu_extract if have_ref else oracle.extract
    t_sample = t_remap = t_extract = 0.0
    edges = rows = 0
    n_batches = 0
    per_epoch = max(1, len(train) // batch)
    t_begin = time.perf_counter()
    while time.perf_counter() - t_begin < seconds or n_batches < 3:
        b = n_batches % per_epoch
        if b == 0 and n_batches:
            train = train[np.random.RandomState(n_batches).permutation(len(train))]
        n_batches += 1
        seeds = train[b * batch:(b + 1) * batch]
        ht = oracle.HashTable(n_node, oracle.predict_num_nodes(len(seeds), fanouts, len(fanouts)) + 1)
        t0 = time.perf_counter()
        ht.fill_with_duplicates(seeds)
        t_remap += time.perf_counter() - t0
        cur = seeds
        for i in range(len(fanouts) - 1, -1, -1):
            t0 = time.perf_counter()
            src, dst = sample(ip, ix, cur, fanouts[i], cores)
            t1 = time.perf_counter()
            ht.fill_with_duplicates(dst)
            ht.map_edges(src, dst)
            cur = ht.unique()
            t2 = time.perf_counter()
            t_sample += t1 - t0
            t_remap += t2 - t1
            edges += src.size
        t0 = time.perf_counter()
        out = extract(feat, cur, cores)
        t_extract += time.perf_counter() - t0
        rows += cur.size
        del out
    total = t_sample + t_remap + t_extract
    return {
        "value": edges / total, "unit": "edges/s", "cores": cores,
        "kind": "reference" if have_ref else "port",
        "sample": f"{n_batches} mini-batches of {batch} seeds, fanout {fanouts}, same synthetic graph; "
                  f"sampler+extract = {'reference CPUSampleKHop0/CPUExtract objects (oracle/_ref)' if have_ref else 'oracle port'}"
                  f" on {cores} threads, dedup/remap = oracle port of CPUHashTable2 (1 thread)",
        "sample_only_edges_per_s": edges / t_sample,
        "feature_GBps": rows * feat_dim * 4 / t_extract / 1e9,
        "seconds": total,
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (MI355X); there is no CPU fallback for the product path")
    # test hooks (one-GPU box): GGMS_BENCH_DEVICE pins every rank to one device, GGMS_BENCH_BACKEND=gloo avoids
    # RCCL's one-rank-per-GPU rule.  The driver's multi-GPU runs use neither: one rank per GPU over RCCL.
    dev_index = int(os.environ.get("GGMS_BENCH_DEVICE", local_rank))
    backend = os.environ.get("GGMS_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", dev_index)

    from xgnn_amd import datagen, ops, parallel

    fanouts = [int(x) for x in args.fanout.split(",")]
    graph = datagen.make_graph(args.preset, seed=42)
    meta = graph["meta"]
    N, dim = meta["num_node"], meta["feat_dim"]

    def to_dev(a):
        return torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(dev)

    indptr, indices = to_dev(graph["indptr"]), to_dev(graph["indices"])
    g = ops.DeviceGraph(indptr, indices)
    labels = (torch.arange(N, dtype=torch.int64, device=dev) % meta["num_class"]).contiguous()

    def feat_rows(node_ids, out):
        """feat[i, j] = float((i*dim + j) & 0xFFFF), generated in place for the given node ids."""
        cols = torch.arange(dim, dtype=torch.int64, device=out.device)
        step = 1 << 22
        for lo in range(0, node_ids.numel(), step):
            ids = node_ids[lo:lo + step].to(out.device, torch.int64)
            out[lo:lo + step] = ((ids[:, None] * dim + cols[None, :]) & 0xFFFF).to(torch.float32)

    # degree policy: cache slot r holds node rank[r] (r < num_cached); table[node] = slot or kEmptyKey
    rank_list = datagen.degree_rank(graph["indptr"])
    t_rank = to_dev(rank_list)
    num_cached = int(N * args.cache_ratio)
    store = None
    if args.store == "replica" or world == 1:
        cache = torch.empty((max(num_cached, 1), dim), dtype=torch.float32, device=dev)
        feat_rows(t_rank[:num_cached], cache)
    else:  # GGMS: cache slot s lives on rank s % world at row s // world
        from xgnn_amd import ggms_store
        cache, holder = ggms_store.shard_rows(feat_rows, t_rank, num_cached, world, rank, dim, torch.float32, dev,
                                              shared=(args.store == "peer"))
    table = torch.full((N,), -1, dtype=torch.int32, device=dev)  # 0xffffffff
    table[t_rank[:num_cached].long()] = torch.arange(num_cached, dtype=torch.int32, device=dev)
    ptab = ops.part_pointer_table([cache], dev)
    host_feat = None
    if num_cached < N:  # host tier: the full table in pinned host memory, read zero-copy by the gather kernel
        host_feat = torch.empty((N, dim), dtype=torch.float32, pin_memory=True)
        feat_rows(torch.arange(N, dtype=torch.int64), host_feat)
    if args.store != "replica" and world > 1:
        store = ggms_store.FeatureShards(cache, table, world, rank, mode=args.store, dist=dist, host_feat=host_feat)
        if args.store == "peer":
            store.connect_peers(holder)

    code = {"khop3": ops.KHOP3, "khop0": ops.KHOP0, "khop2": ops.KHOP2, "khop1": ops.KHOP1,
            "weighted_khop": ops.WEIGHTED_KHOP, "weighted_khop_hash_dedup": ops.WEIGHTED_KHOP_HASH_DEDUP,
            "random_walk": ops.RANDOM_WALK}[args.sample_type]
    extra_kw = {}
    if args.sample_type.startswith("weighted_khop"):  # per-edge acceptance probability + alias neighbour (engine.cc:372-384)
        gen = torch.Generator(device=dev).manual_seed(7)
        E = indices.numel()
        extra_kw = dict(prob_table=torch.rand(E, generator=gen, device=dev, dtype=torch.float32),
                        alias_table=torch.randint(0, N, (E,), generator=gen, device=dev, dtype=torch.int32))
    if args.sample_type == "random_walk":  # common_config.py PinSAGE defaults
        extra_kw = dict(random_walk_length=3, random_walk_restart_prob=0.5, num_random_walk=4)
    # batches in flight: K sampling pipelines (own stream, dedup table, workspace; RNG pool consumed in batch
    # order) + the extract stream; outputs live in batch slots, as in the engine
    K = 1 if args.no_overlap else max(1, args.pipelines)
    NSLOT = K + 1
    sampler = ops.BatchSampler(g, fanouts, args.batch, sample_type=code, seed=0x5EED + rank, device=dev,
                               num_slots=NSLOT, num_pipelines=K, **extra_kw)
    out = [torch.empty((sampler.max_unique, dim), dtype=torch.float32, device=dev) for _ in range(NSLOT)]
    out_label = [torch.empty(sampler.max_seeds, dtype=torch.int64, device=dev) for _ in range(NSLOT)]
    nmiss = torch.zeros(1, dtype=torch.int64, device=dev)
    L = len(fanouts)
    s_samples = [torch.cuda.Stream(device=dev) for _ in range(K)]
    s_extract = s_samples[0] if args.no_overlap else torch.cuda.Stream(device=dev)
    slot_free = [None] * NSLOT  # event: the slot's previous extract has finished

    # DistAlignedShuffler semantics (dist_shuffler_aligned.cc:37-146): pad to a multiple of world,
    # same permutation on every rank, contiguous slice per rank
    train = graph["train_set"]
    steps_per_epoch = parallel.steps_per_epoch(len(train), world, args.batch)
    per_rank = len(parallel.pad_train_set(train, world)) // world

    epoch_cache = {}

    def batch_seeds(step):
        ep, ls = divmod(step, steps_per_epoch)
        if ep not in epoch_cache:
            epoch_cache[ep] = to_dev(parallel.rank_slice(train, world, rank, ep))
        lo = ls * args.batch
        return epoch_cache[ep][lo:min(per_rank, lo + args.batch)]

    # the per-epoch reshuffle + H2D of the rank's slice happens once per epoch in the real loop; keep it
    # out of the timed region (inputs are resident in HBM when timing starts)
    all_seeds = [batch_seeds(s) for s in range(args.warmup + args.steps)]

    acc = torch.zeros(3 * L + 2, dtype=torch.int64, device=dev)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    host_t = [0.0] * 6

    def run_step(step, timed_idx=None):
        seeds = all_seeds[step]
        slot = step % NSLOT
        s_sample = s_samples[step % K]
        h0 = time.perf_counter()
        with torch.cuda.stream(s_sample):
            if slot_free[slot] is not None:
                s_sample.wait_event(slot_free[slot])
            if timed_idx is not None:
                ev[timed_idx][0].record(s_sample)
            h1 = time.perf_counter()
            sampler.sample(seeds, slot=slot, copy_input_nodes=True)
            h2 = time.perf_counter()
            sampled = torch.cuda.Event()
            sampled.record(s_sample)
            if timed_idx is not None:
                ev[timed_idx][1].record(s_sample)
        h3 = time.perf_counter()
        with torch.cuda.stream(s_extract):
            s_extract.wait_event(sampled)
            counts = sampler.counts_slots[slot]
            if timed_idx is not None:
                ev[timed_idx][2].record(s_extract)
            if store is None:
                ops.extract_cached(out[slot], sampler.input_nodes[slot], table, ptab, 0, host_feat,
                                   num=sampler.max_unique, num_dev=counts[3 * L:3 * L + 1], num_miss=nmiss)
            else:  # sharded store: peer loads inside the same kernel, or the all-to-all exchange
                store.extract(sampler.input_nodes[slot], sampler.max_unique, out[slot],
                              num_dev=counts[3 * L:3 * L + 1], num_miss=nmiss)
            if timed_idx is not None:
                ev[timed_idx][3].record(s_extract)
            h4 = time.perf_counter()
            ops.extract(labels, seeds, out=out_label[slot][:seeds.numel()])
            h5 = time.perf_counter()
            acc.add_(counts)
            done = torch.cuda.Event()
            done.record(s_extract)
            slot_free[slot] = done
        h6 = time.perf_counter()
        for i, (a, b) in enumerate([(h0, h1), (h1, h2), (h2, h3), (h3, h4), (h4, h5), (h5, h6)]):
            host_t[i] += b - a

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        run_step(s)
    barrier()
    acc.zero_()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        run_step(args.warmup + k, k)
    barrier()
    elapsed = time.perf_counter() - t0

    if args.host_profile and rank == 0:
        names = ["ev0", "sample", "ev1", "extract_cached", "ev2+label", "acc"]
        print("host enqueue ms/step:", {n: round(1e3 * t / (args.steps + args.warmup), 4) for n, t in zip(names, host_t)},
              file=sys.stderr)
    # the same gather with nothing beside it (one stream), for reference next to the in-pipeline figure
    serial_us = None
    if not args.no_overlap and store is None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        counts = sampler.counts_slots[(args.warmup + args.steps - 1) % NSLOT]
        slot = (args.warmup + args.steps - 1) % NSLOT
        e0.record()
        for _ in range(reps):
            ops.extract_cached(out[slot], sampler.input_nodes[slot], table, ptab, 0, host_feat,
                               num=sampler.max_unique, num_dev=counts[3 * L:3 * L + 1], num_miss=nmiss)
        e1.record()
        torch.cuda.synchronize()
        serial_us = e0.elapsed_time(e1) / reps * 1e3
        serial_rows = int(counts[3 * L].item())
    # self-check outside the timed region: the last batch's rows against the generator's closed form
    last_slot = (args.warmup + args.steps - 1) % NSLOT
    n_last = int(sampler.counts_slots[last_slot][3 * L].item())
    ids = sampler.input_nodes[last_slot][:n_last].to(torch.int64)
    sample_ids = ids[:: max(1, n_last // 4096)]
    want = torch.empty((sample_ids.numel(), dim), dtype=torch.float32, device=dev)
    feat_rows(sample_ids, want)
    rows_ok = bool(torch.equal(out[last_slot][:n_last][:: max(1, n_last // 4096)], want))
    if not rows_ok:
        raise SystemExit("bench self-check failed: gathered rows differ from the feature generator")
    c = acc.cpu().tolist()
    edges = sum(c[3 * i] for i in range(L))
    rows = c[3 * L]
    t_sample_ms = sum(e[0].elapsed_time(e[1]) for e in ev)   # on the sampling stream
    t_extract_ms = sum(e[2].elapsed_time(e[3]) for e in ev)  # HIP events on the stream the gather is launched on

    feat_rate = rows * dim * 4 / (t_extract_ms / 1e3) / 1e9  # this rank's GB/s over its own gather time
    stats = torch.tensor([elapsed, float(edges), float(rows), t_sample_ms, t_extract_ms, feat_rate],
                         dtype=torch.float64, device=dev)
    if world > 1 and backend != "nccl":
        stats = stats.cpu()
    mx, sm = parallel.reduce_stats(stats, dist if world > 1 else None)
    elapsed, edges_all, rows_all, feat_rate_all = mx[0].item(), sm[1].item(), sm[2].item(), sm[5].item()

    if rank == 0:
        row_bytes = dim * 4
        ext_s = t_extract_ms / 1e3
        algo_bytes_per_launch = rows / args.steps * (4 + 2 * row_bytes)
        avg_launch_s = ext_s / args.steps
        achieved = algo_bytes_per_launch / avg_launch_s / 1e9
        tr = measured_traffic()
        traffic = None
        if tr is not None:  # PMC bytes per row (profiled run) x rows of this run / this run's launch time
            traffic = tr["hbm_bytes_per_launch"] / tr["rows_per_launch"] * (rows / args.steps) / avg_launch_s / 1e9
        res = {
            "metric": "sampled edges/s + feature-extract GB/s per epoch-step",
            "value": edges_all / elapsed,
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 ids / f32 rows (bit copy)",
            "data": "synthetic",
            "rows_verified": rows_ok,
            "config": {
                "workload": f"{args.preset}-shaped power-law CSR N={N} E={meta['num_edge']} f32 dim {dim}, "
                            f"GraphSAGE fanout {fanouts} {args.sample_type}, batch {args.batch}, "
                            f"graph in HBM, feature cache_ratio {args.cache_ratio} (rest in pinned host DRAM), "
                            f"seeds DP over {world} GPU(s), feature store: {args.store if world > 1 else 'local'}",
                "global_batch": args.batch * world,
                "parallelism": f"dp{world}",
                "streams": "1 (serial)" if args.no_overlap else
                           f"{K} sampling pipelines (batches in flight, RNG pool consumed in batch order) + 1 extract stream",
            },
            "feature_extract_GBps": feat_rate_all,  # sum over ranks of rows*dim*4 / (time inside the gather kernel)
            "per_gpu": {
                "sample_ms_per_step": t_sample_ms / args.steps,  # latency of one batch on its pipeline (they overlap)
                "extract_ms_per_step": t_extract_ms / args.steps,
                # sampled edges over the time the sampler alone was busy: only meaningful with one pipeline
                "sample_only_edges_per_s": edges / (t_sample_ms / 1e3) if K == 1 else None,
                "edges_per_step": edges / args.steps,
                "rows_per_step": rows / args.steps,
            },
            "roofline": {
                "kernel": "k_gather_rows<16, CachedRows, ident-dst, nt> (ggms_extract_cached)",
                "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                "frac": achieved / 8000.0, "traffic": traffic,
                # same launch with nothing beside it (see roofline_alone): in the pipeline the gather shares HBM with
                # the sampling kernels of the next batches
                "frac_alone": None if serial_us is None else
                serial_rows * (4 + 2 * row_bytes) / (serial_us * 1e-6) / 1e9 / 8000.0,
                "algorithmic_bytes_per_row": 4 + 2 * row_bytes,
                "avg_launch_us": avg_launch_s * 1e6,
                "traffic_source": tr["source"] if tr else None,
            },
            # the step as a whole against the same roof: profiled HBM bytes of one step (gather, corrected, scaled to
            # this run's rows + sampler kernels, raw counters = lower bound) / this run's step time
            "pipeline_hbm": None if not tr or "sampler_hbm_bytes_per_step_raw" not in tr else {
                "bytes_per_step": tr["hbm_bytes_per_launch"] / tr["rows_per_launch"] * (rows / args.steps)
                + tr["sampler_hbm_bytes_per_step_raw"],
                "GBps": (tr["hbm_bytes_per_launch"] / tr["rows_per_launch"] * (rows / args.steps)
                         + tr["sampler_hbm_bytes_per_step_raw"]) / (elapsed / args.steps) / 1e9,
                "frac_of_peak": (tr["hbm_bytes_per_launch"] / tr["rows_per_launch"] * (rows / args.steps)
                                 + tr["sampler_hbm_bytes_per_step_raw"]) / (elapsed / args.steps) / 8e12,
                "source": tr["source"],
            },
            "roofline_alone": None if serial_us is None else {
                "note": "same kernel, last batch's rows, nothing running beside it (10 launches after the timed region)",
                "avg_launch_us": serial_us, "achieved": serial_rows * (4 + 2 * row_bytes) / (serial_us * 1e-6) / 1e9,
                "frac": serial_rows * (4 + 2 * row_bytes) / (serial_us * 1e-6) / 1e9 / 8000.0,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(graph, fanouts, args.batch, dim, args.cpu_seconds)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
