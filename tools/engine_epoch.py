"""One sampling + extract epoch through the reference's operator surface (samgraph.torch: config / init /
sample_once / get_next_batch), timed the way the reference's scripts report it -- the per-epoch log items
kLogEpochSampleTime / kLogEpochCopyTime / kLogEpochFeatureBytes / kLogEpochNumSample (dist_loops.cc:324,361-362,
1276-1281) -- plus the wall clock around the loop.  Prints one JSON line.  bench.py runs this in a child process
for its `engine` sub-record; also usable by hand:

    python tools/engine_epoch.py <dataset_dir> --fanout 5 10 15 --batch-size 8000 --cache-percentage 1.0
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dataset")
    ap.add_argument("--fanout", type=int, nargs="+", default=[5, 10, 15])
    ap.add_argument("--batch-size", type=int, default=8000)
    ap.add_argument("--num-epoch", type=int, default=2, help="the LAST epoch is reported (the first one warms up)")
    ap.add_argument("--sample-type", default="khop3")
    ap.add_argument("--cache-percentage", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--pipelined", action="store_true", help="extract_start(0): the background thread samples ahead")
    ap.add_argument("--replicate-percentage", type=float, default=0.0,
                    help="arch6: this fraction of the cached rows (hottest first) on every GPU, the rest sharded "
                         "(config key replicate_percentage; 0 = pure shards)")
    ap.add_argument("--use-dist-graph", type=float, default=0.0, metavar="FRACTION",
                    help="arch6: GGMS topology shards (config key use_dist_graph, /root/reference README.md:184) -- the leading "
                         "nodes that hold this fraction of the EDGES are sharded over the workers' GPUs (node v in shard "
                         "v %% W at row v / W, peers read over xGMI), every other node is read from the whole CSR in "
                         "registered host memory; 0 = whole CSR on every GPU")
    ap.add_argument("--arch6", type=int, default=0, metavar="WORKERS",
                    help="the multi-GPU deployment instead of arch1: config + data_init here, one forked worker per GPU "
                         "(sample_init / train_init on cuda:<worker>), feature shards across the workers' GPUs "
                         "(part_cache + gpu_extract), every worker samples its slice of the epoch")
    ap.add_argument("--staged-host", action="store_true",
                    help="the host-staged feature path (arch6, one worker, gpu_extract OFF: miss ids to the host, rows "
                         "gathered by --host-threads cores into pinned memory, copied down, scattered; "
                         "dist_loops.cc:1015-1207).  After a few warm-up batches a stretch runs the reference's SERIAL "
                         "sequence with every phase behind its own wait (per-phase rates, as "
                         "study/host-extract-speed-amount/data.dat reports them), then one the chunked pipeline (effective rate).  The zero-filled stand-in table "
                         "is given real pages first (SAMGRAPH_FILL_FAKE_FEAT)")
    ap.add_argument("--host-threads", type=int, default=0, help="--staged-host: omp_thread_num (0 = the cores this process may use)")
    ap.add_argument("--staged-steps", type=int, nargs=2, default=[10, 20], metavar=("SERIAL", "OVERLAPPED"),
                    help="--staged-host: measured batches per stretch (each preceded by --staged-warm warm-up batches)")
    ap.add_argument("--staged-warm", type=int, default=3)
    a = ap.parse_args()
    if a.staged_host:
        return main_staged_host(a)
    if a.arch6:
        return main_arch6(a)
    import samgraph.torch as sam
    cfg = {"dataset_path": a.dataset, "_arch": sam.builtin_archs["arch1"]["arch"], "_sample_type": sam.sample_types[a.sample_type],
           "batch_size": a.batch_size, "num_epoch": a.num_epoch, "_cache_policy": sam.cache_policies["degree"],
           "cache_percentage": a.cache_percentage, "max_sampling_jobs": 10, "max_copying_jobs": 1, "omp_thread_num": 40,
           "num_layer": len(a.fanout), "num_hidden": 256, "lr": 0.003, "dropout": 0.5, "num_fanout": len(a.fanout),
           "fanout": a.fanout, "sampler_ctx": "cuda:0", "trainer_ctx": "cuda:0", "seed": a.seed}
    t0 = time.perf_counter()
    sam.config(cfg)
    sam.init()
    t_init = time.perf_counter() - t0
    steps = sam.steps_per_epoch()
    if a.pipelined:
        sam.extract_start(0)
    out = None
    for e in range(sam.num_epoch()):
        t0 = time.perf_counter()
        for _ in range(steps):
            if not a.pipelined:
                sam.sample_once()
            sam.get_next_batch()
        wall = time.perf_counter() - t0
        ts, tc = sam.get_log_epoch_value(e, sam.kLogEpochSampleTime), sam.get_log_epoch_value(e, sam.kLogEpochCopyTime)
        fb, ns = sam.get_log_epoch_value(e, sam.kLogEpochFeatureBytes), sam.get_log_epoch_value(e, sam.kLogEpochNumSample)
        out = {"steps": steps, "epoch": e, "wall_s": wall, "ms_per_step": wall / steps * 1e3, "edges": ns,
               "edges_per_s": ns / wall, "sample_edges_per_s": ns / ts if ts else None,
               "feature_GBps": fb / tc / 1e9 if tc else None, "feature_bytes": fb,
               "log_items": {"kLogEpochSampleTime": ts, "kLogEpochCopyTime": tc}, "init_s": t_init}
    sam.shutdown()
    print(json.dumps(out), flush=True)


def usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def main_staged_host(a):
    """arch6, ONE worker in this process (no peers to fork for), gpu_extract off.  cache_percentage 0 = BASELINE
    configs[2] the way the reference runs it (every row through the host cores and one PCIe copy); > 0 = the miss path
    of a partial cache (the setting of study/host-extract-speed-amount).  One epoch, three stretches of batches:
    warm-up, the reference's SERIAL sequence (each phase behind its own wait: per-phase rates as data.dat reports
    them), the chunked pipeline (effective rate over the wall clock)."""
    os.environ.setdefault("SAMGRAPH_FILL_FAKE_FEAT", "1")
    import samgraph.torch as sam
    threads = a.host_threads or usable_cores()
    warm, n_serial, n_over = a.staged_warm, a.staged_steps[0], a.staged_steps[1]
    cfg = {"dataset_path": a.dataset, "_arch": sam.builtin_archs["arch6"]["arch"], "_sample_type": sam.sample_types[a.sample_type],
           "batch_size": a.batch_size, "num_epoch": 1, "_cache_policy": sam.cache_policies["degree"],
           "cache_percentage": a.cache_percentage, "max_sampling_jobs": 10, "max_copying_jobs": 1, "omp_thread_num": threads,
           "num_layer": len(a.fanout), "num_hidden": 256, "lr": 0.003, "dropout": 0.5, "num_fanout": len(a.fanout),
           "fanout": a.fanout, "num_worker": 1, "seed": a.seed,
           # the engine counts the batches it ENQUEUES: with the default look-ahead (2 batches beyond the one asked for) the
           # serial stretch's per-phase items are still exactly its own, and the overlapped stretch is measured in steady
           # state (its wall clock enqueues as many batches as it hands out)
           "staged_serial_steps": warm + n_serial}
    t0 = time.perf_counter()
    sam.config(cfg)
    sam.data_init()
    sam.sample_init(0, "cuda:0")
    sam.train_init(0, "cuda:0")
    t_init = time.perf_counter() - t0
    steps = sam.num_local_step()
    assert steps >= 2 * warm + n_serial + n_over, "train set too small for the staged-host stretches"
    warm2 = warm + 2 if steps >= 2 * warm + n_serial + n_over + 2 else warm  # past the batches enqueued ahead in serial mode
    out = {"arch": "arch6, 1 worker, gpu_extract off", "cache_percentage": a.cache_percentage, "host_threads": threads,
           "init_s": t_init}

    def stretch(n):
        keys, t0 = [], time.perf_counter()
        for _ in range(n):
            sam.sample_once()
            keys.append(sam.get_next_batch())
        wall = time.perf_counter() - t0
        item = lambda k: sum(sam.get_log_step_value_by_key(key, k) for key in keys)  # noqa: E731
        return wall, item

    stretch(warm)
    wall, item = stretch(n_serial)  # kLogL3Cache{ExtractMiss, CopyMiss, CombineMiss, CombineCache}Time = 46 .. 49
    miss, feat = item(sam.kLogL1MissBytes), item(sam.kLogL1FeatureBytes)
    g, c, m, h = item(46), item(47), item(48), item(49)
    out["serial"] = {"steps": n_serial, "ms_per_step": wall / n_serial * 1e3, "miss_MB_per_step": miss / n_serial / 1e6,
                     "cpu_gather_GBps": miss / g / 1e9 if g else None, "h2d_GBps": miss / c / 1e9 if c else None,
                     "combine_miss_GBps": miss / m / 1e9 if m and a.cache_percentage > 0 else None,
                     "combine_cache_GBps": (feat - miss) / h / 1e9 if h and feat > miss else None,
                     "effective_GBps": miss / (g + c + m) / 1e9 if g + c + m else None,
                     "phase_ms_per_step": {"cpu_gather": g / n_serial * 1e3, "h2d": c / n_serial * 1e3,
                                           "combine_miss": m / n_serial * 1e3, "combine_cache": h / n_serial * 1e3}}
    stretch(warm2)
    wall, item = stretch(n_over)
    miss, g = item(sam.kLogL1MissBytes), item(46)
    out["overlapped"] = {"steps": n_over, "ms_per_step": wall / n_over * 1e3, "miss_MB_per_step": miss / n_over / 1e6,
                         "cpu_gather_busy_GBps": miss / g / 1e9 if g else None,
                         # miss bytes over the wall clock of the stretch: sampling, split and the hits' combine included
                         "effective_GBps": miss / wall / 1e9,
                         "edges_per_s": item(sam.kLogL1NumSample) / wall}
    sam.shutdown()
    print(json.dumps(out), flush=True)


def main_arch6(a):
    """example/samgraph/sgnn/train_graphsage.py:106-108,136-155,397-412: the parent configures and loads the dataset
    (no GPU), forks one worker per GPU, every worker initialises its sampler + partitioned cache and runs the epochs
    over its slice of the shuffled train set.  Workers meet at the engine's barrier before every epoch, so the epoch's
    wall time is the slowest worker's."""
    import tempfile
    import samgraph.torch as sam
    W = a.arch6
    cfg = {"dataset_path": a.dataset, "_arch": sam.builtin_archs["arch6"]["arch"], "_sample_type": sam.sample_types[a.sample_type],
           "batch_size": a.batch_size, "num_epoch": a.num_epoch, "_cache_policy": sam.cache_policies["degree"],
           "cache_percentage": a.cache_percentage, "max_sampling_jobs": 10, "max_copying_jobs": 1, "omp_thread_num": 40,
           "num_layer": len(a.fanout), "num_hidden": 256, "lr": 0.003, "dropout": 0.5, "num_fanout": len(a.fanout),
           "fanout": a.fanout, "num_worker": W, "part_cache": "True", "gpu_extract": "True", "seed": a.seed}
    if a.replicate_percentage > 0:
        cfg["replicate_percentage"] = a.replicate_percentage
    if a.use_dist_graph > 0:
        cfg["use_dist_graph"] = a.use_dist_graph
    out_dir = tempfile.mkdtemp(prefix="ggms_engine_")
    # data_init probes the node's topology in a forked child (P2P reachability + a timed 128-MiB copy per GPU pair,
    # engine.cc:DetectTopo <- PartitionSolver::DetectTopo): keep its file where this script can read it back
    os.environ.setdefault("SAMGRAPH_TOPO_FILE", os.path.join(out_dir, "detect_topo"))
    sam.config(cfg)
    sam.data_init()  # host only: the GPUs are first touched in the workers
    pids = []
    for w in range(W):
        pid = os.fork()
        if pid == 0:
            code = 1
            try:
                t0 = time.perf_counter()
                sam.sample_init(w, f"cuda:{w}")
                sam.train_init(w, f"cuda:{w}")
                t_init = time.perf_counter() - t0
                steps = sam.num_local_step()
                rec = None
                for e in range(sam.num_epoch()):
                    sam.forward_barrier()
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        sam.sample_once()
                        sam.get_next_batch()
                    wall = time.perf_counter() - t0
                    rec = {"steps": steps, "wall_s": wall, "init_s": t_init,
                           "edges": sam.get_log_epoch_value(e, sam.kLogEpochNumSample),
                           "sample_s": sam.get_log_epoch_value(e, sam.kLogEpochSampleTime),
                           "copy_s": sam.get_log_epoch_value(e, sam.kLogEpochCopyTime),
                           "feature_bytes": sam.get_log_epoch_value(e, sam.kLogEpochFeatureBytes)}
                json.dump(rec, open(os.path.join(out_dir, f"w{w}.json"), "w"))
                sam.shutdown()
                code = 0
            except BaseException as ex:  # noqa: BLE001
                print("worker", w, "failed:", repr(ex), file=sys.stderr, flush=True)
            os._exit(code)
        pids.append(pid)
    bad = sum(sam.wait_one_child() for _ in pids)
    if bad:
        sys.exit(1)
    recs = [json.load(open(os.path.join(out_dir, f"w{w}.json"))) for w in range(W)]
    wall = max(r["wall_s"] for r in recs)
    edges = sum(r["edges"] for r in recs)
    out = {"arch": "arch6", "workers": W, "replicate_percentage": a.replicate_percentage,
           "cache_percentage": a.cache_percentage, "use_dist_graph": a.use_dist_graph, "steps": recs[0]["steps"], "wall_s": wall, "ms_per_step": wall / recs[0]["steps"] * 1e3,
           "edges": edges, "edges_per_s": edges / wall,
           "sample_edges_per_s": sum(r["edges"] / r["sample_s"] for r in recs if r["sample_s"]),
           "feature_GBps": sum(r["feature_bytes"] / r["copy_s"] / 1e9 for r in recs if r["copy_s"]),
           "feature_bytes": sum(r["feature_bytes"] for r in recs), "init_s": max(r["init_s"] for r in recs)}
    out["topology"] = read_topology(os.environ["SAMGRAPH_TOPO_FILE"], W)
    print(json.dumps(out), flush=True)


def read_topology(path, workers):
    """The engine's probe file (the reference's format) -> {"p2p": [[0/1]], "copy_GBps": [[..]]} over the workers' GPUs,
    [reader][owner]; None when the engine did not probe (one worker, the one-GPU rehearsal hook, a failed probe)."""
    import ctypes as C
    from xgnn_amd import _lib
    if not os.path.exists(path):
        return None
    t = _lib.Topology()
    if _lib.lib().ggms_topology_read_host(C.byref(t), path.encode()) != 0:
        return None
    n = min(workers, t.num_device)
    return {"p2p": [[int(t.can_access[i][j]) for j in range(n)] for i in range(n)],
            "copy_GBps": [[round(float(t.copy_GBps[i][j]), 1) for j in range(n)] for i in range(n)],
            "what": "engine.cc:DetectTopo (forked probe child at data_init): hipDeviceCanAccessPeer and a timed 128-MiB "
                    "hipMemcpyAsync INTO row FROM column; the diagonal is a local copy counted read + write"}


if __name__ == "__main__":
    main()
