#!/usr/bin/env python3
"""bench.py -- GGMS hot path on MI355X: sampled edges/s + feature-extract GB/s.

One "step" = one mini-batch of the hot path on ONE GPU, inputs resident in HBM:
    seeds (8000 train nodes) -> k-layer khop3 neighbour sampling with ordered
    dedup/remap (DoGPUSample) -> feature rows of the batch's input nodes gathered
    from the GGMS feature store + label gather.
Default workload = the largest single-GPU configuration of BASELINE.json: papers100M-shaped
synthetic power-law CSR (N 111,059,956, E ~1.62e9, f32 dim 128 = 56.9 GB of features), 3-hop GCN
fanout [5,10,15] (sgnn/train_gcn.py:84), batch 8000, graph + features in HBM (cache_ratio 1.0).
The same line carries, as sub-records measured after the timed region:
    host_tier     BASELINE configs[2]: the same workload with every feature row in pinned host DRAM
                  (cache_ratio 0), read zero-copy by the gather kernel, next to the box's pinned-copy rate;
    cpu_baseline  the reference's CPU sampler / extractor objects on the host cores, same graph.
`--preset products` is BASELINE configs[1] (fanout [25,10]).

N > 1 (driver: torch.distributed.run, one rank per GPU): data parallel over seed mini-batches
(DistAlignedShuffler slices), graph replicated, FEATURES SHARDED across the GPUs (GGMS, `--store peer`:
slot s on GPU s % N, rows read from the owner's HBM inside the gather kernel over xGMI); the line also
reports the replicated store (no xGMI traffic) and the hybrid one (hot prefix replicated, tail sharded).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
T0 = time.perf_counter()  # the wall-clock budget of the optional sub-records counts from here (--budget-s)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=3,
                    help="the K timed steps are run this many times (each block bracketed by barrier + synchronize); "
                         "the line reports the MEDIAN block (a block is a few ms: one hiccup must not move the headline)")
    ap.add_argument("--preset", default="papers100M")
    ap.add_argument("--batch", type=int, default=8000)
    ap.add_argument("--fanout", default=None, help="default: 5,10,15 (papers100M / friendster), 25,10 (products)")
    ap.add_argument("--sample-type", default="khop3",
                    choices=["khop3", "khop0", "khop2", "khop1", "weighted_khop", "weighted_khop_prefix",
                             "weighted_khop_hash_dedup", "random_walk"],
                    help="random_walk: PinSAGE defaults (walk length 3, restart 0.5, 4 walks); --fanout gives the "
                         "top-K per layer, e.g. 5,5,5.  weighted_khop: alias tables built from synthetic weights")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-tier", action="store_true", help="skip the cache_ratio 0 sub-record (configs[2])")
    ap.add_argument("--no-engine", action="store_true", help="skip the `engine` sub-record (samgraph.torch surface, child process)")
    ap.add_argument("--no-configs", action="store_true",
                    help="N = 1: skip the `configs` sub-record (the other single-GPU BASELINE configurations, each one block "
                         "of this script in a child process: products GraphSAGE [25,10] = configs[1], Friendster-scale PinSAGE)")
    ap.add_argument("--no-staged-host", action="store_true",
                    help="N = 1: skip `host_tier.staged` (cache 0 / 0.64 through the engine's host-staged path, gpu_extract off)")
    ap.add_argument("--no-xgmi-probe", action="store_true", help="N > 1: skip the `xgmi` sub-record (link probe)")
    ap.add_argument("--engine-timeout", type=float, default=300.0,
                    help="upper bound of the engine child's run time; the wall-clock budget may leave it less")
    ap.add_argument("--budget-s", type=float, default=400.0,
                    help="wall-clock budget of the whole run, counted from process start.  The headline line is printed as "
                         "soon as the main region is measured; every optional sub-record (sampler roofline, other stores, "
                         "engine, host tier, CPU baseline) runs only if its estimated time still fits, else it is recorded "
                         "as {\"skipped\": \"budget\"}; the engine child's timeout is what the budget leaves")
    ap.add_argument("--dist-graph", type=float, default=None, metavar="FRACTION",
                    help="XGNN mode's topology (use_dist_graph, /root/reference README.md:184): the leading nodes holding this "
                         "fraction of the edges in topology shards (node v in shard v %% P at row v / P), the rest read from "
                         "the whole CSR in registered host memory.  N = 1: the timed sampler runs through --topology-shards "
                         "LOGICAL shards in this process; N > 1: one shard per rank's GPU, peers mapped with hipIpc and read "
                         "in-kernel over xGMI -- in the main region AND in the `engine` record (arch6 + use_dist_graph)")
    ap.add_argument("--topology-shards", type=int, default=2, help="N = 1 with --dist-graph: logical shards (<= 8)")
    ap.add_argument("--no-xgnn-mode", action="store_true",
                    help="N > 1 without --dist-graph: skip the `xgnn_mode` sub-record (one block with the topology sharded "
                         "over the GPUs as well, use_dist_graph 1.0)")
    ap.add_argument("--host-indptr", action="store_true",
                    help="N = 1 with --dist-graph < 1: keep the host slot's indptr in host memory too (the reference's layout: "
                         "two dependent PCIe round trips per uncached seed); default: the whole indptr stays in HBM (4 B per "
                         "node) and only the uncached nodes' neighbour lists are read over PCIe, as the engine does")
    ap.add_argument("--no-sampler-roofline", action="store_true",
                    help="skip the sampler-alone timing and the memory-side rate probe (roofline_sampler)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline: keep sampling mini-batches this long")
    ap.add_argument("--host-steps", type=int, default=10, help="timed steps of the host-tier sub-record")
    ap.add_argument("--host-profile", action="store_true", help="print host enqueue time per section to stderr")
    ap.add_argument("--pipelines", type=int, default=None,
                    help="sampling batches in flight (each on its own stream with its own dedup table).  Default: 2 for "
                         "khop0, whose generator is re-seeded per launch -- its batches do not have to consume a shared "
                         "generator pool in order, so a second one in flight fills the first one's latency chains "
                         "(profiles/r03_ab_pipelines_products.txt); every other sampler: 1 or 2, whichever a 10-step trial "
                         "before the timed region finds faster (config.pipelines_trial)")
    ap.add_argument("--extract-streams", type=int, default=None, choices=[1, 2],
                    help="streams the feature gathers are issued on: with 2, consecutive batches' gathers alternate and may "
                         "overlap (no wait packet between two gathers, one gather's head fills the other's tail).  Default: "
                         "tried before the timed region like --pipelines (config.pipelines_trial)")
    ap.add_argument("--slots", type=int, default=0,
                    help="batch slots (a batch's outputs stay valid until its extract is done); default pipelines + 1")
    ap.add_argument("--heavy-after-gather", action="store_true",
                    help="the last (largest) layer's sampler launch of batch k+1 waits for the gather of batch k: the two "
                         "fabric-heaviest kernels run one after the other, the smaller layers still overlap the gather")
    ap.add_argument("--no-distinct-seeds", action="store_true",
                    help="do not tell the sampler that a batch's seeds are distinct (they are: slices of a shuffled train set): "
                         "the seeds then take the general insert / ordered scan / look-up launches (A/B hook)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="one stream: extract of batch k and sampling of batch k+1 run back to back "
                         "(default: two streams, the HBM-bound gather overlaps the latency-bound sampler)")
    ap.add_argument("--store", default=None, choices=["replica", "peer", "a2a", "hybrid"],
                    help="N > 1 (default hybrid): 'hybrid' = GGMS placement under a per-GPU HBM budget -- the hottest rows "
                         "(degree rank) on every GPU, as many as --hbm-budget-gb holds beside this GPU's share of the "
                         "sharded rest (--replicate-frac auto; what the reference's PartitionSolver decides on NVLink), "
                         "one gather kernel serves replica, local shard and peer shards (xGMI, hipIpc); 'peer' = the "
                         "budget-minimum case: pure feature shards (slot %% N), 7/8 of the rows read over xGMI inside the "
                         "gather; 'a2a' = same shards, rows exchanged with RCCL all-to-all; 'replica' = every GPU holds "
                         "all cached rows (DP over seeds only)")
    ap.add_argument("--replicate-frac", default="auto",
                    help="hybrid store: 'auto' (default) = the largest hot prefix whose replica + this GPU's share of the "
                         "sharded rest fits --hbm-budget-gb per GPU (ggms_store.plan_replication), or a fraction of the "
                         "cached rows")
    ap.add_argument("--hbm-budget-gb", type=float, default=48.0,
                    help="hybrid store with --replicate-frac auto: HBM one GPU may spend on feature rows (default 48 of "
                         "the 288 GB: papers100M's 57-GB table is then 0.82 replicated, the rest sharded)")
    ap.add_argument("--other-stores", default="peer,replica",
                    help="N > 1: stores measured after the main timed region (one block) and reported under 'stores'")
    ap.add_argument("--neighbour-skew", type=float, default=0.0,
                    help="0 (default, SURVEY 8d): neighbour ids uniform; s > 0: a neighbour is drawn with probability "
                         "proportional to in-degree^s (hub-heavy frontiers, as real power-law graphs have)")
    ap.add_argument("--cache-ratio", type=float, default=1.0,
                    help="fraction of feature rows (by degree rank) resident in HBM; the rest is gathered from "
                         "pinned host memory by the same kernel (GGMS host tier)")
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


def measured_traffic(preset):
    """HBM traffic of the extract kernel from the committed PMC profile (profiles/*_extract_traffic_<preset>.json):
    collected with separate `rocprofv3 --pmc` passes of this same command, FETCH_SIZE corrected x2."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_extract_traffic_{preset}.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    d["source"] = os.path.relpath(files[-1], ROOT)
    # the counters describe the gather kernel as it was when they were collected: the profile carries the sha256 of
    # extract.hip at that time (tools/summarize_profiles.py); a different kernel source voids them
    import hashlib
    now = hashlib.sha256(open(os.path.join(ROOT, "xgnn_amd", "csrc", "extract.hip"), "rb").read()).hexdigest()
    d["stale"] = d.get("extract_hip_sha256") != now
    return d


def measured_sampler_traffic(preset):
    """HBM-side bytes of the sampler chain per sampled edge from the committed PMC passes
    (profiles/*_sampler_traffic_<preset>.json, tools/summarize_profiles.py); void when the sampler sources changed."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_sampler_traffic_{preset}.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    d["source"] = os.path.relpath(files[-1], ROOT)
    csrc = os.path.join(ROOT, "xgnn_amd", "csrc")
    now = hashlib.sha256(b"".join(open(os.path.join(csrc, f), "rb").read() for f in d.get("sources", []))).hexdigest()
    d["stale"] = d.get("sources_sha256") != now
    return d


def cpu_baseline(graph, fanouts, batch, feat, seconds):
    """CPU leg on the same graph: the reference's own CPU leaves (oracle/_ref: CPUSampleKHop0, CPUExtract) where
    shipped, else the oracle's port; dedup/remap = the oracle's OpenMP restatement of CPUHashTable2 (the reference's
    default table) -- every leg on all usable host cores.  Bounded: mini-batches for about `seconds` of CPU work."""
    import oracle
    cores = usable_cores()
    ip, ix, train = graph["indptr"], graph["indices"], graph["train_set"]
    n_node = ip.size - 1
    have_ref = oracle.ref_lib() is not None
    sample = oracle.ref_cpu_sample_khop0 if have_ref else oracle.cpu_sample_khop0
    extract = oracle.ref_cpu_extract if have_ref else oracle.extract
    t_sample = t_remap = t_extract = 0.0
    edges = rows = 0
    n_batches = 0
    per_epoch = max(1, len(train) // batch)
    ht = oracle.CpuHashTable2(n_node, cores)
    t_begin = time.perf_counter()
    while time.perf_counter() - t_begin < seconds or n_batches < 3:
        b = n_batches % per_epoch
        if b == 0 and n_batches:
            train = train[np.random.RandomState(n_batches).permutation(len(train))]
        n_batches += 1
        seeds = train[b * batch:(b + 1) * batch]
        t0 = time.perf_counter()
        ht.reset()
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
    dim = feat.shape[1]
    return {
        "value": edges / total, "unit": "edges/s", "cores": cores,
        # "reference" would claim every leg is reference code; the dedup/remap leg is a port
        "kind": "port",
        "legs": {"sample": "reference CPUSampleKHop0 (oracle/_ref)" if have_ref else "oracle port of CPUSampleKHop0",
                 "remap": "oracle OpenMP port of CPUHashTable2 (Populate/MapEdges/Reset, cpu_hashtable2.cc:53-191)",
                 "extract": "reference CPUExtract (oracle/_ref)" if have_ref else "oracle port of CPUExtract"},
        "sample": f"{n_batches} mini-batches of {batch} seeds, fanout {fanouts}, same synthetic graph and feature table; "
                  f"all three legs on {cores} threads",
        "seconds": {"sample": t_sample, "remap": t_remap, "extract": t_extract, "total": total},
        "sample_only_edges_per_s": edges / t_sample,
        "sample_plus_remap_edges_per_s": edges / (t_sample + t_remap),
        "feature_GBps": rows * dim * 4 / t_extract / 1e9,
    }


def slots_for(pipelines, extract_streams):
    """Batch slots for `pipelines` sampling pipelines and `extract_streams` extract streams: one per batch that can be
    in flight + one spare (the sampler may run one batch further ahead: - 1 % on the default workload,
    profiles/r05_ab_slots_two_streams.txt), rounded up to a multiple of the pipelines -- a slot's next user then runs
    on the sampling stream that gathered the slot's labels and needs no cross-stream wait for them."""
    return (pipelines + extract_streams + 1 + pipelines - 1) // pipelines * pipelines


def choose_streams(ms_per_step, margin=0.02):
    """The streams trial's verdict.  ms_per_step: {(pipelines, extract streams): ms} in candidate order, baseline first.
    The baseline stays unless the fastest candidate is at least `margin` faster."""
    cands = list(ms_per_step)
    best = min(cands, key=ms_per_step.get)
    return best if ms_per_step[best] < (1.0 - margin) * ms_per_step[cands[0]] else cands[0]


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) started as a plain script: become the launcher of N ranks, one per GPU --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py <same flags>` as a
    child process, BEFORE this process has touched the GPU (it never does) -- the reference's own model: the parent
    forks one worker per GPU after data_init (operation.cc:509-533, sgnn/train_graphsage.py:136-155).  Relays the
    ranks' stdout (rank 0's JSON line), exits with the launcher's code, and exits non-zero if no line with
    n_gpus == N came back: this script never reports a GPU count other than the ranks that really ran."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    # the host driver only supports dmabuf IPC: without it hipIpcGetMemHandle fails ("Environment")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    seen = None
    for line in p.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
        if line.startswith("{"):
            try:
                seen = json.loads(line).get("n_gpus")
            except ValueError:
                pass
    rc = p.wait()
    if rc == 0 and seen != args.gpus:
        print(f"bench.py: launched {args.gpus} ranks but the line reports n_gpus = {seen}", file=sys.stderr)
        rc = 1
    sys.exit(rc)


def scratch_dir(need_bytes, prefix):
    """A directory for a few GB of short-lived files: memory-backed /dev/shm when it has the room (a container may
    cap it at 64 MB), else the temp directory."""
    import shutil
    import tempfile
    for base in ("/dev/shm", tempfile.gettempdir()):
        try:
            if os.path.isdir(base) and shutil.disk_usage(base).free > 1.2 * need_bytes:
                return tempfile.mkdtemp(prefix=prefix, dir=base)
        except OSError:
            pass
    return tempfile.mkdtemp(prefix=prefix)


class DatasetDir:
    """The run's graph in the reference's on-disk format (without feat.bin / label.bin: the loader then maps zero-filled
    tables, engine.cc:LoadDataset), written ONCE on first use -- the engine children of several sub-records share it."""

    def __init__(self, datagen, graph, log):
        self.datagen, self.graph, self.log, self.dir = datagen, graph, log, None

    def path(self):
        if self.dir is None:
            g = self.graph
            self.dir = scratch_dir(g["indptr"].nbytes + g["indices"].nbytes + 8 * g["indptr"].size, "ggms_bench_ds_")
            t0 = time.perf_counter()
            self.datagen.write_dataset(self.dir, g, minimal=True)
            self.log(f"engine: dataset written in {time.perf_counter() - t0:.1f} s")
        return self.dir

    def close(self):
        if self.dir is not None:
            import shutil
            shutil.rmtree(self.dir, ignore_errors=True)
            self.dir = None


def child_json(cmd, timeout, env=None, what="child"):
    """Run a child process under a timeout; -> its last JSON line, or {"error": ...} (a failure is recorded, never
    propagated: a sub-record must not cost the line its headline)."""
    import subprocess
    env = dict(os.environ if env is None else env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)  # the child is not a rank of this job
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    except subprocess.TimeoutExpired:
        return {"error": f"{what} exceeded the {timeout:.0f} s the wall-clock budget left it (--budget-s)"}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        err = [l for l in r.stderr.splitlines() if "FATAL" in l or "Error" in l or "error" in l or "failed" in l]
        return {"error": f"{what} rc {r.returncode}: " + (" | ".join(err[:3]) if err else r.stderr[-300:])[:600]}
    try:
        return json.loads(lines[-1])
    except ValueError as e:
        return {"error": f"{what}: unparsable line ({e})"}


def engine_record(ds, fanouts, args, log, workers=0, force_device=None, replicate=0.0, timeout=None, dist_graph=0.0):
    """The same workload through the operator surface north_star names (samgraph.torch: config / init / sample_once /
    get_next_batch) in a CHILD process: the graph is written in the reference's on-disk format without feat.bin /
    label.bin (the loader then maps zero-filled tables, engine.cc:199-235 -- topology and sizes are the real ones), one
    warm-up epoch, one reported epoch; rates from the reference's own log items (tools/engine_epoch.py).
    workers > 0: the multi-GPU deployment (arch6) -- the child forks one engine worker per GPU, feature shards across
    the workers' GPUs behind hipIpc (part_cache + gpu_extract), every worker samples its slice of the epoch."""
    d = ds.path()
    env = dict(os.environ)
    env.setdefault("SAMGRAPH_IPC_TIMEOUT_S", "90")  # a worker that cannot reach its peers ends the child, not the bench
    if force_device is not None:  # one-GPU rehearsal: every engine worker on that device
        env["SAMGRAPH_FORCE_DEVICE"] = str(force_device)
    timeout = args.engine_timeout if timeout is None else timeout
    cmd = ([sys.executable, os.path.join(ROOT, "tools", "engine_epoch.py"), d, "--fanout"] + [str(f) for f in fanouts]
           + ["--batch-size", str(args.batch), "--sample-type", args.sample_type, "--cache-percentage", "1.0"]
           + (["--arch6", str(workers), "--replicate-percentage", f"{replicate:.6f}"] if workers else [])
           + (["--use-dist-graph", f"{dist_graph:.6f}"] if workers and dist_graph else []))
    if os.environ.get("GGMS_BENCH_TEST_ENGINE_SLEEP"):  # test hook: a child that outlives whatever it is given
        cmd = [sys.executable, "-c", f"import time; time.sleep({float(os.environ['GGMS_BENCH_TEST_ENGINE_SLEEP'])})"]
    e = child_json(cmd, timeout, env, what="engine child")
    if "error" in e:
        return e
    e["surface"] = ("samgraph.torch config / init / sample_once / get_next_batch (arch1, cache_percentage 1.0) on the "
                    "same graph written to disk in the reference's format, zero-filled feature table; second epoch; "
                    "sample_edges_per_s and feature_GBps from kLogEpochNumSample / kLogEpochSampleTime and "
                    "kLogEpochFeatureBytes / kLogEpochCopyTime" if not workers else
                    f"samgraph.torch arch6: data_init in the parent, {workers} forked workers (sample_init / train_init / "
                    "sample_once / get_next_batch), feature table sharded over the workers' GPUs (part_cache, gpu_extract, "
                    "hipIpc peers"
                    + (f"; the {replicate:.2f} hottest of the rows on every GPU (replicate_percentage), the rest sharded"
                       if replicate else "")
                    + "), "
                    + (f"topology sharded over the workers' GPUs (use_dist_graph {dist_graph:g}: the leading nodes holding "
                       "that fraction of the edges, peers over hipIpc; the rest from the whole CSR in registered host "
                       "memory)" if dist_graph else "whole CSR on every GPU")
                    + "; second epoch, the slowest worker's wall time; rates summed "
                    "over the workers from the reference's log items")
    return e


def shared_graph(datagen, args, world, local_rank, dist):
    """The synthetic CSR of this run.  N > 1: generated ONCE per node (local rank 0, numpy) and handed to the other
    ranks through /dev/shm as read-only mappings -- the reference's parent loads the dataset once into shared
    memory before it forks its workers (engine.cc:109-180) -- instead of N generations and N host copies."""
    kw = dict(seed=42, neighbour_skew=args.neighbour_skew)
    if world == 1:
        return datagen.make_graph(args.preset, **kw)
    import shutil
    g = d = None
    try:
        if local_rank == 0:
            g = datagen.make_graph(args.preset, **kw)
            d = scratch_dir(g["indptr"].nbytes + g["indices"].nbytes, "ggms_bench_graph_")
            for k in ("indptr", "indices", "train_set"):
                np.save(os.path.join(d, k + ".npy"), g[k])
            json.dump(g["meta"], open(os.path.join(d, "meta.json"), "w"))
        where = [d]
        dist.broadcast_object_list(where, src=0)  # also the meeting point: the files are complete
        d = where[0]
        if local_rank != 0:
            g = {k: np.load(os.path.join(d, k + ".npy"), mmap_mode="r") for k in ("indptr", "indices", "train_set")}
            g["meta"] = json.load(open(os.path.join(d, "meta.json")))
        dist.barrier()  # everybody holds its mappings: the files can go
    finally:
        if local_rank == 0 and d:
            shutil.rmtree(d, ignore_errors=True)
    return g


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # never measure one thing and print another
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before the first HIP call of this process

    import torch
    import torch.distributed as dist

    # one rank per GPU (counting devices does not initialise the GPU); the one-GPU rehearsal hook is explicit
    if "GGMS_BENCH_DEVICE" not in os.environ and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} needs {world} GPUs, this node shows {torch.cuda.device_count()} "
                         "(one-GPU rehearsal: GGMS_BENCH_DEVICE=0 GGMS_BENCH_BACKEND=gloo)")
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

    from xgnn_amd import datagen, ggms_store, ops, parallel

    # ---- N > 1, before anything is placed: can every GPU of the group reach every other (hipDeviceCanAccessPeer per pair,
    # PartitionSolver::DetectTopo_child, cuda/dist_graph.cu:812-818)?  Every rank holds the same matrix and takes the same
    # turn: the sharded stores need it, whole-table replicas do not.
    preflight = None
    if world > 1:
        preflight = ggms_store.peer_access_preflight(world, rank, dist, dev_index, ggms_store.HipProbeLeaf(dev))
        if preflight["refused"] and args.dist_graph is not None:
            raise SystemExit(f"bench.py: --dist-graph needs P2P access between every pair of GPUs; refused [reader, owner]: "
                             f"{preflight['refused']}")
    # ---- N > 1: what the xGMI links carry, measured the way the stores use them (ggms_store.link_probe: per pair alone,
    # then every rank from all its peers at once) -- BEFORE the store is placed: the planned placement sizes its sharded
    # tail by it (ggms_store.plan_with_links, the role of the reference's PartitionSolver), and the line quotes every
    # store's xGMI bytes against it.  A few seconds; a refused mapping is a verdict of every rank.
    xgmi = None
    if world > 1 and not args.no_xgmi_probe and not preflight["refused"]:
        try:
            xgmi = ggms_store.link_probe(world, rank, dist, ggms_store.HipProbeLeaf(dev))
        except (RuntimeError, MemoryError) as e:  # PeerConnectError: raised on every rank alike
            xgmi = {"error": f"{type(e).__name__}: {str(e)[:300]}"}

    t_start = time.perf_counter()
    if os.environ.get("GGMS_BENCH_VERBOSE"):  # where is a stuck rank?  Python stacks every 2 minutes
        import faulthandler
        faulthandler.dump_traceback_later(120, repeat=True, file=sys.stderr)

    def log(msg):  # GGMS_BENCH_VERBOSE=1: phase marks on stderr (which phase a slow or stuck run is in)
        if os.environ.get("GGMS_BENCH_VERBOSE"):
            print(f"[bench r{rank} +{time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    if args.fanout is None:
        args.fanout = "25,10" if args.preset in ("products", "tiny") else "5,10,15"
    fanouts = [int(x) for x in args.fanout.split(",")]
    main_store = (args.store or "hybrid") if world > 1 else "local"
    graph = shared_graph(datagen, args, world, local_rank, dist)
    log("graph generated")
    meta = graph["meta"]
    N, dim = meta["num_node"], meta["feat_dim"]
    row_bytes = dim * 4

    def to_dev(a):
        return torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(dev)

    indptr, indices = to_dev(graph["indptr"]), to_dev(graph["indices"])
    g = ops.DeviceGraph(indptr, indices)
    topo_text, topo_record, topo_keep = "graph in HBM", None, None
    if args.dist_graph is not None and world == 1:
        # XGNN mode's graph view (DeviceDistGraph, cuda/dist_graph.h:114-158) in ONE process: P logical shards in HBM +
        # the whole CSR in registered host memory for the nodes beyond num_cache_node
        P = max(1, min(8, args.topology_shards))
        ncn = ggms_store.num_cache_node_for(graph["indptr"], args.dist_graph)
        pip, pix = ggms_store.topology_shards(indptr, indices, P, ncn)
        if ncn < N:
            # the device copy of the whole `indices` goes: the host slot serves the uncached nodes' neighbour lists
            topo_keep = (ops.RegisteredHost(graph["indptr"], dev) if args.host_indptr else None,
                         ops.RegisteredHost(graph["indices"], dev))
            slot = (topo_keep[0].tensor if args.host_indptr else indptr, topo_keep[1].tensor)
            del g, indices
            if args.host_indptr:
                del indptr
            torch.cuda.empty_cache()
        else:
            slot = (indptr, indices)  # never read: every node is cached
        g = ops.DeviceGraph(None, None, part_indptr=pip + [slot[0]], part_indices=pix + [slot[1]], num_cache_node=ncn)
        topo_record = {"use_dist_graph": args.dist_graph, "logical_shards": P, "num_cache_node": ncn,
                       "cached_node_fraction": ncn / N,
                       "host_slot": (("whole CSR" if args.host_indptr else "neighbour lists of the whole CSR (its indptr stays in HBM)")
                                     + " in hipHostRegister'ed host memory, read zero-copy over PCIe") if ncn < N else None}
        topo_text = (f"graph in {P} logical topology shards in HBM (use_dist_graph {args.dist_graph:g}: the {ncn} leading nodes"
                     + (", every other node read from the whole CSR in registered host memory)" if ncn < N else ")"))
    elif args.dist_graph is not None:
        # XGNN mode across the ranks (DistGraph, cuda/dist_graph.cu:228-385): rank r keeps topology shard r of the leading
        # nodes in ITS HBM and publishes it (hipIpc); every rank's sampling kernels read a peer's list heads and
        # neighbour lists in place over xGMI.  The nodes beyond num_cache_node: indptr in HBM, lists in registered host memory.
        ncn = ggms_store.num_cache_node_for(graph["indptr"], args.dist_graph)
        if ncn < N:
            host_ix = np.array(graph["indices"]) if isinstance(graph["indices"], np.memmap) else graph["indices"]
            topo_keep = (None, ops.RegisteredHost(host_ix, dev))
            slot = (indptr, topo_keep[1].tensor)
        else:
            slot = (indptr, torch.zeros(4, dtype=torch.int32, device=dev))  # never read: every node is cached
        topo = ggms_store.TopologyShards(indptr, indices, world, rank, ncn, dist, slot)
        del g, indices  # the whole CSR's neighbour lists leave this GPU: 1 / world of them stay, in the shard
        torch.cuda.empty_cache()
        g = topo.graph
        topo_keep = (topo, topo_keep)
        topo_record = {"use_dist_graph": args.dist_graph, "shards": world, "num_cache_node": ncn, "cached_node_fraction": ncn / N,
                       "placement": "shard r in rank r's HBM, peers mapped with hipIpc and read in-kernel over xGMI",
                       "host_slot": "neighbour lists of the whole CSR (its indptr stays in HBM) in hipHostRegister'ed host "
                                    "memory, read zero-copy over PCIe" if ncn < N else None}
        topo_text = (f"graph in {world} topology shards, one per GPU, peers read over xGMI (use_dist_graph {args.dist_graph:g}: "
                     f"the {ncn} leading nodes" + (", every other node's list read from registered host memory)" if ncn < N else ")"))
    labels = (torch.arange(N, dtype=torch.int64, device=dev) % meta["num_class"]).contiguous()

    def feat_rows(node_ids, out):
        """feat[i, j] = float((i*dim + j) & 0xFFFF), generated in place for the given node ids."""
        cols = torch.arange(dim, dtype=torch.int64, device=out.device)
        step = 1 << 21
        for lo in range(0, node_ids.numel(), step):
            ids = node_ids[lo:lo + step].to(out.device, torch.int64)
            out[lo:lo + step] = ((ids[:, None] * dim + cols[None, :]) & 0xFFFF).to(torch.float32)

    # ---- the feature store ---------------------------------------------------------------------------------
    # cache_ratio 1.0: every row in HBM, kept in NODE order (slot = node id, no id -> slot table: ggms_extract_cached
    # with table = NULL).  cache_ratio < 1: degree policy -- slot r holds node rank[r] (r < num_cached), table[node] =
    # slot or kEmptyKey, the rest comes from pinned host memory.
    full = args.cache_ratio >= 1.0
    num_cached = N if full else int(N * args.cache_ratio)
    # cache rank list (cache_by_degree.bin: node ids by descending in-degree, ties by id) -- datagen.degree_rank's order,
    # computed on the GPU: the host's stable argsort of 111 M degrees is 15-20 s per rank, the device's a fraction of one
    t_rank = None
    if not (full and world == 1):
        try:
            ip_src = indptr
        except NameError:  # --dist-graph --host-indptr at N = 1 gave the device copy back
            ip_src = to_dev(graph["indptr"])
        ip64 = ip_src.to(torch.int64) & 0xFFFFFFFF
        del ip_src
        t_rank = torch.argsort(ip64[1:] - ip64[:-1], descending=True, stable=True).to(torch.int32)
        del ip64
        torch.cuda.empty_cache()  # the sort's scratch (several GB at this size) goes back to the device
    host_feat = None
    if not full:  # host tier: the full table in pinned host memory, read zero-copy by the gather kernel
        host_feat = torch.empty((N, dim), dtype=torch.float32, pin_memory=True)
        chunk = 1 << 22  # generated on the GPU chunk by chunk and copied down (the CPU would take minutes at this size)
        tmp = torch.empty((min(chunk, N), dim), dtype=torch.float32, device=dev)
        for lo in range(0, N, chunk):
            m = min(chunk, N - lo)
            feat_rows(torch.arange(lo, lo + m, dtype=torch.int64, device=dev), tmp[:m])
            host_feat[lo:lo + m].copy_(tmp[:m])
        del tmp
        log("host tier filled")

    hybrid_plan = {}

    def build_store(kind):
        """-> (extract(nodes, num_max, out, num_dev, counters), what it keeps alive) for one store kind."""
        log(f"building store {kind}")
        keep = {}
        if kind in ("local", "replica"):
            if full:
                cache = torch.empty((N, dim), dtype=torch.float32, device=dev)
                feat_rows(torch.arange(N, dtype=torch.int64, device=dev), cache)
                table = None
            else:
                cache = torch.empty((max(num_cached, 1), dim), dtype=torch.float32, device=dev)
                feat_rows(t_rank[:num_cached], cache)
                table = torch.full((N,), -1, dtype=torch.int32, device=dev)  # 0xffffffff
                table[t_rank[:num_cached].long()] = torch.arange(num_cached, dtype=torch.int32, device=dev)
            ptab = ops.part_pointer_table([cache], dev)
            keep.update(cache=cache, table=table, ptab=ptab)

            def extract(nodes, num_max, out, num_dev, counters):
                # (a full cache has no misses to count: no counter, no memset node in front of every gather)
                ops.extract_cached(out, nodes, table, ptab, 0, host_feat, num=num_max, num_dev=num_dev,
                                   num_miss=None if table is None else counters[0:1])
            return extract, keep
        # sharded kinds.  Everything cached and nothing replicated: slot = node id, no table (as on one GPU).
        # Otherwise slots in degree-rank order (hot first); hybrid replicates the first R of them on every GPU.
        R = 0
        if kind == "hybrid":
            R = (ggms_store.plan_replication(num_cached, row_bytes, world, int(args.hbm_budget_gb * 1e9))
                 if args.replicate_frac == "auto" else int(num_cached * float(args.replicate_frac)))
            link_plan = None
            if args.replicate_frac == "auto" and xgmi and "error" not in xgmi:
                # the probe's rates: what this GPU's inbound links sustained with every rank pulling from all its peers,
                # and the same gather kernel on local HBM (the diagonal); capacity: what every GPU can really hold
                free_b = [None] * world
                dist.all_gather_object(free_b, int(torch.cuda.mem_get_info(dev)[0]))
                local = min(xgmi["per_pair_gather_GBps"][r][r] for r in range(world))
                R, link_plan = ggms_store.plan_with_links(num_cached, row_bytes, world, R, xgmi["inbound_all_peers_min_GBps"],
                                                          local, int(0.6 * min(free_b)))
            R = min(R, num_cached - world)  # keep a sharded tail
            hybrid_plan.update(replicated_rows=int(R), replicated_fraction=R / max(1, num_cached),
                               hbm_budget_gb=args.hbm_budget_gb if args.replicate_frac == "auto" else None,
                               hbm_spent_gb=(R + (num_cached - R + world - 1) // world) * row_bytes / 1e9,
                               **({"link_plan": link_plan} if link_plan else {}))
        if full and R == 0:
            order, table = torch.arange(N, dtype=torch.int64, device=dev), None
        else:
            order = t_rank
            table = torch.full((N,), -1, dtype=torch.int32, device=dev)
            table[order[:num_cached].long()] = torch.arange(num_cached, dtype=torch.int32, device=dev)
        replica = None
        if R:
            replica = torch.empty((R, dim), dtype=torch.float32, device=dev)
            feat_rows(order[:R], replica)
        try:
            shard, holder = ggms_store.shard_rows(feat_rows, order[R:], num_cached - R, world, rank, dim, torch.float32,
                                                  dev, shared=(kind != "a2a"))
        except (RuntimeError, MemoryError) as e:
            if kind == "a2a":
                raise
            # this rank cannot build its shard (memory): the peers are about to exchange handles -- walk through the same
            # collectives with the reason, so that every rank raises PeerConnectError together (and takes the same turn)
            replica = table = None
            torch.cuda.empty_cache()
            ggms_store.connect_shared(ggms_store.FailedShard(f"{type(e).__name__}: {e}"), world, rank, dist,
                                      what="feature shard")
            raise  # not reached: the verdict above raises on every rank
        st = ggms_store.FeatureShards(shard, table, world, rank, mode="a2a" if kind == "a2a" else "peer", dist=dist,
                                      host_feat=host_feat, replica=replica)
        log(f"store {kind}: shard filled")
        if kind != "a2a":
            st.connect_peers(holder)
        log(f"store {kind}: peers connected")
        keep.update(store=st, holder=holder, table=table, replica=replica)

        def extract(nodes, num_max, out, num_dev, counters):
            st.extract(nodes, num_max, out, num_dev=num_dev, counters=counters)
        extract.single_launch = kind != "a2a"  # a2a: several gathers and two collectives per batch -- timed with an event pair
        return extract, keep

    code = {"khop3": ops.KHOP3, "khop0": ops.KHOP0, "khop2": ops.KHOP2, "khop1": ops.KHOP1,
            "weighted_khop": ops.WEIGHTED_KHOP, "weighted_khop_prefix": ops.WEIGHTED_KHOP_PREFIX,
            "weighted_khop_hash_dedup": ops.WEIGHTED_KHOP_HASH_DEDUP, "random_walk": ops.RANDOM_WALK}[args.sample_type]
    extra_kw = {}
    if args.sample_type == "weighted_khop_prefix":  # running weight sums per neighbour list (create_prob_prefix_table.cc)
        pre = datagen.build_prob_prefix_table(graph["indptr"], datagen.edge_weights(graph, "default", 7),
                                              num_threads=usable_cores())
        extra_kw = dict(prob_table=torch.from_numpy(pre).to(dev))
    elif args.sample_type.startswith("weighted_khop"):  # per-edge acceptance probability + alias neighbour (engine.cc:372-384)
        # valid alias tables from seeded per-edge weights, as the reference's weight tool builds them
        prob, alias = datagen.build_alias_tables(graph["indptr"], graph["indices"], datagen.edge_weights(graph, "default", 7),
                                                 num_threads=usable_cores())
        extra_kw = dict(prob_table=torch.from_numpy(prob).to(dev), alias_table=to_dev(alias))
    if args.sample_type == "random_walk":  # common_config.py PinSAGE defaults
        extra_kw = dict(random_walk_length=3, random_walk_restart_prob=0.5, num_random_walk=4)
    # batches in flight: K sampling pipelines (own stream, dedup table, workspace; RNG pool consumed in batch
    # order) + the extract stream; outputs live in batch slots, as in the engine
    # Default (no --pipelines / --extract-streams / --slots): (sampling pipelines, extract streams) = (1, 1), (1, 2), (2, 1)
    # -- khop0, whose batches need no shared generator pool: (2, 2), (2, 1) -- are TRIED before the timed region (two blocks
    # of 12 steps each, outside it); the first one stays unless another is at least 2 % faster.  Two sampling chains fill
    # each other's latency gaps on small frontiers (papers100M [25,10]: 0.19 -> 0.165 ms/step) and only add memory-side
    # contention on large ones (profiles/r05_ab_pipelines.txt); two extract streams let consecutive batches' gathers
    # overlap where the gather bounds the step (default workload - 3..5 %, profiles/r05_ab_extract_streams.txt).
    auto_pipes = (args.pipelines is None and args.extract_streams is None and not args.slots and not args.no_overlap)
    khop0 = args.sample_type == "khop0"
    if args.pipelines is None:
        args.pipelines = 2 if (khop0 or auto_pipes) else 1
    K = 1 if args.no_overlap else max(1, args.pipelines)
    # extract streams in use (allocated: two unless --no-overlap / --extract-streams 1)
    n_xs = 1 if args.no_overlap else (args.extract_streams or (2 if auto_pipes else 1))
    # batch slots: one per batch that can be in flight + one spare (the sampler may run one batch further ahead: - 1 % on the
    # default workload, profiles/r05_ab_slots_two_streams.txt); allocated for the largest candidate of the trial
    NSLOT = args.slots if args.slots else (max(slots_for(1, 2), slots_for(2, 1)) if auto_pipes and not khop0 else slots_for(K, n_xs))
    sampler = ops.BatchSampler(g, fanouts, args.batch, sample_type=code, seed=0x5EED + rank, device=dev,
                               num_slots=NSLOT, num_pipelines=K, **extra_kw)
    out = [torch.empty((sampler.max_unique, dim), dtype=torch.float32, device=dev) for _ in range(NSLOT)]
    out_label = [torch.empty(sampler.max_seeds, dtype=torch.int64, device=dev) for _ in range(NSLOT)]
    L = len(fanouts)
    # A/B hook: HIP stream priorities ("sample" / "extract" = that side's streams at high priority); default: none
    prio = os.environ.get("GGMS_BENCH_STREAM_PRIORITY", "")
    s_samples = [torch.cuda.Stream(device=dev, priority=-1 if prio == "sample" else 0) for _ in range(K)]
    s_extract = s_samples[0] if args.no_overlap else torch.cuda.Stream(device=dev, priority=-1 if prio == "extract" else 0)
    # a second extract stream: consecutive batches' gathers alternate and may overlap (profiles/r05_ab_extract_streams.txt)
    s_extracts = [s_extract] + [torch.cuda.Stream(device=dev) for _ in range(n_xs - 1)]
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

    def seeds_distinct(seeds):
        """What the caller of ggms_sample_batch knows about its seeds (ggms_sample_extra_t.seeds_distinct): a slice of a
        shuffled train set is distinct unless both copies of a padding node of the aligned epoch fall into it -- checked
        here, outside every timed region, as the engine checks its own batches."""
        return (not args.no_distinct_seeds) and int(torch.unique(seeds).numel()) == int(seeds.numel())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    host_t = [0.0] * 6
    last_gather = [None]  # completion event of the most recent feature gather (--heavy-after-gather)

    def make_step(extract_fn, counters, accs, seeds_of):
        """One step of the hot path: sample on the batch's pipeline stream (label gather and count totals behind it, same
        stream), the feature gather -- and nothing else -- on the extract stream."""
        def run_step(step, ev4=None, tm=None):
            seeds, distinct = seeds_of(step)
            slot = step % NSLOT
            s_sample = s_samples[step % K]
            h0 = time.perf_counter()
            with torch.cuda.stream(s_sample):
                # the slot's previous rows are out (extract stream) and its labels too (the sampling stream they were gathered
                # on: the slot count is a multiple of K, so that is THIS stream again and needs no wait -- a cross-stream wait
                # per step on a sampling stream is dead time where the sampling streams bound the step: khop0)
                prev_done, prev_labelled, prev_pipe = slot_free[slot] or (None, None, None)
                if prev_done is not None:
                    if isinstance(prev_done, ops.LaunchTimer):
                        prev_done.wait(s_sample)
                    else:
                        s_sample.wait_event(prev_done)
                    if prev_pipe != step % K:
                        s_sample.wait_event(prev_labelled)
                if ev4 is not None:
                    ev4[0].record(s_sample)
                h1 = time.perf_counter()
                sampler.sample(seeds, slot=slot, copy_input_nodes=True, distinct=distinct,
                               heavy_wait=last_gather[0] if args.heavy_after_gather else None)
                h2 = time.perf_counter()
                sampled = torch.cuda.Event()
                sampled.record(s_sample)
                if ev4 is not None:
                    ev4[1].record(s_sample)
                # The label gather (needs the seeds only) and the batch's counts ride BEHIND the batch on its sampling stream,
                # not between two gathers on the extract stream, which bounds the step.  (Not on a stream of their own: HIP
                # streams share 4 hardware queues -- a fifth stream serialised the two khop0 pipelines, 0.36 -> 0.42 ms.)
                ops.extract(labels, seeds, out=out_label[slot][:seeds.numel()])
                accs[step % K].add_(sampler.counts_slots[slot])
                labelled = torch.cuda.Event()
                labelled.record(s_sample)
            h3 = time.perf_counter()
            # (an extract with collectives inside -- the a2a store -- stays on ONE stream: every rank issues them in one order)
            s_extract = s_extracts[step % n_xs] if getattr(extract_fn, "single_launch", True) else s_extracts[0]
            with torch.cuda.stream(s_extract):
                s_extract.wait_event(sampled)
                counts = sampler.counts_slots[slot]
                # The extract stream carries the gather and as little else as possible: every event record / wait is a
                # barrier packet the command processor works through between two gathers (profiles/r05_ab_extract_stream.txt:
                # 28 us of dead time per step with five of them).  The slot's labels are waited for by the NEXT user of the
                # slot on its sampling stream, which has the slack -- not here.  The gather's timing and its "rows are out"
                # event ride on the kernel's own dispatch packet (ggms_launch_timer_t, include/ggms.h); the event-pair form
                # is kept for the store whose extract is several launches (a2a) and as an A/B (GGMS_BENCH_GATHER_TIMING=events).
                if tm is not None:
                    tm.arm()
                    extract_fn(sampler.input_nodes[slot], sampler.max_unique, out[slot], counts[3 * L:3 * L + 1], counters)
                    done = tm
                else:
                    if ev4 is not None:
                        ev4[2].record(s_extract)
                    extract_fn(sampler.input_nodes[slot], sampler.max_unique, out[slot], counts[3 * L:3 * L + 1], counters)
                    done = ev4[3] if ev4 is not None else torch.cuda.Event()
                    done.record(s_extract)
                if args.heavy_after_gather:
                    last_gather[0] = torch.cuda.Event()
                    last_gather[0].record(s_extract)
                h4 = h5 = time.perf_counter()
                slot_free[slot] = (done, labelled, step % K)
            h6 = time.perf_counter()
            for i, (a, b) in enumerate([(h0, h1), (h1, h2), (h2, h3), (h3, h4), (h4, h5), (h5, h6)]):
                host_t[i] += b - a
        return run_step

    step_events = os.environ.get("GGMS_BENCH_STEP_EVENTS", "1") != "0"    # A/B hook: no per-step timing at all
    gather_timing = os.environ.get("GGMS_BENCH_GATHER_TIMING", "timer")   # "events": HIP event pair around the launch (A/B)

    def measure(extract_fn, steps, warmup, repeats, first_step=0):
        """warmup untimed steps, then `repeats` blocks of `steps` timed steps, each bracketed by barrier + synchronize.
        Returns per-block dicts (elapsed = max over ranks) and the step index after the last one."""
        total = warmup + steps * repeats
        seeds_all = [batch_seeds(first_step + s) for s in range(total)]  # the per-epoch reshuffle + H2D stay outside
        seeds_all = [(s, seeds_distinct(s)) for s in seeds_all]
        # rows by tier: [host misses, remote-shard rows, local-shard rows, replica rows]
        counters = torch.zeros(4, dtype=torch.int64, device=dev)
        accs = [torch.zeros(3 * L + 2, dtype=torch.int64, device=dev) for _ in range(K)]  # one per sampling pipeline
        run_step = make_step(extract_fn, counters, accs, lambda s: seeds_all[s])
        log(f"measure: {warmup} warm-up + {repeats} x {steps} steps")
        timed = gather_timing == "timer" and getattr(extract_fn, "single_launch", True)
        slot_tm = [ops.LaunchTimer() for _ in range(NSLOT)] if timed else None
        step_tm = [ops.LaunchTimer() for _ in range(steps)] if timed else None
        for s in range(warmup):
            run_step(s, None, slot_tm[s % NSLOT] if timed else None)
        barrier()
        warm_status = int(sum(a[3 * L + 1] for a in accs).item())  # the warm-up batches' status words (zeroed before each timed block)
        if warmup and warm_status:
            raise SystemExit(f"device status {warm_status} during warm-up")
        log("warm-up done")
        blocks = []
        for r in range(repeats):
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
            barrier()
            for a in accs:
                a.zero_()
            counters.zero_()
            barrier()
            t0 = time.perf_counter()
            for k in range(steps):
                run_step(warmup + r * steps + k, ev[k] if step_events else None,
                         (step_tm[k] if step_events else slot_tm[k % NSLOT]) if timed else None)
            barrier()
            elapsed = time.perf_counter() - t0
            if not step_events:  # A/B hook: what the per-step timing costs the step (the per-stream figures are then bounds)
                for e in ev:
                    for x in e:
                        x.record()
                torch.cuda.synchronize()
            c = torch.stack(accs).sum(0).cpu().tolist()
            if c[3 * L + 1]:
                ops.check_device_status("bench")
                raise SystemExit(f"device status {c[3 * L + 1]} after a timed block")
            edges, rows = sum(c[3 * i] for i in range(L)), c[3 * L]
            inputs = sum(c[3 * i + 2] for i in range(L))  # sampler inputs (seeds of every layer)
            t_sample_ms = sum(e[0].elapsed_time(e[1]) for e in ev)   # on the sampling stream
            if timed and step_events:  # HIP events on the gather's own dispatch packet, on the stream it is launched on
                t_extract_ms = sum(t.elapsed_us() for t in step_tm) / 1e3
                span_ms = step_tm[0].span_us(step_tm[-1]) / 1e3  # first launch's start -> last launch's end
            else:                      # HIP event pair around the launch on that stream
                t_extract_ms = sum(e[2].elapsed_time(e[3]) for e in ev)
                span_ms = ev[0][2].elapsed_time(ev[-1][3]) if step_events else t_extract_ms
            if not step_events:
                t_sample_ms = t_extract_ms = span_ms = elapsed * 1e3
            # gathers on two streams overlap: a launch's own duration then counts time it shares with its neighbour;
            # sum of durations / span = launches in flight on average (below 1 on one stream: the gaps between them)
            in_flight = max(1.0, t_extract_ms / span_ms) if span_ms > 0 else 1.0
            stats = torch.tensor([elapsed, float(edges), float(rows), t_sample_ms, t_extract_ms,
                                  rows * row_bytes / (t_extract_ms / in_flight / 1e3) / 1e9]
                                 + [float(x) for x in counters.cpu().tolist()], dtype=torch.float64, device=dev)
            if world > 1 and backend != "nccl":
                stats = stats.cpu()
            mx, sm = parallel.reduce_stats(stats, dist if world > 1 else None)
            blocks.append(dict(elapsed=mx[0].item(), edges_all=sm[1].item(), rows_all=sm[2].item(), edges=edges,
                               rows=rows, inputs=inputs, t_sample_ms=t_sample_ms, t_extract_ms=t_extract_ms, in_flight=in_flight,
                               feat_rate_all=sm[5].item(), tiers_all=[sm[6 + i].item() for i in range(4)]))
        return blocks, first_step + total

    def median_block(blocks):
        order = sorted(range(len(blocks)), key=lambda i: blocks[i]["elapsed"])
        return blocks[order[(len(blocks) - 1) // 2]]

    # ---- the main timed region -------------------------------------------------------------------------------
    repeats = max(1, args.repeats)
    store_requested, store_error = main_store, None
    try:
        if preflight and preflight["refused"] and main_store in ("peer", "hybrid"):
            raise ggms_store.PeerConnectError("hipDeviceCanAccessPeer refuses [reader, owner] " + str(preflight["refused"])
                                              + ": the sharded stores dereference peer HBM in place")
        extract_main, keep_main = build_store(main_store)
    except ggms_store.PeerConnectError as e:
        # a shard could not be exported / mapped on this node (every rank sees the same verdict and takes the same
        # turn): the line is then measured on whole-table replicas -- the reference's deployment without part_cache --
        # and SAYS so: config.store_requested / store_error, `feature store: replica` in the workload text
        store_error = str(e)[:600]
    if store_error is not None:
        # outside the except block: the exception's traceback holds the failed build's frame -- shard, replica and table,
        # up to the whole HBM budget -- and only once it is gone can the allocator give that memory back
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        print(f"[bench] store {main_store!r} cannot be built: {store_error}\n[bench] measuring on 'replica' instead",
              file=sys.stderr, flush=True)
        main_store = "replica"
        extract_main, keep_main = build_store(main_store)
    main_plan = dict(hybrid_plan) if main_store == "hybrid" else {}
    pipes_trial, first_main = None, 0
    if auto_pipes:
        # every rank sees the same (max-over-ranks) times and takes the same turn
        trial_steps, per = 12, {}
        # (sampling pipelines, extract streams); baseline first -- khop0's is (2, 2): 2-3 % ahead of (2, 1) in every pairing
        # of profiles/r05_ab_extract_streams.txt, which a 2 % margin would decide by the box's noise
        cands = [(2, 2), (2, 1)] if khop0 else [(1, 1), (1, 2), (2, 1)]
        if not getattr(extract_main, "single_launch", True):
            cands = [c for c in cands if c[1] == 1]
        for k, x in cands:
            K, n_xs, NSLOT = k, x, slots_for(k, x)
            sampler.use_pipelines(k)
            slot_free[:] = [None] * len(slot_free)
            tb, first_main = measure(extract_main, trial_steps, 3, 2, first_step=first_main)
            per[(k, x)] = min(b["elapsed"] for b in tb) / trial_steps * 1e3  # the better of two blocks
        K, n_xs = choose_streams(per)
        NSLOT = slots_for(K, n_xs)
        sampler.use_pipelines(K)
        slot_free[:] = [None] * len(slot_free)
        pipes_trial = {"ms_per_step": {f"{k} pipeline(s), {x} extract stream(s)": v for (k, x), v in per.items()},
                       "chosen": {"pipelines": K, "extract_streams": n_xs}, "steps_each": trial_steps, "blocks_each": 2, "warmup_each": 3,
                       "rule": f"two blocks of {trial_steps} steps each before the timed region, the better block counts; the first "
                               "entry unless another is at least 2 % faster"}
        log(f"pipelines trial: {per} -> {K} x {n_xs}")
    blocks, next_step = measure(extract_main, args.steps, args.warmup, repeats, first_step=first_main)
    log("main region done")
    blk = median_block(blocks)
    elapsed, edges_all = blk["elapsed"], blk["edges_all"]
    edges, rows = blk["edges"], blk["rows"]
    if args.host_profile and rank == 0:
        names = ["ev0", "sample", "label+acc", "extract", "wait-label", "done"]
        n_all = args.warmup + args.steps * repeats
        print("host enqueue ms/step:", {n: round(1e3 * t / n_all, 4) for n, t in zip(names, host_t)}, file=sys.stderr)

    # self-check outside the timed region: the rows of the last batch in EVERY slot (with two extract streams the last two
    # gathers ran side by side) against the generator's closed form
    total_steps = args.warmup + args.steps * repeats
    last_slot = (total_steps - 1) % NSLOT
    rows_ok = True
    for back in range(min(NSLOT, total_steps)):
        slot_c = (total_steps - 1 - back) % NSLOT
        n_c = int(sampler.counts_slots[slot_c][3 * L].item())
        ids = sampler.input_nodes[slot_c][:n_c].to(torch.int64)
        stride = max(1, n_c // 4096)
        want = torch.empty((ids[::stride].numel(), dim), dtype=torch.float32, device=dev)
        feat_rows(ids[::stride], want)
        rows_ok = rows_ok and n_c > 0 and bool(torch.equal(out[slot_c][:n_c][::stride], want))
    n_last = int(sampler.counts_slots[last_slot][3 * L].item())
    if not rows_ok:
        raise SystemExit("bench self-check failed: gathered rows differ from the feature generator")

    # the same gather with nothing beside it (one stream), for reference next to the in-pipeline figure
    serial_us = serial_rows = None
    if not args.no_overlap:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        counts = sampler.counts_slots[last_slot]
        scratch = torch.zeros(4, dtype=torch.int64, device=dev)
        barrier()
        e0.record()
        for _ in range(reps):
            extract_main(sampler.input_nodes[last_slot], sampler.max_unique, out[last_slot], counts[3 * L:3 * L + 1], scratch)
        e1.record()
        torch.cuda.synchronize()
        serial_us = e0.elapsed_time(e1) / reps * 1e3
        serial_rows = n_last

    stores_main = None
    if world > 1:
        def store_record(b, kind):
            t = b["tiers_all"]
            remote_rows = t[1] if kind != "a2a" else b["rows_all"] * (world - 1) / world
            return {"edges_per_s": b["edges_all"] / b["elapsed"], "ms_per_step": b["elapsed"] / args.steps * 1e3,
                    "feature_extract_GBps": b["feat_rate_all"], "rows_per_step": b["rows_all"] / args.steps,
                    "remote_row_fraction": remote_rows / max(1.0, b["rows_all"]),
                    "xgmi_bytes_per_step": remote_rows * row_bytes / args.steps,
                    "rows_by_tier": {"host": t[0], "remote_shard": t[1], "local_shard": t[2], "replica": t[3]},
                    **(hybrid_plan if kind == "hybrid" else {})}
        stores_main = {main_store: store_record(blk, main_store)}

    # ---- the headline: complete as soon as the main region, the row check and the gather-alone timing exist ---------
    # It is printed (and flushed) NOW, and again at the end with the sub-records: the last line wins, and a
    # sub-record that is slow, stuck or fatal can no longer cost the run its result (at N = 8 the engine child alone
    # needs a minute of start-up, under a limit somebody else set).
    gather_kernel = {
        "local": "k_gather_rows<16-B chunks, IdentRows, ident-dst> (ggms_extract_cached)" if full else
                 "k_gather_rows<16-B chunks, CachedRows, ident-dst> (ggms_extract_cached)",
        "replica": "k_gather_rows<16-B chunks, IdentRows, ident-dst> (ggms_extract_cached)" if full else
                   "k_gather_rows<16-B chunks, CachedRows, ident-dst> (ggms_extract_cached)",
        "peer": "k_gather_rows<16-B chunks, TieredRows, ident-dst> (ggms_extract_tiered: local shard + peer shards over xGMI)",
        "hybrid": "k_gather_rows<16-B chunks, TieredRows, ident-dst> (ggms_extract_tiered: replica + local shard + peer "
                  "shards over xGMI)",
        "a2a": "k_gather_rows<16-B chunks, PlainRows> at the owner + RCCL all-to-all of ids and rows",
    }[main_store]
    # ---- what THIS box sustains (boxes of one pool differ by a few per cent; the line carries its own yardstick):
    # a device-to-device copy (read + write bytes per second, the guide's 6.29 TB/s figure) and a pinned H2D copy
    box = None
    if rank == 0:
        import ctypes as C
        from xgnn_amd import lib as _lib
        n_copy = 1 << 30
        a_buf = torch.empty(n_copy, dtype=torch.uint8, device=dev)
        b_buf = torch.zeros(n_copy, dtype=torch.uint8, device=dev)
        rate = C.c_double(0)
        cur = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        d2d = None
        for wg_per_cu in (1, 2, 3, 4):  # the copy ceiling of the box = the best grid of the streaming kernel (profiles/r05_copy_grid.txt)
            rc = _lib().ggms_link_probe_copy(C.c_void_p(a_buf.data_ptr()), C.c_void_p(b_buf.data_ptr()), n_copy, 5, wg_per_cu,
                                             C.byref(rate), cur)
            if rc == 0:
                d2d = max(d2d or 0.0, 2.0 * rate.value)
        rc = _lib().ggms_link_probe_copy(C.c_void_p(a_buf.data_ptr()), C.c_void_p(b_buf.data_ptr()), n_copy, 5, 0, C.byref(rate), cur)
        d2d_memcpy = 2.0 * rate.value if rc == 0 else None
        n_pin = 256 << 20
        h_buf = torch.empty(n_pin, dtype=torch.uint8, pin_memory=True)
        rc = _lib().ggms_link_probe_copy(C.c_void_p(a_buf.data_ptr()), C.c_void_p(h_buf.data_ptr()), n_pin, 3, 0, C.byref(rate), cur)
        h2d = rate.value if rc == 0 else None
        del a_buf, b_buf, h_buf
        box = {"d2d_copy_GBps": d2d, "d2d_memcpy_GBps": d2d_memcpy, "pinned_h2d_GBps": h2d,
               "what": "1 GiB device to device, bytes read + written per second: a 16-B-per-lane streaming copy kernel (the "
                       "guide's MI355X figure is 6290) and hipMemcpyAsync; 256 MiB from hipHostMalloc memory; atomics_per_s "
                       "/ loads_per_s are added by the sampler roofline's probe",
               "d2d_over_guide": d2d / 6290.0 if d2d else None}
    res = None
    if rank == 0:
        ext_s = blk["t_extract_ms"] / 1e3
        algo_bytes_per_launch = rows / args.steps * (4 + 2 * row_bytes)
        avg_launch_s = ext_s / args.steps  # a launch's own duration (what rocprof's kernel trace averages)
        # launches in flight on average: 1 on one extract stream; with two, consecutive gathers overlap and every launch's own
        # duration counts the time it shares with its neighbour -- the kernel moves in_flight launches' bytes in that time
        in_flight = blk["in_flight"]
        achieved = algo_bytes_per_launch * in_flight / avg_launch_s / 1e9
        tr = measured_traffic(args.preset)
        traffic = None
        if tr is not None and not tr["stale"] and world == 1 and full:  # (the profiled kernel: one GPU, every row in HBM) PMC bytes per row (profiled run of this command) x rows of this run / this run's launch time
            traffic = tr["hbm_bytes_per_launch"] / tr["rows_per_launch"] * (rows / args.steps) * in_flight / avg_launch_s / 1e9
        elapsed_all = [b["elapsed"] for b in blocks]
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
            "repeats": {"blocks": len(blocks), "reported": "median block of K timed steps",
                        "edges_per_s_each": [b["edges_all"] / b["elapsed"] for b in blocks],
                        "spread": (max(elapsed_all) - min(elapsed_all)) / statistics.median(elapsed_all)},
            "config": {
                "workload": f"{args.preset}-shaped power-law CSR N={N} E={meta['num_edge']} f32 dim {dim}, "
                            f"{'neighbours uniform' if not args.neighbour_skew else f'neighbour skew {args.neighbour_skew} (prob. of a degree-proportional pick)'}, "
                            f"fanout {fanouts} {args.sample_type}, batch {args.batch}, "
                            f"{topo_text}, feature cache_ratio {args.cache_ratio}"
                            f"{(' (all rows in HBM, slots in degree-rank order)' if main_store == 'hybrid' else ' (all rows in HBM, node order)') if full else ' (rest in pinned host DRAM)'}, "
                            f"seeds DP over {world} GPU(s), feature store: {main_store}"
                            + (f" ({main_plan.get('replicated_fraction', 0):.2f} of the cached rows on every GPU"
                               + (f" under a {args.hbm_budget_gb:g}-GB per-GPU budget" if args.replicate_frac == "auto" else "")
                               + (" raised by the link probe's bound (xGMI rows arrive while the local rows stream: "
                                  f"{main_plan['hbm_spent_gb']:.1f} GB spent)"
                                  if (main_plan.get("link_plan") or {}).get("replicated_rows_chosen", 0)
                                  > (main_plan.get("link_plan") or {}).get("replicated_rows_budget_plan", 0) else "")
                               + f", the rest sharded over the {world} GPUs)" if main_store == "hybrid" and world > 1 else ""),
                "global_batch": args.batch * world,
                "parallelism": f"dp{world}",
                "streams": "1 (serial)" if args.no_overlap else
                           f"{K} sampling pipeline(s) (batches in flight, RNG pool consumed in batch order; label gather behind the "
                           f"batch) + {n_xs} extract stream(s) (the feature gathers alone"
                           + ("; consecutive batches' gathers alternate between them and may overlap)" if n_xs > 1 else ")"),
                **({"pipelines_trial": pipes_trial} if pipes_trial else {}),
                "neighbour_skew": args.neighbour_skew,
                "seeds_distinct_promise": not args.no_distinct_seeds,
                **({"topology": topo_record} if topo_record else {}),
                **({"store_requested": store_requested, "store_error": store_error} if store_error else {}),
                "not_covered": "the reference's example scripts themselves were not run (they import DGL, absent from "
                               "this image): the line times the operator surface they call; parity is against the "
                               "oracle, whose four curand_init constants are unverified (no CUDA in the pipeline)",
            },
            "feature_extract_GBps": blk["feat_rate_all"],  # sum over ranks of rows*dim*4 / (time inside the gather kernel)
            "per_gpu": {
                "sample_ms_per_step": blk["t_sample_ms"] / args.steps,  # latency of one batch on its pipeline (they overlap)
                "extract_ms_per_step": blk["t_extract_ms"] / args.steps,
                # sampled edges over the time the sampler alone was busy: only meaningful with one pipeline
                "sample_only_edges_per_s": edges / (blk["t_sample_ms"] / 1e3) if K == 1 else None,
                "edges_per_step": edges / args.steps,
                "rows_per_step": rows / args.steps,
            },
            "roofline": {
                "kernel": gather_kernel,
                "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                "frac": achieved / 8000.0, "traffic": traffic,
                # the READ side alone, rows * (dim * 4 + 4) / t / 8e12 (a gather also writes every byte
                # it reads, so this figure cannot exceed half of what the memory system sustains)
                "hbm_read_frac": rows / args.steps * (row_bytes + 4) * in_flight / avg_launch_s / 8e12,
                # north_star's target is stated on this read-only figure; a gather writes every byte it reads, so the
                # figure cannot pass HALF of what a device-to-device copy sustains on the box (guide: 6.29 of 8 TB/s)
                "hbm_read_frac_target": 0.60,
                "hbm_read_frac_ceiling_for_a_copy": (box["d2d_copy_GBps"] / 2 / 8000.0) if box and box["d2d_copy_GBps"] else 0.393,
                # same launch with nothing beside it: in the pipeline the gather shares HBM with the sampling kernels
                "frac_alone": None if serial_us is None else
                serial_rows * (4 + 2 * row_bytes) / (serial_us * 1e-6) / 1e9 / 8000.0,
                "hbm_read_frac_alone": None if serial_us is None else
                serial_rows * (row_bytes + 4) / (serial_us * 1e-6) / 8e12,
                "algorithmic_bytes_per_row": 4 + 2 * row_bytes,
                "avg_launch_us": avg_launch_s * 1e6,
                # achieved = algorithmic bytes per launch x launches_in_flight / avg_launch_us
                "launches_in_flight": in_flight,
                "extract_streams": n_xs,
                "frac_of_one_launch": algo_bytes_per_launch / avg_launch_s / 1e9 / 8000.0,
                **({"in_flight_note": "two extract streams: consecutive batches' gathers overlap (sum of the launches' own durations "
                                      "/ time from the first launch's start to the last one's end = launches_in_flight); a launch's "
                                      "own duration -- avg_launch_us, rocprof's average -- counts the time it shares with its "
                                      "neighbour, so the kernel's rate is launches_in_flight launches' bytes per avg_launch_us; "
                                      "frac_of_one_launch is what ONE of the overlapping launches gets"} if in_flight > 1.0 else {}),
                "avg_launch_us_alone": serial_us,
                "traffic_source": (tr["source"] + (" (stale: extract.hip changed since; traffic nulled)" if tr["stale"] else
                                                   f" @ extract.hip sha256 {tr['extract_hip_sha256'][:12]}")) if tr else None,
            },
            "box": box,
            # the headline normalised by the box's own copy rate against the guide's figure: comparable across boxes
            "value_over_box": (edges_all / elapsed) / box["d2d_over_guide"] if box and box["d2d_over_guide"] else None,
            "budget": {"budget_s": args.budget_s, "headline_at_s": round(time.perf_counter() - T0, 1)},
        }
        if xgmi is not None:
            res["xgmi"] = xgmi
        if preflight is not None:
            res["peer_access"] = {"devices": preflight["devices"], "can_access": preflight["can_access"],
                                  "refused": preflight["refused"],
                                  "what": "hipDeviceCanAccessPeer [reader][owner], asked by every rank for its own row "
                                          "before anything was placed (ranks sharing a device reach each other)"}
        if stores_main is not None:
            res["stores"] = stores_main
        # N > 1: the headline goes to stdout NOW and the enriched line follows (the last line wins) -- the multi-GPU
        # run is the one whose sub-records (stores, engine child with a minute of start-up) can be slow or stuck under a
        # limit somebody else set.  N = 1 keeps the contract's ONE line on stdout; its early headline goes to stderr.
        print(json.dumps(res), flush=True, file=(sys.stdout if world > 1 else sys.stderr))

    # ---- sub-records: one wall-clock budget for all of them; what no longer fits is skipped and says so -------------
    size_factor = max(0.02, meta["num_edge"] / 1.6e9)  # the default workload = 1

    def left():
        return args.budget_s - (time.perf_counter() - T0)

    def fits(name, need_s):
        """Rank 0 decides, every rank follows (the sub-records hold collectives)."""
        ok = [left() >= need_s]
        if world > 1:
            dist.broadcast_object_list(ok, src=0)
        if not ok[0] and res is not None:
            res[name] = {"skipped": "budget", "needed_s": round(need_s, 1), "left_s": round(left(), 1)}
        log(f"sub-record {name}: {'runs' if ok[0] else 'skipped (budget)'}, {left():.0f} s left")
        return ok[0]

    # ---- N > 1: next to every sharded store's xGMI bytes the remote time they predict: bytes per GPU and step over the
    # inbound rate the probe measured
    def predict_remote(rec):
        """stores.<kind> + the probe: xgmi bytes one GPU pulls per step over the rate its inbound links sustained when every
        rank pulled from all its peers at once -> the time the remote rows alone need (they overlap the local rows
        inside one gather kernel: the step cannot be shorter than this, and need not be longer than local + this)."""
        if not isinstance(rec, dict) or "xgmi_bytes_per_step" not in rec:
            return rec
        per_gpu = rec["xgmi_bytes_per_step"] / world
        rec["xgmi_bytes_per_step_per_gpu"] = per_gpu
        inbound = (xgmi or {}).get("inbound_all_peers_min_GBps")
        if inbound:
            rec["predicted_remote_ms_per_step"] = per_gpu / (inbound * 1e9) * 1e3
            rec["predicted_remote_formula"] = "xgmi_bytes_per_step / n_gpus / xgmi.inbound_all_peers_min_GBps"
        return rec

    if res is not None and "stores" in res:
        for k in list(res["stores"]):
            res["stores"][k] = predict_remote(res["stores"][k])

    # ---- N > 1: the same store with the TOPOLOGY sharded too (XGNN mode, use_dist_graph 1.0), one block -------------
    # The main region keeps the whole CSR on every GPU (6.9 GB of 288: the MI355X-first placement); this block is the
    # reference's defining configuration beside it: rank r keeps topology shard r, every list head and neighbour read of
    # a node in a peer's shard crosses xGMI inside the sampling kernels.
    if world > 1 and args.dist_graph is None and not args.no_xgnn_mode and fits("xgnn_mode", 8 + 14 * size_factor):
        sampler_main, topo, rec = sampler, None, None
        try:
            topo = ggms_store.TopologyShards(indptr, indices, world, rank, N, dist, (indptr, indices))
            sampler = ops.BatchSampler(topo.graph, fanouts, args.batch, sample_type=code, seed=0x5EED + rank, device=dev,
                                       num_slots=NSLOT, num_pipelines=K, **extra_kw)
            bx, next_step = measure(extract_main, args.steps, 2, 1, first_step=next_step)
            rec = {**predict_remote(store_record(bx[0], main_store)), "use_dist_graph": 1.0,
                   "what": f"one block of the main region's workload with the topology in {world} shards, one per GPU "
                           "(node v in shard v % N at row v / N, DeviceDistGraph): peers' list heads and neighbour lists "
                           "read in-kernel over xGMI through hipIpc mappings; feature store as in the main region",
                   "vs_main_edges_per_s": bx[0]["edges_all"] / bx[0]["elapsed"] / (edges_all / elapsed)}
        except (RuntimeError, MemoryError) as e:
            rec = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        sampler = sampler_main
        barrier()  # nobody unmaps a shard a peer's kernels may still be reading
        if topo is not None:
            topo.close()
        topo = None
        if res is not None:
            res["xgnn_mode"] = rec

    # ---- the sampler chain with nothing beside it, and the memory-side ceilings it runs against ----------------
    # (rank 0's GPU; N > 1: every rank does the same work, only rank 0 reports)
    if not args.no_sampler_roofline and fits("roofline_sampler", 4 + 6 * size_factor):
        n_alone = max(4, min(args.steps, 20))
        seeds_alone = [batch_seeds(next_step + k) for k in range(n_alone + 1)]
        distinct_alone = [seeds_distinct(s) for s in seeds_alone]
        next_step += n_alone + 1
        s0 = s_samples[0]
        barrier()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s0):
            sampler.sample(seeds_alone[0], slot=0, copy_input_nodes=True, distinct=distinct_alone[0])  # warm
            a0.record(s0)
            for k in range(n_alone):
                sampler.sample(seeds_alone[1 + k], slot=k % NSLOT, copy_input_nodes=True, distinct=distinct_alone[1 + k])
            a1.record(s0)
        torch.cuda.synchronize()
        sampler_alone_ms = a0.elapsed_time(a1) / n_alone
        # a batch that hit a device-side bound here must not pass silently: every batch reports its own status word
        bad = [int(sampler.counts_slots[k][3 * L + 1].item()) for k in range(NSLOT)]
        if any(bad):
            raise SystemExit(f"device status {bad} in the sampler-alone region")
        # ceilings: random requests on a table the size of the dedup table (one 64-bit word per node id), one launch =
        # one step's worth of edges.  Returning atomicMin / 4-byte load / both per request (ggms_fabric_probe).
        import ctypes as C
        from xgnn_amd import lib as _lib
        words = N
        ptab = torch.full((words,), -1, dtype=torch.int64, device=dev)
        sink = torch.zeros(1, dtype=torch.int32, device=dev)
        reqs = max(1, int(edges / args.steps))
        probe = {"table_bytes": words * 8, "requests_per_launch": reqs}
        salt = 0x7fffff00
        for kind, name in ((0, "atomic"), (1, "load"), (2, "load_atomic_pair")):
            p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for rep in range(5):
                if rep == 1:
                    p0.record()
                rc = _lib().ggms_fabric_probe(kind, C.c_void_p(ptab.data_ptr()), words, reqs, salt, C.c_void_p(sink.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0
                salt -= 1
            p1.record()
            torch.cuda.synchronize()
            probe[name + "s_per_s"] = reqs / (p0.elapsed_time(p1) / 4 / 1e3)
        del ptab
        if res is not None and res.get("box"):
            res["box"].update(atomics_per_s=probe["atomics_per_s"], loads_per_s=probe["loads_per_s"],
                              load_atomic_pairs_per_s=probe["load_atomic_pairs_per_s"])
        if res is not None:
            E_step, S_step = edges / args.steps, blk["inputs"] / args.steps
            algo = 12 * S_step + 28 * E_step  # per seed id + indptr pair; per edge neighbour + bucket + COO
            t = sampler_alone_ms / 1e3
            # request floor: one (neighbour load, dedup atomic) pair per edge + one indptr sector per seed, at the
            # rates this device sustains for exactly those requests (measured above, same process, same table size)
            floor_s = E_step / probe["load_atomic_pairs_per_s"] + S_step / probe["loads_per_s"]
            st = (measured_sampler_traffic(args.preset)
                  if args.sample_type == "khop3" and not args.neighbour_skew and not topo_record else None)
            s_traffic = None
            if st is not None and not st["stale"]:  # PMC bytes per edge (profiled run of this command) x this run's edges / time
                s_traffic = st["hbm_bytes_per_batch"] / st["edges_per_batch"] * E_step / t / 1e9
            res["roofline_sampler"] = {
                "kernel": (f"sampler chain of one batch alone on one stream: k_khop3_fused x {L} (the first also enters the "
                           f"distinct seeds and runs the batch prologue), k_owner_scan_chunked x {L}, k_map_rest_all"
                           if not args.no_distinct_seeds else
                           f"sampler chain of one batch alone on one stream: k_ht_insert, k_khop3_fused x {L}, "
                           f"k_owner_scan_chunked x {L + 1}, k_map_rest_all x 2") if args.sample_type == "khop3" else
                          f"sampler chain of one batch ({args.sample_type}) alone on one stream",
                "bound": "hbm", "achieved": algo / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": algo / t / 8e12,
                # bytes at the L2's memory side (FETCH_SIZE + WRITE_SIZE of the chain's kernels): 64-byte lines for 4-byte
                # reads and 32-byte atomic payloads -- 3.5 x the algorithmic figure; the table line's read / write-back
                # behind every atomic is not in these counters
                "traffic": s_traffic,
                "traffic_source": None if st is None else st["source"] + (" (stale: sampler sources changed since; traffic nulled)"
                                                                          if st["stale"] else ""),
                "algorithmic_bytes_per_step": algo, "alone_ms": sampler_alone_ms,
                "edges_per_step": E_step, "seeds_per_step": S_step, "edges_per_s_alone": E_step / t,
                # the chain is bounded by REQUESTS at the memory side, not by bytes: its atomics and
                # random loads per second against what the device sustains
                "memory_side": {
                    "atomics_per_s": E_step / t, "atomics_per_s_ceiling": probe["atomics_per_s"],
                    "random_loads_per_s": (E_step + S_step) / t, "random_loads_per_s_ceiling": probe["loads_per_s"],
                    "load_atomic_pairs_per_s_ceiling": probe["load_atomic_pairs_per_s"],
                    "probe_table_bytes": probe["table_bytes"], "probe_requests_per_launch": probe["requests_per_launch"],
                    "request_floor_ms": floor_s * 1e3, "chain_over_floor": t / floor_s,
                },
            }

    # ---- N > 1: the other stores, one block each ------------------------------------------------------------------
    if world > 1:
        import gc

        def other_store(kind, first_step):
            """One block on another store; everything it built dies with this frame (the next store, and the engine's
            workers after it, need the HBM)."""
            ex, keep = build_store(kind)
            b2, after = measure(ex, args.steps, 2, 1, first_step=first_step)
            return store_record(b2[0], kind), after

        for kind in [k for k in args.other_stores.split(",") if k and k != main_store]:
            if not fits(f"stores.{kind}", 8 + 35 * size_factor):
                if res is not None:
                    res["stores"][kind] = res.pop(f"stores.{kind}")
                continue
            extract_main = keep_main = None
            gc.collect()
            torch.cuda.empty_cache()
            rec = None
            try:  # a store that cannot be built here (memory) must not cost the line its main result
                rec, next_step = other_store(kind, next_step)
            except (RuntimeError, MemoryError) as e:  # PeerConnectError is one: raised on every rank alike
                rec = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
            if res is not None:
                res["stores"][kind] = predict_remote(rec)
            gc.collect()
            torch.cuda.empty_cache()

    ds_dir = DatasetDir(datagen, graph, log)  # written on first use, shared by the engine children below
    # ---- the same workload through the samgraph.torch surface (child process): arch1 at N = 1, arch6 with N workers ----
    if full and not args.no_engine and args.sample_type.startswith("khop"):
        need = (12 + 25 * size_factor) if world == 1 else (25 + 45 * size_factor + 8 * world * size_factor)
        if fits("engine", need):
            # the child may use what the budget leaves, minus what this process needs to finish and print
            child_timeout = max(5.0, min(args.engine_timeout, left() - 10.0))
            engine = None
            if world == 1:
                engine = engine_record(ds_dir, fanouts, args, log, timeout=child_timeout)
            else:  # rank 0 runs the child (which forks one engine worker per GPU), the other ranks wait
                extract_main = keep_main = None  # the engine's workers build their own shards on these GPUs
                # nothing after this sub-record runs another batch at N > 1: the batch slots, the sampler's tables and the
                # CSR go back to the device too (one GPU holds a rank AND an engine worker; a one-GPU rehearsal two of each)
                out.clear()
                out_label.clear()
                slot_free[:] = []
                sampler = g = indptr = indices = labels = t_rank = None
                import gc
                gc.collect()
                torch.cuda.empty_cache()
                barrier()
                if rank == 0:
                    # the same placement as the main region's: the planned hot prefix on every GPU (0 = pure shards);
                    # --dist-graph: the topology sharded over the workers' GPUs too (XGNN mode, use_dist_graph)
                    engine = engine_record(ds_dir, fanouts, args, log, workers=world,
                                           force_device=os.environ.get("GGMS_BENCH_DEVICE"),
                                           replicate=(main_plan.get("replicated_fraction", 0.0)
                                                      if main_store == "hybrid" else 0.0),
                                           timeout=child_timeout, dist_graph=args.dist_graph or 0.0)
                # the other ranks wait on the HOST (the rendezvous store), not inside an RCCL collective: a collective's
                # kernel would spin on their GPUs for the minute the engine's workers are measuring on them
                try:
                    import datetime
                    store = dist.distributed_c10d._get_default_store()
                    if rank == 0:
                        store.set("ggms_bench_engine_done", "1")
                    else:
                        store.wait(["ggms_bench_engine_done"], datetime.timedelta(seconds=child_timeout + 60))
                except Exception as e:  # noqa: BLE001 -- no such store: the collective below is the meeting point
                    log(f"host-side wait unavailable ({type(e).__name__}: {e}); waiting in the barrier")
                barrier()
            if res is not None and engine is not None:
                if "error" not in engine:  # how far the operator surface is from the headline of this run
                    engine["vs_headline_edges_per_s"] = engine["edges_per_s"] / res["value"]
                res["engine"] = engine
            log("engine sub-record done")

    default_workload = (world == 1 and args.preset == "papers100M" and args.sample_type == "khop3" and full
                        and not topo_record and not args.neighbour_skew and args.fanout == "5,10,15")
    # test hook: treat this run as the default workload and run the `configs` children on this preset (small sizes)
    test_configs = os.environ.get("GGMS_BENCH_TEST_CONFIGS")
    if test_configs and world == 1 and full:
        default_workload = True

    # ---- N = 1: the other single-GPU configurations BASELINE.json names, one block of this script each (child process) ---
    if default_workload and not args.no_configs:
        me = [sys.executable, os.path.abspath(__file__), "--no-host-tier", "--no-cpu-baseline",
              "--no-sampler-roofline", "--no-configs", "--steps", str(args.steps), "--warmup", str(args.warmup),
              "--repeats", str(repeats)]
        res["configs"] = {}
        for name, flags, need, what in (
                ("products_graphsage_25_10", ["--preset", "products"], 25.0,  # (with its `engine` record: seconds at this size)
                 "BASELINE configs[1]: ogbn-products-shaped CSR, GraphSAGE fanout [25,10], graph + features in HBM"),
                ("friendster_pinsage_rw", ["--preset", "friendster", "--sample-type", "random_walk", "--fanout", "5,5,5", "--no-engine"], 30.0,
                 "BASELINE configs[4]'s workload on one GPU: Friendster-scale power-law CSR, 256-dim f32 rows, PinSAGE "
                 "random walk (length 3, restart 0.5, 4 walks, top-5, 3 layers)"),
                ("papers100M_graphsage_25_10", ["--fanout", "25,10", "--no-engine"], 30.0,
                 "BASELINE configs[3]'s workload on one GPU: papers100M-shaped CSR, GraphSAGE fanout [25,10]")):
            if test_configs:
                flags = (["--preset", test_configs, "--batch", str(args.batch)]
                         + (["--sample-type", "random_walk", "--fanout", "5,5,5"] if "pinsage" in name else [])
                         + ([] if name.startswith("products") else ["--no-engine"]))
            if not fits(f"configs.{name}", need):
                res["configs"][name] = res.pop(f"configs.{name}")
                continue
            r = child_json(me + flags, timeout=max(10.0, min(300.0, left() - 10.0)), what=f"bench.py {' '.join(flags)}")
            if "error" not in r:
                rf = r["roofline"]
                r = {"what": what, "workload": r["config"]["workload"], "ms_per_step": r["ms_per_step"], "edges_per_s": r["value"],
                     "feature_extract_GBps": r["feature_extract_GBps"], "gather_frac": rf["frac"],
                     "gather_frac_alone": rf["frac_alone"], "rows_verified": r["rows_verified"],
                     "edges_per_step": r["per_gpu"]["edges_per_step"], "rows_per_step": r["per_gpu"]["rows_per_step"],
                     "spread": r["repeats"]["spread"], "streams": r["config"]["streams"],
                     "pipelines_trial": r["config"].get("pipelines_trial"),
                     # the same workload through the samgraph.torch surface (the child's own `engine` record)
                     **({"engine": {k: v for k, v in r["engine"].items() if k != "surface"}} if isinstance(r.get("engine"), dict) else {})}
            res["configs"][name] = r
        log("configs sub-record done")

    # ---- N = 1: BASELINE configs[2] the way the reference runs it -- the HOST-STAGED path (gpu_extract off): miss ids to
    # the host, rows gathered by the host cores into pinned memory, copied down, scattered (dist_loops.cc:1015-1207,
    # dist_loops_arch6.cc:111-133), through the engine in a child process; cache 0 (every row) and 0.64 (the README's
    # quick-start cache).  Quoted against the only numbers the reference publishes for this path.
    staged = None
    if default_workload and not args.no_host_tier and not args.no_staged_host:
        staged = {"reference_published": {"cpu_gather_GBps": [33.71, 60.21], "h2d_GBps": 23.33, "combine_miss_GBps": 619.77,
                                          "source": "study/host-extract-speed-amount/data.dat:2-16 (papers100M, GraphSAGE "
                                                    "2-hop, degree cache 1-40 %; 8 x V100 box, PCIe 3)"}}
        for cache in (0.0, 0.64):
            name = f"cache_{cache:g}"
            if not fits(f"staged.{name}", 10 + 35 * size_factor):
                staged[name] = res.pop(f"staged.{name}")
                continue
            env = dict(os.environ, SAMGRAPH_IPC_TIMEOUT_S="90")
            n_ep = (len(train) + args.batch - 1) // args.batch  # batches of the (single) epoch the child has to spend
            st_warm, st_serial, st_over = (3, 10, 20) if n_ep >= 36 else (1, max(1, (n_ep - 2) // 3), max(1, (n_ep - 2) // 3))
            cmd = ([sys.executable, os.path.join(ROOT, "tools", "engine_epoch.py"), ds_dir.path(), "--staged-host", "--fanout"]
                   + [str(f) for f in fanouts] + ["--batch-size", str(args.batch), "--cache-percentage", str(cache),
                                                  "--staged-warm", str(st_warm), "--staged-steps", str(st_serial), str(st_over)])
            r = child_json(cmd, timeout=max(10.0, min(300.0, left() - 10.0)), env=env, what="staged-host engine child")
            if "error" not in r and r.get("serial") and r.get("overlapped"):
                sr, ov = r["serial"], r["overlapped"]
                bound = min(x for x in (sr["cpu_gather_GBps"], sr["h2d_GBps"]) if x)
                ov["over_min_of_cpu_gather_and_h2d"] = ov["effective_GBps"] / bound
            staged[name] = r
        log("staged host tier done")

    # ---- N = 1: BASELINE configs[2], every row in pinned host DRAM (cache_ratio 0) -----------------------------
    cpu_feat = None
    if world == 1 and full and not args.no_host_tier and fits("host_tier", 4 + 40 * size_factor):
        cache = keep_main["cache"]
        t0 = time.perf_counter()
        hf = torch.empty((N, dim), dtype=torch.float32, pin_memory=True)  # hipHostMalloc, device-mapped
        t_pin = time.perf_counter() - t0
        hf.copy_(cache)  # the same rows the HBM tier holds (node order)
        torch.cuda.synchronize()
        # the box's pinned-copy rate, the ceiling of this tier: H2D of a 4-GiB slice of the table, 3 times
        probe_rows = min(N, (4 << 30) // row_bytes)
        dst = torch.empty((probe_rows, dim), dtype=torch.float32, device=dev)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dst.copy_(hf[:probe_rows], non_blocking=True)
        c0.record()
        for _ in range(3):
            dst.copy_(hf[:probe_rows], non_blocking=True)
        c1.record()
        torch.cuda.synchronize()
        pinned_GBps = 3 * probe_rows * row_bytes / (c0.elapsed_time(c1) / 1e3) / 1e9
        del dst

        def extract_host(nodes, num_max, o, num_dev, counters):  # DoGPUFeatureExtract, dist_loops.cc:585-634
            ops.gather_scatter(o, hf, nodes, None, num=num_max, num_dev=num_dev)
        hb, next_step = measure(extract_host, args.host_steps, 1, 1, first_step=next_step)
        h = hb[0]
        res["host_tier"] = {
            "config": "BASELINE configs[2]: same workload, cache_ratio 0 -- every feature row in pinned host DRAM "
                      "(hipHostMalloc, device-mapped), gathered zero-copy by the same kernel",
            "steps": args.host_steps, "ms_per_step": h["elapsed"] / args.host_steps * 1e3,
            "edges_per_s": h["edges_all"] / h["elapsed"],
            "feature_extract_GBps": h["feat_rate_all"], "rows_per_step": h["rows"] / args.host_steps,
            "pinned_h2d_copy_GBps": pinned_GBps, "frac_of_pinned_copy": h["feat_rate_all"] / pinned_GBps,
            "pinned_GiB": N * row_bytes / 2 ** 30, "pin_seconds": t_pin,
        }
        cpu_feat = hf.numpy()
    if staged is not None:
        if isinstance(res.get("host_tier"), dict):
            res["host_tier"]["staged"] = staged
        else:
            res["host_tier_staged"] = staged

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and fits("cpu_baseline", args.cpu_seconds * 1.3 + 3 + 25 * size_factor * (cpu_feat is None)):
            if cpu_feat is None:  # no host copy of the table yet (partial cache: host_feat; else generate it)
                if host_feat is not None:
                    cpu_feat = host_feat.numpy()
                else:
                    cpu_feat = np.empty((N, dim), np.float32)
                    feat_rows(torch.arange(N, dtype=torch.int64), torch.from_numpy(cpu_feat))
            res["cpu_baseline"] = cpu_baseline(graph, fanouts, args.batch, cpu_feat, args.cpu_seconds)
        res["budget"]["finished_at_s"] = round(time.perf_counter() - T0, 1)
        print(json.dumps(res), flush=True)
    ds_dir.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
