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
    a = ap.parse_args()
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


if __name__ == "__main__":
    main()
