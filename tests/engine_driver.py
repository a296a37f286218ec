"""Runs the samgraph_* engine like a reference example script would and dumps every batch to .npz.

    python tests/engine_driver.py <dataset_dir> <out_prefix> <arch0|arch1|arch6> [num_worker] [extra k=v ...]

arch6 follows example/samgraph/sgnn/train_graphsage.py:106-108,397-412: config + data_init in the parent, one
forked worker per GPU (os.fork before anything touches the GPU), each worker sample_init/train_init and
then loops sample_once / get_next_batch.  Used by tests/test_engine.py (subprocess) -- also a usage example.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_worker(sam, worker_id, num_layers, out_prefix, pipelined):
    import torch
    batches = {}
    n_local = sam.num_local_step()
    n_epoch = sam.num_epoch()
    if pipelined:
        sam.extract_start(0)
    for _ in range(n_epoch * n_local):
        if not pipelined:
            sam.sample_once()
        key = sam.get_next_batch()
        rec = {"feat": sam.get_graph_feat(key).cpu().numpy(), "label": sam.get_graph_label(key).cpu().numpy(),
               "input_nodes": sam.get_graph_input_nodes(key).cpu().numpy(),
               "output_nodes": sam.get_graph_output_nodes(key).cpu().numpy()}
        for i, (row, col, ns, nd) in enumerate(sam.get_graph_coo(key, num_layers)):
            rec[f"row{i}"], rec[f"col{i}"] = row.cpu().numpy(), col.cpu().numpy()
            rec[f"data{i}"] = sam.get_graph_data(key, i).cpu().numpy()
            rec[f"num_src{i}"], rec[f"num_dst{i}"] = ns, nd
            assert sam.get_graph_num_edge(key, i) == row.numel()
        rec["miss_bytes"] = sam.get_log_step_value_by_key(key, sam.kLogL1MissBytes)
        rec["num_sample"] = sam.get_log_step_value_by_key(key, sam.kLogL1NumSample)
        for k, v in rec.items():
            batches[f"{key}:{k}"] = v
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    np.savez(f"{out_prefix}.w{worker_id}.npz", **batches)
    sam.shutdown()


def main():
    dataset, out_prefix, arch = sys.argv[1:4]
    num_worker = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    extra = dict(a.split("=", 1) for a in sys.argv[5:])
    import samgraph.torch as sam
    fanout = [int(x) for x in extra.pop("fanout", "5 4").split()]
    pipelined = extra.pop("pipelined", "0") == "1"
    die_worker = int(extra.pop("die_worker", -1))  # this worker exits before it initialises (deadline tests)
    cfg = {"dataset_path": dataset, "_arch": sam.builtin_archs[arch]["arch"],
           "_sample_type": sam.sample_types[extra.pop("sample_type", "khop3")],
           "batch_size": int(extra.pop("batch_size", 64)), "num_epoch": int(extra.pop("num_epoch", 2)),
           "_cache_policy": sam.cache_policies[extra.pop("cache_policy", "degree")],
           "cache_percentage": float(extra.pop("cache_percentage", 0.0)), "max_sampling_jobs": 10,
           "max_copying_jobs": 2, "omp_thread_num": int(extra.pop("omp_thread_num", 4)), "num_layer": len(fanout), "num_hidden": 256, "lr": 0.003,
           "dropout": 0.5, "num_fanout": len(fanout), "fanout": fanout, "seed": int(extra.pop("seed", 1234))}
    if cfg["_sample_type"] == sam.kRandomWalk:  # operation.cc:164-175: no fanout keys, num_neighbor per layer
        cfg.pop("num_fanout"), cfg.pop("fanout")
        cfg.update(random_walk_length=3, random_walk_restart_prob=0.5, num_random_walk=4, num_neighbor=5)
    cfg.update(extra)
    if arch == "arch0":  # CPU sampler + extractor (cpu_engine.cc); trainer_ctx=cpu:0 keeps the batch on the host
        cfg["sampler_ctx"] = "cpu:0"
        cfg.setdefault("trainer_ctx", "cuda:0")
        sam.config(cfg)
        sam.init()
        run_worker(sam, 0, len(fanout), out_prefix, pipelined)
        return
    if arch == "arch1":
        cfg.update(sampler_ctx="cuda:0", trainer_ctx="cuda:0")
        sam.config(cfg)
        sam.init()
        run_worker(sam, 0, len(fanout), out_prefix, pipelined)
        return
    cfg["num_worker"] = num_worker
    sam.config(cfg)
    sam.data_init()  # host only: the GPU is first touched in the children
    pids = []
    for w in range(num_worker):
        pid = os.fork()
        if pid == 0:
            try:
                if w == die_worker:
                    os._exit(0)
                ctx = f"cuda:{w}"
                sam.sample_init(w, ctx)
                sam.train_init(w, ctx)
                run_worker(sam, w, len(fanout), out_prefix, pipelined)
                os._exit(0)
            except BaseException as e:  # noqa: BLE001
                print("worker failed:", repr(e), file=sys.stderr)
                os._exit(1)
        pids.append(pid)
    bad = 0
    for _ in pids:
        bad += sam.wait_one_child()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
