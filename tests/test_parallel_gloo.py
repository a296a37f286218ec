"""N > 1 path on CPU: two gloo ranks partition the seeds and reduce the bench counters."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from xgnn_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, train, batch, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = []
    for ep in range(2):
        mine = parallel.rank_slice(train, world, rank, ep)
        gathered = [torch.zeros(len(mine), dtype=torch.int64) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(mine.astype(np.int64)))
        out.append(torch.cat(gathered).numpy())
    stats = torch.tensor([1.0 + rank, 10.0 * (rank + 1), 3.0], dtype=torch.float64)
    mx, sm = parallel.reduce_stats(stats, dist)
    dist.barrier()
    if rank == 0:
        q.put((out, mx.tolist(), sm.tolist()))
    dist.destroy_process_group()


def test_two_rank_seed_partition_and_reduction():
    world, batch = 2, 16
    train = np.random.RandomState(0).permutation(1000)[:101].astype(np.uint32)  # odd: needs padding
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, train, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, mx, sm = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    padded = parallel.pad_train_set(train, world)
    assert len(padded) == 102 and padded[-1] == train[0]
    for ep in range(2):
        # union of the ranks' slices == padded train set (test_um_multi_sample.cc:57-104 semantics)
        assert sorted(out[ep].tolist()) == sorted(padded.astype(np.int64).tolist())
    assert not np.array_equal(out[0], out[1])  # reshuffled per epoch
    assert mx == [2.0, 20.0, 3.0] and sm == [3.0, 30.0, 6.0]
    assert parallel.steps_per_epoch(101, 2, 16) == 4
