"""N > 1: data parallelism over seed mini-batches (the only axis the path shards on).

One process per GPU.  The train set is padded to a multiple of the world size, every rank applies the
same per-epoch permutation and takes a contiguous slice (DistAlignedShuffler,
dist/dist_shuffler_aligned.cc:37-146).  No collective touches the data path; torch.distributed is used
only for the timing barrier and for reducing the counters that bench.py reports.
"""
import numpy as np


def pad_train_set(train, world):
    """dist_shuffler_aligned.cc:46-56: pad with the first entries up to a multiple of `world`."""
    pad = (-len(train)) % world
    return np.concatenate([train, train[:pad]]) if pad else train.copy()


def rank_slice(train, world, rank, epoch, permute=None):
    """The rank's seeds for `epoch`: same permutation on every rank, contiguous slice per rank.
    `permute(data, epoch)` defaults to numpy's seeded permutation (bench); the engine uses minstd Fisher-Yates."""
    data = pad_train_set(train, world)
    if permute is None:
        data = data[np.random.RandomState(epoch).permutation(len(data))]
    else:
        data = permute(data, epoch)
    per_rank = len(data) // world
    return data[rank * per_rank:(rank + 1) * per_rank].copy()


def steps_per_epoch(num_train, world, batch):
    per_rank = (num_train + world - 1) // world
    return (per_rank + batch - 1) // batch


def reduce_stats(stats, dist=None):
    """stats: 1-D float64 tensor on the rank's device.  Returns (max over ranks, sum over ranks)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return stats.clone(), stats.clone()
    mx, sm = stats.clone(), stats.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    return mx, sm
