"""N > 1 path on CPU: two gloo ranks partition the seeds and reduce the bench counters."""
import os
import socket

import numpy as np
import pytest
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


# ---- GGMS feature shards, exchange form: host logic on two CPU ranks --------------------------------------
class _NumpyLeaf:
    """Test stand-in for the HIP leaf operators (checker side: plain numpy on CPU tensors)."""

    def split_by_owner(self, table, nodes, num, num_part, order):
        """buckets in the sequence `order` (a permutation of 0..num_part), stable inside a bucket"""
        slots = table[nodes[:num].long()]
        owner = torch.where(slots < 0, torch.full_like(slots, num_part), slots % num_part)
        place = torch.empty(num_part + 1, dtype=torch.int64)
        place[order] = torch.arange(num_part + 1)
        by = torch.argsort(place[owner.long()], stable=True)
        counts = torch.bincount(owner, minlength=num_part + 1).to(torch.int64)
        row = torch.where(slots < 0, nodes[:num], slots // num_part)[by].to(torch.int32)
        return row, by.to(torch.int32), counts

    def gather(self, src, index):
        return src[index.long()]

    def gather_scatter(self, out, src, src_index, dst_index):
        rows = src if src_index is None else src[src_index.long()]
        if dst_index is None:
            out[:rows.shape[0]] = rows
        else:
            out[dst_index.long()] = rows


    def gather_tiered(self, out, nodes, num, table, replica, parts, num_part, my_part, host_feat, num_dev=None,
                      counters=None):
        """The tier map of ggms_extract_tiered (include/ggms.h) in plain torch; parts = list of every rank's shard."""
        R = 0 if replica is None else replica.shape[0]
        for i in range(num):
            node = int(nodes[i])
            slot = node if table is None else int(table[node])
            if slot < 0:
                out[i], tier = host_feat[node], 0
            elif slot < R:
                out[i], tier = replica[slot], 3
            else:
                s = slot - R
                out[i], tier = parts[s % num_part][s // num_part], (2 if s % num_part == my_part else 1)
            if counters is not None:
                counters[tier] += 1


def _hybrid_worker(rank, world, port, q):
    """'hybrid' store on two CPU ranks: the R hottest slots replicated on every rank, the tail sharded modulo P."""
    from xgnn_amd.ggms_store import FeatureShards
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, dim, num_cached, R = 400, 5, 300, 77
    g = torch.Generator().manual_seed(4)
    feat = torch.randn(N, dim, generator=g)
    rank_list = torch.randperm(N, generator=g)
    table = torch.full((N,), -1, dtype=torch.int32)
    table[rank_list[:num_cached]] = torch.arange(num_cached, dtype=torch.int32)
    replica = feat[rank_list[:R]].contiguous()
    shard = feat[rank_list[R + rank:num_cached:world]].contiguous()  # slot s >= R -> rank (s - R) % P, row (s - R) // P
    shards = [None] * world
    dist.all_gather_object(shards, shard)  # stands in for the hipIpc mapping of the peers' shards
    store = FeatureShards(shard, table, world, rank, mode="peer", dist=dist, leaf=_NumpyLeaf(), host_feat=feat,
                          replica=replica)
    store.parts_table = shards
    ok = True
    for b in range(3):
        nodes = torch.randint(0, N, (120 + 11 * rank + b,), generator=torch.Generator().manual_seed(7 * b + rank),
                              dtype=torch.int32)
        out = torch.zeros(nodes.numel() + 2, dim)
        counters = torch.zeros(4, dtype=torch.int64)
        store.extract(nodes, nodes.numel(), out, counters=counters)
        slots = table[nodes.long()]
        want = [int((slots < 0).sum()), int(((slots >= R) & ((slots - R) % world != rank)).sum()),
                int(((slots >= R) & ((slots - R) % world == rank)).sum()), int(((slots >= 0) & (slots < R)).sum())]
        ok = ok and torch.equal(out[:nodes.numel()], feat[nodes.long()]) and counters.tolist() == want
    # a peer store without a host tier must refuse a table that has uncached nodes (it would fault on the GPU)
    try:
        FeatureShards(shard, table, world, rank, mode="peer", dist=dist, leaf=_NumpyLeaf(), host_feat=None)
        ok = False
    except ValueError:
        pass
    dist.barrier()
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_hybrid_store_tier_map(world):
    """replica + 2-way and 8-way shards + host rows behind one gather: the tier map and its counters"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hybrid_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]


def _shard_worker(rank, world, port, q):
    from xgnn_amd.ggms_store import FeatureShards
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, dim, num_cached = 500, 7, 320
    g = torch.Generator().manual_seed(3)
    feat = torch.randn(N, dim, generator=g)
    rank_list = torch.randperm(N, generator=g)
    table = torch.full((N,), -1, dtype=torch.int32)
    table[rank_list[:num_cached]] = torch.arange(num_cached, dtype=torch.int32)
    shard = feat[rank_list[rank:num_cached:world]].contiguous()  # slot s -> rank s % P, row s // P
    store = FeatureShards(shard, table, world, rank, mode="a2a", dist=dist, leaf=_NumpyLeaf(), host_feat=feat)
    ok = True
    for b in range(3):
        nodes = torch.randint(0, N, (90 + 17 * rank + b,), generator=torch.Generator().manual_seed(10 * b + rank),
                              dtype=torch.int32)
        out = torch.zeros(nodes.numel() + 5, dim)
        store.extract(nodes, nodes.numel(), out)
        ok = ok and torch.equal(out[:nodes.numel()], feat[nodes.long()]) and bool((out[nodes.numel():] == 0).all())
    # a batch whose rows all live on ONE owner (empty buckets elsewhere), and an empty batch
    only0 = rank_list[0:num_cached:world][:min(40, (num_cached + world - 1) // world)].to(torch.int32)
    out = torch.zeros(only0.numel(), dim)
    store.extract(only0, only0.numel(), out)
    ok = ok and torch.equal(out, feat[only0.long()])
    store.extract(only0[:0], 0, torch.zeros(1, dim))
    dist.barrier()
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_feature_shards_all_to_all(world):
    """The exchange form on 2 and on EIGHT CPU ranks (the node size north_star names): 8-way owner buckets laid out
    [ranks ascending without me | me | host], split sizes with a zero for the own bucket, ids out / rows back."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]


def test_plan_replication_fills_the_budget():
    from xgnn_amd.ggms_store import plan_replication
    row = 512
    n = 111_059_956
    assert plan_replication(n, row, 1, 10 ** 9) == n                    # one GPU: nothing to shard
    assert plan_replication(n, row, 8, n * row) == n                    # everything fits: replicate all
    assert plan_replication(n, row, 8, n * row // 8) == 0               # just the shard fits: pure modulo shards
    for world, budget in ((8, 24 * 10 ** 9), (4, 30 * 10 ** 9), (2, 40 * 10 ** 9)):
        r = plan_replication(n, row, world, budget)
        used = r * row + -(-(n - r) // world) * row
        assert 0 < r < n and used <= budget < used + world * row * 2    # the largest prefix that fits


def test_topology_shards_blocks_are_bounded_by_edges():
    """_DatasetPartition (dist_graph.cu:228-272) restated on tensors: every shard count, cached prefix and block size
    (one block, blocks smaller than a hub's list, blocks of a few lists) gives the shards a plain loop gives."""
    import torch
    from xgnn_amd import ggms_store
    rng = np.random.default_rng(0)
    n = 3000
    deg = rng.integers(0, 40, n)
    deg[7] = 2500  # a hub longer than the small block sizes
    ip = np.zeros(n + 1, np.int64)
    ip[1:] = np.cumsum(deg)
    ix = rng.integers(0, n, ip[-1]).astype(np.int32)
    tip, tix = torch.from_numpy(ip.astype(np.int32)), torch.from_numpy(ix)
    for P in (1, 2, 3, 8):
        for ncn in (n, n // 2, 17, 2, 0):
            for step in (1 << 26, 100, 4000):
                pip, pix = ggms_store.topology_shards(tip, tix, P, ncn, edges_per_step=step)
                for p in range(P):
                    nodes = np.arange(p, ncn, P)
                    want_ip = np.zeros(len(nodes) + 1, np.int64)
                    want_ip[1:] = np.cumsum(deg[nodes])
                    want_ix = (np.concatenate([ix[ip[v]:ip[v + 1]] for v in nodes]) if want_ip[-1]
                               else np.zeros(0, np.int32))
                    assert np.array_equal(pip[p].numpy(), want_ip.astype(np.int32)), (P, ncn, step, p)
                    assert np.array_equal(pix[p].numpy()[:want_ip[-1]], want_ix), (P, ncn, step, p)


def test_plan_with_links_keeps_xgmi_off_the_critical_path():
    """The placement after the link probe (PartitionSolver's role): the sharded tail is sized so that a batch's remote
    rows arrive while its local rows stream; never less replication than the budget plan, never more than fits."""
    from xgnn_amd.ggms_store import plan_replication, plan_with_links
    n, rb, P = 111_059_956, 512, 8
    budget = plan_replication(n, rb, P, int(48e9))
    r, rec = plan_with_links(n, rb, P, budget, inbound_GBps=400.0, local_GBps=2500.0, capacity_bytes=int(200e9))
    f = 0.8 * 400 / 2900
    assert abs(rec["remote_row_share_the_links_hide"] - f) < 1e-12 and r == rec["replicated_rows_chosen"] >= budget
    tail = (n - r) / n
    assert tail * (P - 1) / P <= f + 1e-7 and tail * (P - 1) / P > f - 1e-3  # the bound binds: links slower than the budget plan assumes
    # fast links: the budget plan already hides the remote rows -- nothing changes
    r2, rec2 = plan_with_links(n, rb, P, budget, inbound_GBps=2000.0, local_GBps=2500.0, capacity_bytes=int(200e9))
    assert r2 == budget and rec2["replicated_rows_link_plan"] < budget
    # capacity binds: no more replica than the GPU holds beside its share of the tail
    r3, rec3 = plan_with_links(n, rb, P, budget, inbound_GBps=50.0, local_GBps=2500.0, capacity_bytes=int(50e9))
    assert r3 == rec3["replicated_rows_capacity"] == plan_replication(n, rb, P, int(50e9)) < rec3["replicated_rows_link_plan"]
    # no probe (or one GPU): the budget plan
    assert plan_with_links(n, rb, P, budget, None, None, int(200e9)) == (budget, None)
    assert plan_with_links(n, rb, 1, n, 400.0, 2500.0, int(200e9)) == (n, None)
