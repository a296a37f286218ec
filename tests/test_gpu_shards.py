"""GGMS feature shards across two processes sharing the box's one GPU: peer (hipIpc, in-kernel loads) and
exchange (all-to-all; gloo through host memory here, RCCL on a real node) must both reproduce extract()."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, q):
    import oracle
    from xgnn_amd import ggms_store
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    N, dim, num_cached = 20_000, 100, 13_001
    rng = np.random.RandomState(5)
    feat = rng.standard_normal((N, dim)).astype(np.float32)
    rank_list = rng.permutation(N).astype(np.int64)
    table = np.full(N, -1, np.int32)
    table[rank_list[:num_cached]] = np.arange(num_cached, dtype=np.int32)
    t_table = torch.from_numpy(table).to(dev)
    host_feat = torch.from_numpy(feat).pin_memory()
    t_rank = torch.from_numpy(rank_list)

    def rows_of(node_ids, out):
        out.copy_(torch.from_numpy(feat[node_ids.numpy()]))

    # hybrid: the R hottest slots replicated on every GPU, the tail sharded; ident: everything cached, slot = node id
    R = 4_000 if mode == "hybrid" else 0
    replica = None
    if mode == "ident":
        num_cached, rank_list, t_table = N, np.arange(N, dtype=np.int64), None
        t_rank = torch.from_numpy(rank_list)
    if R:
        replica = torch.from_numpy(feat[rank_list[:R]]).to(dev)
    peer = mode != "a2a"
    shard, holder = ggms_store.shard_rows(rows_of, t_rank[R:], num_cached - R, world, rank, dim, torch.float32, dev,
                                          shared=peer)
    store = ggms_store.FeatureShards(shard, t_table, world, rank, mode="peer" if peer else "a2a", dist=dist,
                                     host_feat=host_feat, replica=replica)
    if peer:
        store.connect_peers(holder)
    ok = True
    for b in range(3):
        n = 5000 + 333 * rank + b
        nodes = np.random.RandomState(100 * b + rank).randint(0, N, n).astype(np.uint32)
        t_nodes = torch.from_numpy(nodes.view(np.int32)).to(dev)
        out = torch.zeros((n + 3, dim), dtype=torch.float32, device=dev)
        counters = torch.zeros(4, dtype=torch.int64, device=dev)
        if mode in ("hybrid", "ident"):
            store.extract(t_nodes, n, out, counters=counters)
        else:
            store.extract(t_nodes, n, out)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        ok = ok and got[:n].tobytes() == oracle.extract(feat, nodes).tobytes() and not got[n:].any()
        if mode in ("hybrid", "ident"):  # rows by tier: host, remote shard, local shard, replica
            slots = nodes.astype(np.int64) if t_table is None else table[nodes].astype(np.int64)
            sh = slots >= R
            want = [int((slots < 0).sum()), int((sh & ((slots - R) % world != rank)).sum()),
                    int((sh & ((slots - R) % world == rank)).sum()), int(((slots >= 0) & (slots < R)).sum())]
            ok = ok and counters.cpu().tolist() == want
    dist.barrier()  # nobody unmaps a shard a peer may still be reading
    q.put((rank, ok))
    dist.barrier()
    if holder is not None:
        holder.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["peer", "a2a", "hybrid", "ident"])
def test_two_processes_one_gpu(mode):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def _big_shard_worker(rank, world, port, q):
    from xgnn_amd import ops
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    rows = 3000 * (1 << 20) // 512  # 3000 MiB: bit 31 of the byte size is set
    sh = ops.SharedShard((rows, 128), torch.float32, torch.device("cuda", 0))
    sh.tensor[:4] = float(rank + 1)
    sh.tensor[-4:] = float(10 * (rank + 1))
    torch.cuda.synchronize()
    handles = [None] * world
    dist.all_gather_object(handles, sh.export_handle())
    p = sh.import_peer(handles[1 - rank])
    peer = torch.as_tensor(ops._RawDevice(p, (rows, 128), "<f4"), device="cuda:0")
    ok = float(peer[0, 0].item()) == float(2 - rank) and float(peer[-1, -1].item()) == float(10 * (2 - rank))
    dist.barrier()
    q.put((rank, ok))
    dist.barrier()
    sh.close()
    dist.destroy_process_group()


def test_shard_size_with_bit31_set_can_be_opened():
    """ROCm 7.2: hipIpcOpenMemHandle never returns for an allocation whose byte size has bit 31 set (3000 MiB, 7.1 GB,
    28.4 GB -- the papers100M shards at 8 and 2 GPUs); ggms_device_alloc sizes shards around that (include/ggms.h
    ggms_ipc_safe_bytes).  Would hang (and time out) without it."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_big_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=120) for _ in range(world))
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert res == [(0, True), (1, True)]


def _topology_worker(rank, world, port, fraction, q):
    """XGNN mode's graph across processes: rank r keeps topology shard r, maps the peers' (hipIpc) and samples its own
    batches through the view; every batch must be the oracle's batch on the plain CSR."""
    import oracle
    from graphgen import powerlaw_csr
    from xgnn_amd import ggms_store, ops
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ip, ix = powerlaw_csr(40_000, mean_deg=18, seed=8)
    N = ip.size - 1
    t_ip = torch.from_numpy(ip.view(np.int32)).to(dev)
    t_ix = torch.from_numpy(ix.view(np.int32)).to(dev)
    ncn = ggms_store.num_cache_node_for(ip, fraction)
    host = None
    if ncn < N:  # the engine's layout of the last slot: indptr in HBM, neighbour lists in registered host memory
        host = ops.RegisteredHost(ix, dev)
        slot = (t_ip, host.tensor)
    else:
        slot = (t_ip, torch.zeros(4, dtype=torch.int32, device=dev))
    topo = ggms_store.TopologyShards(t_ip, t_ix, world, rank, ncn, dist, slot)
    del t_ix  # only the shard (and the peers' mappings) serve the cached nodes from here on
    fanouts = [10, 5]
    ok = True
    for stype, code, ocode in (("khop3", ops.KHOP3, oracle.KHOP3), ("khop0", ops.KHOP0, oracle.KHOP0)):
        bs = ops.BatchSampler(topo.graph, fanouts, 1000, sample_type=code, seed=40 + rank)
        st = oracle.random_states(bs.states.shape[0], 40 + rank) if stype != "khop0" else None
        rng = np.random.RandomState(7 + rank)
        for rep in range(2):
            seeds = rng.permutation(N)[:1000].astype(np.uint32)
            bs.sample(torch.from_numpy(seeds.view(np.int32)).to(dev), distinct=True)
            got = bs.result()
            want = oracle.do_sample(ocode, ip, ix, seeds, fanouts, st)
            ok = ok and np.array_equal(got["input_nodes"].cpu().numpy().view(np.uint32), want["input_nodes"])
            for i in range(2):
                ok = ok and np.array_equal(got["layers"][i]["row"].cpu().numpy().view(np.uint32), want["layers"][i]["row"])
                ok = ok and np.array_equal(got["layers"][i]["col"].cpu().numpy().view(np.uint32), want["layers"][i]["col"])
        del bs
    torch.cuda.synchronize()
    dist.barrier()  # nobody unmaps a shard a peer may still be reading
    q.put((rank, ok))
    dist.barrier()
    topo.close()
    if host is not None:
        host.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,fraction", [(2, 1.0), (2, 0.5), (4, 0.7)])
def test_topology_shards_across_processes(world, fraction):
    """DistGraph across processes on the box's one GPU (cuda/dist_graph.cu:228-385): own shard + hipIpc-mapped peers + the
    host slot behind one DeviceGraph; khop3 and khop0 batches equal the oracle's on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_topology_worker, args=(r, world, port, fraction, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=300) for _ in range(world))
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert res == [(r, True) for r in range(world)]
    assert all(p.exitcode == 0 for p in procs)
