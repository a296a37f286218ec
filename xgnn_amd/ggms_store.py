"""GGMS feature shards across the GPUs of one node, one process per GPU.

Cache slot s (position in the cache rank list, s < num_cached) lives on rank s % P at row s // P -- the
reference's partition cache (GPUCacheManager ctor, cuda/cuda_cache_manager_host.cc:133-254); nodes that are
not cached stay in host memory.  Two interchangeable ways to bring a batch's rows together:

  mode "peer"  every rank maps its peers' shards (hipIpc) and ONE gather kernel dereferences the owner's HBM
               directly over xGMI -- what the reference does over NVLink (combine_cache_data_for_partition,
               cuda_cache_manager_device.cu:277-299).  No exchange step, no staging.
  mode "a2a"   one exchange: bucket the batch by owner -> all-to-all of row ids -> every owner gathers the
               rows asked of it from its own HBM -> all-to-all of rows -> scatter into the batch.  Fewer,
               larger xGMI transfers through RCCL; ONE host wait per batch (the split sizes are host arguments of
               the collective), no concatenations.  The measured alternative to "peer", not the default.

Both produce the bytes `extract(full_table, nodes)` would.  The device work goes through `leaf` (default:
the HIP operators of xgnn_amd.ops); tests exercise the host logic on CPU ranks by passing their own leaf.
"""
import os
import sys
import threading

import numpy as np
import torch


def ipc_timeout_s():
    """Deadline of every wait on a peer while the shards are connected (GGMS_IPC_TIMEOUT_S, default 120 s)."""
    try:
        v = float(os.environ.get("GGMS_IPC_TIMEOUT_S", "0"))
    except ValueError:
        v = 0.0
    return v if v > 0 else 120.0


def with_deadline(fn, what, seconds=None, on_timeout=None):
    """Run fn() on a helper thread and wait for it at most `seconds`.  A wait on a peer that never ends --
    a rank that died before publishing its shard, or a hipIpcOpenMemHandle that does not return (seen on ROCm 7.2
    for exporter sizes with bit 31 set, include/ggms.h ggms_ipc_safe_bytes) -- must end the run with a message,
    not hold it until somebody's time limit.  The stuck call cannot be cancelled and there is nothing to retry
    in-process, so on expiry the process prints `what` and EXITS with code 3 (on_timeout, a test hook, replaces
    the exit).  Exceptions of fn are re-raised here."""
    seconds = ipc_timeout_s() if seconds is None else seconds
    box = {}
    # the current device is per thread: the helper must work on the caller's (RCCL stages objects on it)
    dev = torch.cuda.current_device() if torch.cuda.is_available() and torch.cuda.is_initialized() else None

    def run():
        try:
            if dev is not None:
                torch.cuda.set_device(dev)
            box["value"] = fn()
        except BaseException as e:  # noqa: BLE001 -- handed to the caller
            box["error"] = e

    t = threading.Thread(target=run, name="ggms-deadline", daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        msg = f"[ggms] gave up after {seconds:.0f} s: {what} (GGMS_IPC_TIMEOUT_S); exiting"
        if on_timeout is not None:
            return on_timeout(msg)
        print(msg, file=sys.stderr, flush=True)
        os._exit(3)
    if "error" in box:
        raise box["error"]
    return box.get("value")


class PeerConnectError(RuntimeError):
    """A peer's shard could not be mapped (hipIpcOpenMemHandle / hipIpcGetMemHandle returned an error).  Raised on
    EVERY rank of the group with the failures of all of them, so the caller may decide together what to do."""


class FailedShard:
    """Stands in for a shard this rank could not build: connect_shared(FailedShard(reason), ...) takes part in the
    handle exchange and the verdict and raises PeerConnectError on every rank, this one's reason included."""

    def __init__(self, reason):
        self.reason = str(reason)[:300]

    def release_peers(self):
        pass


def connect_shared(shared_shard, world, rank, dist, group=None, what="shard"):
    """Publish this rank's shard (an ops.SharedShard) and map every peer's: DistGraph::_DataIpcShare
    (cuda/dist_graph.cu:274-307) over torch.distributed's byte channel.  Returns the shards' device addresses in this
    process, by rank.  Every wait on a peer has a deadline; an IPC call that FAILS on any rank raises PeerConnectError
    on EVERY rank (after closing the mappings that did open), so nobody is left waiting in the next collective."""
    if isinstance(shared_shard, FailedShard):
        # this rank could not even build its shard (out of memory, ...): it still walks through the same collectives --
        # publishing "no handle" and its reason -- so that every rank gets the same verdict instead of waiting for it
        me = f"rank {rank} of {world}"
        nbytes, mine, failed = 0, None, f"{me}: could not build its {what}: {shared_shard.reason}"
    else:
        # the shard must be COMPLETE before a peer may read it: drain this device's queue before publishing (the
        # fill kernels are asynchronous), and meet the peers again once everybody has mapped everybody (below)
        if shared_shard.tensor.is_cuda:
            torch.cuda.synchronize(shared_shard.tensor.device)
        nbytes = int(np.prod(shared_shard.shape)) * shared_shard.tensor.element_size()
        me = f"rank {rank} of {world} (device {shared_shard.tensor.device})"
        failed = None
        try:
            mine = shared_shard.export_handle()
        except Exception as e:  # noqa: BLE001 -- published as "no handle"; reported with the verdicts below
            mine, failed = None, f"{me}: hipIpcGetMemHandle of its own {what} ({nbytes} bytes): {type(e).__name__}: {e}"
    if world == 1:
        handles = [(mine, nbytes)]
    else:
        handles = [None] * world
        # every wait on a peer is bounded (with_deadline): exit with a message instead of hanging
        with_deadline(lambda: dist.all_gather_object(handles, (mine, nbytes), group=group),
                      f"{me} waiting for the peers' {what} handles (all_gather): a rank never published its shard")
    ptrs = []
    try:
        for r in range(world):
            if isinstance(shared_shard, FailedShard):
                break
            if r == rank:
                ptrs.append(shared_shard.ptr)
                continue
            h, peer_bytes = handles[r]
            if h is None:  # that rank could not export: it says so itself
                ptrs.append(0)
                continue
            ptrs.append(with_deadline(lambda h=h: shared_shard.import_peer(h),
                                      f"{me}: hipIpcOpenMemHandle of rank {r}'s {what} ({peer_bytes} bytes) did not return"))
    except Exception as e:  # noqa: BLE001 -- an open that FAILED (one that hangs ends the process above)
        failed = failed or f"{me}: {what} of rank {r} ({peer_bytes} bytes): {type(e).__name__}: {e}"
    if world > 1:
        # the meeting point after the mapping doubles as the verdict: every rank learns of every failure and all
        # of them leave together (a rank raising alone would strand the others in the next collective)
        verdicts = [None] * world
        with_deadline(lambda: dist.all_gather_object(verdicts, failed, group=group),
                      f"{me} waiting for the peers after mapping their {what}s: a rank is stuck opening one")
        failed = "; ".join(v for v in verdicts if v) or None
    if failed:
        # nothing will own the mappings that did open: close them here (every rank gets here, see above)
        shared_shard.release_peers()
        raise PeerConnectError(failed)
    return ptrs


class HipLeaf:
    """The device operators the store needs, on HIP (xgnn_amd.ops)."""

    def __init__(self):
        from . import ops
        self.ops = ops

    def split_by_owner(self, table, nodes, num, num_part, order, num_dev=None):
        """Group the batch by owning shard.  -> (bucket_row, bucket_pos, counts) with counts[p] = rows owned by
        shard p (p = num_part: host tier) as an int64 DEVICE tensor; the buckets lie in `bucket_row` / `bucket_pos`
        in the sequence `order` (a device int64 permutation of 0..num_part).  No host round trip: the bucket
        cursors are the exclusive prefix of the histogram taken in that sequence, computed on the device.
        num is an upper bound when the batch size lives on the device (num_dev)."""
        ops, dev = self.ops, nodes.device
        counts = torch.zeros(num_part + 1, dtype=torch.int64, device=dev)
        slots = torch.empty(max(1, num), dtype=torch.int32, device=dev)
        ops.owner_histogram(table, nodes, num_part, slots, counts, num=num, num_dev=num_dev)
        in_order = counts[order]
        cursor = torch.empty_like(counts)
        cursor[order] = torch.cumsum(in_order, 0) - in_order
        row = torch.empty(max(1, num), dtype=torch.int32, device=dev)
        pos = torch.empty(max(1, num), dtype=torch.int32, device=dev)
        ops.owner_bucket(slots, nodes, num_part, cursor, row, pos, num=num, num_dev=num_dev)  # advances `cursor`
        return row, pos, counts

    def gather(self, src, index):
        return self.ops.extract(src, index)

    def gather_scatter(self, out, src, src_index, dst_index):
        n = (src_index if src_index is not None else dst_index).numel()
        if n:
            self.ops.gather_scatter(out, src, src_index, dst_index, num=n)

    def pointer_table(self, ptrs):
        return self.ops.PartTable(ptrs)

    def gather_peer(self, out, nodes, num, table, parts_table, num_part, host_feat, num_dev=None, num_miss=None):
        self.ops.extract_cached(out, nodes, table, parts_table, num_part, host_feat, num=num, num_dev=num_dev,
                                num_miss=num_miss)

    def gather_tiered(self, out, nodes, num, table, replica, parts_table, num_part, my_part, host_feat, num_dev=None,
                      counters=None):
        self.ops.extract_tiered(out, nodes, table, replica, parts_table, num_part, my_part, host_feat, num=num,
                                num_dev=num_dev, tier_rows=counters)


def _all_to_all(dist, out, inp, out_splits, in_splits, group=None):
    """all_to_all_single on the tensors' own device with RCCL; through host memory with any other backend
    (gloo on a one-GPU test box)."""
    if dist.get_backend(group) == "nccl" or not inp.is_cuda:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        return
    o, i = out.cpu(), inp.cpu()
    dist.all_to_all_single(o, i, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
    out.copy_(o)


class FeatureShards:
    """This rank's view of the partitioned feature cache."""

    def __init__(self, shard, table, world, rank, mode="peer", dist=None, leaf=None, host_feat=None, group=None,
                 replica=None):
        """shard: [ceil((num_sharded - rank) / world), dim] rows of this rank (a SharedShard's tensor in peer mode);
        table: int32[N] node -> slot or -1 (None: slot = node id, everything cached); host_feat: full table in
        (device-mapped) host memory for misses; replica (peer mode, "hybrid" store): this GPU's copy of the R hottest
        slots -- slot s < R is read from it, slot s >= R from shard (s - R) % world at row (s - R) // world."""
        assert mode in ("peer", "a2a")
        assert replica is None or mode == "peer", "hot-row replication rides on the peer gather"
        self.shard, self.table, self.world, self.rank, self.mode = shard, table, world, rank, mode
        self.dist, self.group, self.host_feat, self.replica = dist, group, host_feat, replica
        if mode == "peer" and host_feat is None and table is not None:
            # an uncached node would be read from host row `node` of a NULL table: a GPU memory fault.  Refuse it here.
            if bool((table == -1).any()):
                raise ValueError("FeatureShards(peer): the cache table has uncached nodes but no host tier was given")
        self.leaf = leaf if leaf is not None else HipLeaf()
        self.parts_table = None
        self._shared = None
        self._order = None  # a2a: bucket sequence [ranks ascending without me | me | host], on the device

    # ---- mode "peer": publish / map the shards ------------------------------------------------------
    def connect_peers(self, shared_shard):
        """Exchange hipIpc handles (all_gather of 64-byte blobs) and build the pointer table (a host array: the gather
        takes the shard pointers by value)."""
        assert self.mode == "peer"
        self._shared = shared_shard
        ptrs = connect_shared(shared_shard, self.world, self.rank, self.dist, self.group, what="feature shard")
        self.parts_table = self.leaf.pointer_table(ptrs)
        return self

    # ---- one batch ----------------------------------------------------------------------------------
    def extract(self, nodes, num, out, num_dev=None, num_miss=None, counters=None):
        """out[i, :] = feature row of nodes[i], i < num; with num_dev (device int64[1]) num is an upper bound.
        counters (peer mode): int64[4] on the device, rows served by {host, remote shard, local shard, replica}."""
        if self.mode == "peer":
            if counters is not None or self.replica is not None:
                # the tiered gather counts rows per tier (added to `counters`); num_miss = this call's host rows,
                # taken from a scratch counter set
                tiers = counters if num_miss is None else torch.zeros(4, dtype=torch.int64, device=out.device)
                self.leaf.gather_tiered(out, nodes, num, self.table, self.replica, self.parts_table, self.world,
                                        self.rank, self.host_feat, num_dev=num_dev, counters=tiers)
                if num_miss is not None:
                    num_miss.copy_(tiers[0:1])
                    if counters is not None:
                        counters.add_(tiers)
            else:
                self.leaf.gather_peer(out, nodes, num, self.table, self.parts_table, self.world, self.host_feat,
                                      **({"num_dev": num_dev, "num_miss": num_miss} if num_dev is not None else {}))
            return out
        return self._extract_a2a(nodes, num, out, num_dev)

    def _extract_a2a(self, nodes, num, out, num_dev=None):
        """One exchange, ONE host wait (the split sizes are host arguments of the collective):
        histogram -> bucket cursors on the device -> buckets laid out as [ranks ascending without me | me | host] so
        that the remote part is one contiguous block in rank order (no concatenation) -> request counts exchanged
        on the device (RCCL) -> the only device-to-host copy of the batch brings both count vectors -> ids out,
        rows gathered at their owners, rows back, scattered into the batch."""
        P, me, leaf, dist = self.world, self.rank, self.leaf, self.dist
        dev = nodes.device
        if self._order is None or self._order.device != dev:
            self._order = torch.tensor([p for p in range(P) if p != me] + [me, P], dtype=torch.int64, device=dev)
        row, pos, counts = leaf.split_by_owner(self.table, nodes, num, P, self._order,
                                               **({"num_dev": num_dev} if num_dev is not None else {}))
        send_dev = counts[:P].clone()
        send_dev[me] = 0  # my own bucket never leaves the GPU
        if P > 1 and dist.get_backend(self.group) == "nccl":
            recv_dev = torch.empty_like(send_dev)
            dist.all_to_all_single(recv_dev, send_dev, group=self.group)
            both = torch.stack([counts[:P], recv_dev, counts[P:P + 1].expand(P)]).cpu()  # the batch's one host wait
            counts_h, recv, n_host = both[0].tolist(), both[1].tolist(), int(both[2][0])
        else:  # other backends (gloo: tests, one-GPU boxes) exchange the counts through host memory
            c = counts.cpu()  # the batch's one host wait
            counts_h, n_host = c[:P].tolist(), int(c[P])
            recv = [0] * P
            if P > 1:
                s_h = c[:P].clone()
                s_h[me] = 0
                r_h = torch.zeros_like(s_h)
                dist.all_to_all_single(r_h, s_h, group=self.group)
                recv = r_h.tolist()
        send = [0 if p == me else int(counts_h[p]) for p in range(P)]
        recv = [int(x) for x in recv]
        n_remote, n_mine = sum(send), int(counts_h[me])
        # my own bucket and the host bucket: local gathers
        leaf.gather_scatter(out, self.shard, row[n_remote:n_remote + n_mine], pos[n_remote:n_remote + n_mine])
        if n_host:
            assert self.host_feat is not None, "batch has uncached nodes but no host tier was given"
            lo = n_remote + n_mine
            leaf.gather_scatter(out, self.host_feat, row[lo:lo + n_host], pos[lo:lo + n_host])
        if P == 1:
            return out
        ids_in = torch.empty(sum(recv), dtype=row.dtype, device=row.device)
        _all_to_all(dist, ids_in, row[:n_remote], recv, send, self.group)  # remote buckets: contiguous, rank order
        rows_out = leaf.gather(self.shard, ids_in) if sum(recv) else self.shard[:0]
        rows_in = torch.empty((n_remote,) + tuple(self.shard.shape[1:]), dtype=self.shard.dtype, device=row.device)
        _all_to_all(dist, rows_in, rows_out.contiguous(), send, recv, self.group)
        leaf.gather_scatter(out, rows_in, None, pos[:n_remote])
        return out

    def _exchange_counts(self, send):
        """send[p] = ids this rank asks of rank p -> ids every rank asks of this one (on the device under RCCL)."""
        on_dev = self.dist.get_backend(self.group) == "nccl"
        s = torch.tensor(send, dtype=torch.int64, device=self.shard.device if on_dev else "cpu")
        r = torch.zeros_like(s)
        self.dist.all_to_all_single(r, s, group=self.group)
        return [int(x) for x in r.tolist()]


def plan_replication(num_cached, row_bytes, world, hbm_budget_bytes):
    """How many of the hottest cached slots to keep on EVERY GPU (the rest is sharded modulo `world`), given the
    bytes of HBM one GPU may spend on feature rows.  The role the reference's PartitionSolver plays on NVLink
    (dist_graph.cu:40-222: replicas until no GPU fetches more than its links carry): on a fully connected xGMI node
    every remote row costs the same, so the best placement under a memory budget is simply the largest replicated
    prefix that fits --  R * row + ceil((num_cached - R) / world) * row <= budget.
    Returns R in [0, num_cached]; num_cached means 'replicate everything' (no xGMI traffic at all)."""
    if world <= 1 or num_cached * row_bytes <= hbm_budget_bytes:
        return num_cached
    rows = hbm_budget_bytes // row_bytes  # rows one GPU can hold
    # R + (num_cached - R) / world <= rows  ->  R <= (rows * world - num_cached) / (world - 1)
    r = (rows * world - num_cached) // (world - 1)
    return int(max(0, min(num_cached, r)))


def shard_rows(feat_rows_fn, rank_list, num_cached, world, rank, dim, dtype, device, shared=False):
    """Build this rank's shard: row k = feature of node rank_list[rank + k * world] (partition_feature).
    feat_rows_fn(node_ids, out) fills `out` with the rows of `node_ids`.  shared=True allocates an
    IPC-exportable shard (returns (tensor, SharedShard)), else a plain tensor (returns (tensor, None))."""
    mine = rank_list[rank:num_cached:world]
    n = int(mine.numel() if hasattr(mine, "numel") else len(mine))
    holder = None
    if shared:
        from . import ops
        holder = ops.SharedShard((max(n, 1), dim), dtype, device)
        t = holder.tensor
    else:
        t = torch.empty((max(n, 1), dim), dtype=dtype, device=device)
    if n:
        feat_rows_fn(mine, t[:n])
    return t, holder


# ---- GGMS topology shards (DistGraph, cuda/dist_graph.cu:228-385) -------------------------------------------------
def num_cache_node_for(indptr, fraction):
    """How many leading nodes hold `fraction` of the edges: dist_engine.cc:225 (num_cache_edge = num_edge * fraction)
    + DistGraph::GraphLoad's scan (dist_graph.cu:318-325).  indptr: uint32 host array."""
    num_edge = int(indptr[-1])
    cache_edge = np.uint32(int(num_edge * float(fraction)) & 0xFFFFFFFF)
    return int(np.searchsorted(indptr[:-1], cache_edge, side="left"))


def topology_shards(indptr, indices, num_part, num_cache_node, rows_per_step=1 << 24, only=None):
    """_DatasetPartition (dist_graph.cu:228-272) on the GPU: shard p = the CSR of the nodes v = p (mod num_part),
    v < num_cache_node, at rows v // num_part.  indptr / indices: int32 device tensors holding the uint32 CSR.
    Returns (part_indptr, part_indices): two lists of int32 device tensors -- the shards of every worker (one process
    that plays all of them: logical shards), or with only=p just worker p's own."""
    dev = indptr.device
    P = int(num_part)
    pip, pix = [], []
    for p in (range(P) if only is None else [int(only)]):
        nodes = torch.arange(p, num_cache_node, P, dtype=torch.int64, device=dev)
        start = indptr[nodes].to(torch.int64) & 0xFFFFFFFF
        end = indptr[nodes + 1].to(torch.int64) & 0xFFFFFFFF
        deg = end - start
        ip = torch.zeros(nodes.numel() + 1, dtype=torch.int64, device=dev)
        torch.cumsum(deg, 0, out=ip[1:])
        total = int(ip[-1].item())
        ix = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        # rows in blocks: the (row -> edge slots) expansion needs 8-byte temporaries per edge
        for lo in range(0, nodes.numel(), rows_per_step):
            hi = min(nodes.numel(), lo + rows_per_step)
            d = deg[lo:hi]
            n_e = int((ip[hi] - ip[lo]).item())
            if n_e == 0:
                continue
            # source slot of shard edge k of row j: start[j] + (k - ip[j])
            shift = torch.repeat_interleave(start[lo:hi] - ip[lo:hi], d)
            src = torch.arange(int(ip[lo].item()), int(ip[lo].item()) + n_e, dtype=torch.int64, device=dev) + shift
            ix[int(ip[lo].item()):int(ip[lo].item()) + n_e] = indices[src]
            del shift, src
        pip.append(ip.to(torch.int32))  # values < 2^32 kept bit for bit (uint32 in an int32 tensor)
        pix.append(ix)
    return pip, pix


class TopologyShards:
    """This rank's view of the sharded topology (DistGraph, cuda/dist_graph.cu:228-385): it builds shard `rank` of the
    leading num_cache_node nodes, publishes it (two hipIpc allocations: indptr, indices), maps every peer's, and hands
    out a DeviceGraph whose kernels read a peer's list heads and neighbour lists in place over xGMI.  slot: (indptr,
    indices) tensors of the whole CSR for the nodes beyond num_cache_node (device or registered host memory)."""

    def __init__(self, indptr, indices, world, rank, num_cache_node, dist, slot, group=None):
        from . import ops
        pip, pix = topology_shards(indptr, indices, world, num_cache_node, only=rank)
        self.holders = []
        tabs = []
        for name, t in (("topology indptr shard", pip[0]), ("topology indices shard", pix[0])):
            h = ops.SharedShard((max(1, t.numel()),), torch.int32, t.device)
            h.tensor[:t.numel()].copy_(t)
            self.holders.append(h)
            tabs.append(connect_shared(h, world, rank, dist, group, what=name))
        del pip, pix
        self.num_cache_node, self.world, self.rank = int(num_cache_node), world, rank
        self._slot = slot
        self.graph = ops.DeviceGraph(None, None, part_indptr=[_Addr(a) for a in tabs[0]] + [slot[0]],
                                     part_indices=[_Addr(a) for a in tabs[1]] + [slot[1]], num_cache_node=num_cache_node)

    def close(self):
        for h in self.holders:
            h.close()
        self.holders = []


class _Addr:
    """A device address standing in for a tensor where only data_ptr() / numel() are asked (peer mappings)."""

    def __init__(self, ptr):
        self._ptr = int(ptr)

    def data_ptr(self):
        return self._ptr

    def numel(self):
        return 0
