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

MAX_PARTS = 8  # GGMS_MAX_PARTS (include/ggms.h): shard base pointers travel in the kernels' arguments


def check_num_parts(num_part, what):
    """Refuse a group larger than the kernels carry BEFORE anything is allocated, exported or mapped (the reference's
    DeviceDistGraph / DeviceDistFeature take any num_part through a device pointer table, dist_graph.h:114-212; here
    the pointers are kernel arguments and the first gather of a larger group would return GGMS_ERR_INVALID)."""
    if int(num_part) > MAX_PARTS:
        raise ValueError(f"{what}: {num_part} shards, but the kernels carry at most GGMS_MAX_PARTS = {MAX_PARTS} shard "
                         "pointers (include/ggms.h); use the all-to-all store or a smaller group")


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


# ---- which GPUs reach which, and how fast (PartitionSolver::DetectTopo, cuda/dist_graph.cu:684-884) -----------------
class HipProbeLeaf:
    """What the link probe needs of the device, on HIP (include/ggms.h, "Link / topology probe")."""

    def __init__(self, device):
        import ctypes as C
        from . import ops
        from ._lib import check, lib
        self.C, self.ops, self.check, self.lib, self.device = C, ops, check, lib, torch.device(device)

    def peer_access(self, dev, peer):
        v = self.C.c_int(0)
        self.check(self.lib().ggms_peer_access(int(dev), int(peer), self.C.byref(v)), "ggms_peer_access")
        return int(v.value)

    def shard(self, rows, row_words, fill):
        h = self.ops.SharedShard((rows, row_words), torch.int32, self.device)
        h.tensor.fill_(fill)
        return h

    def scratch(self, nbytes):
        return torch.empty(max(4, nbytes) // 4, dtype=torch.int32, device=self.device)

    def copy_rate(self, dst, src_ptr, nbytes, reps, with_kernel=0):
        C, g = self.C, self.C.c_double(0)
        self.check(self.lib().ggms_link_probe_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src_ptr), nbytes, reps, with_kernel,
                                                   C.byref(g), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "ggms_link_probe_copy")
        return g.value

    def gather_rate(self, out, part_ptrs, rows_per_part, row_bytes, num_rows, seed, reps, index_ws):
        C, g = self.C, self.C.c_double(0)
        tab = self.ops.PartTable(part_ptrs)
        self.check(self.lib().ggms_link_probe_gather(C.c_void_p(out.data_ptr()), tab.ptr(), len(part_ptrs), rows_per_part,
                                                     row_bytes, num_rows, seed, reps, C.c_void_p(index_ws.data_ptr()),
                                                     C.byref(g), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "ggms_link_probe_gather")
        return g.value

    def first_word(self, t):
        return int(t[0].item())


def peer_access_preflight(world, rank, dist, device_index, leaf, group=None):
    """hipDeviceCanAccessPeer for every pair of the group's GPUs (cudaDeviceCanAccessPeer, dist_graph.cu:812-818),
    BEFORE any shard is built: rank r asks for its own row, the rows are gathered, every rank holds the same matrix
    and takes the same turn.  Ranks that share a device (one-GPU rehearsal) reach each other by definition.
    -> {"devices": [device index per rank], "can_access": [[0/1]], "refused": [[reader, owner], ...]}"""
    devs = [None] * world
    if world > 1:
        with_deadline(lambda: dist.all_gather_object(devs, int(device_index), group=group),
                      f"rank {rank} of {world} exchanging device indices for the peer-access preflight")
    else:
        devs = [int(device_index)]
    row = [1 if devs[r] == devs[rank] else leaf.peer_access(devs[rank], devs[r]) for r in range(world)]
    rows = [None] * world
    if world > 1:
        with_deadline(lambda: dist.all_gather_object(rows, row, group=group),
                      f"rank {rank} of {world} exchanging the peer-access rows")
    else:
        rows = [row]
    refused = [[i, j] for i in range(world) for j in range(world) if i != j and not rows[i][j]]
    return {"devices": devs, "can_access": rows, "refused": refused}


def link_probe(world, rank, dist, leaf, probe_bytes=128 << 20, row_bytes=512, reps=3, group=None):
    """What a rank's xGMI links carry, measured the way the product uses them (one process per GPU, peers read in place
    through hipIpc mappings) -- the one-process-per-GPU form of DetectTopo_child's timed 128-MiB peer copies
    (dist_graph.cu:822-848).  Every rank publishes a probe_bytes buffer and maps its peers'; then
      per pair, ALONE on the node (everybody else waits at a barrier): a timed copy out of the mapping
        (hipMemcpyAsync, the copy engines), a streaming copy kernel reading it in place (what in-kernel loads over
        the link can reach at best) and the feature store's own gather kernel reading random row_bytes rows of it
        (ggms_link_probe_gather);
      all ranks at once, each from ALL its peers (slots modulo the peers, what the `peer` / `hybrid` stores do every
        batch): the rate a GPU's inbound links sustain together while its outbound links serve the others.
    Matrices are [reader][owner]; the diagonal is the reader's own HBM through the same code.  Returns the same dict
    on every rank.  A refused mapping raises PeerConnectError on every rank (connect_shared)."""
    import time
    t0 = time.perf_counter()
    rows_per_part = probe_bytes // row_bytes
    nbytes = rows_per_part * row_bytes
    holder = leaf.shard(rows_per_part, row_bytes // 4, rank + 1)
    try:
        ptrs = connect_shared(holder, world, rank, dist, group, what="link-probe buffer")
        dst = leaf.scratch(nbytes)
        num_rows = 2 * rows_per_part  # rows gathered per launch: twice the buffer's rows, drawn at random
        out = leaf.scratch(num_rows * row_bytes)
        index_ws = leaf.scratch(num_rows * 4)

        def barrier():
            if world > 1:
                dist.barrier(group=group)

        copy_row, stream_row, gather_row, wrong = [0.0] * world, [0.0] * world, [0.0] * world, []
        barrier()
        for reader in range(world):
            for owner in range(world):
                if reader == rank:
                    copy_row[owner] = leaf.copy_rate(dst, ptrs[owner], nbytes, reps)
                    if leaf.first_word(dst) != owner + 1:  # the mapping reads the owner's memory, not somebody else's
                        wrong.append(owner)
                    stream_row[owner] = leaf.copy_rate(dst, ptrs[owner], nbytes, reps, 1)
                    gather_row[owner] = leaf.gather_rate(out, [ptrs[owner]], rows_per_part, row_bytes, num_rows,
                                                         17 * reader + owner, reps, index_ws)
                barrier()
        peers = [ptrs[p] for p in range(world) if p != rank]
        inbound = leaf.gather_rate(out, peers, rows_per_part, row_bytes, num_rows, 1000 + rank, reps, index_ws) if peers else 0.0
        barrier()  # nobody unmaps a buffer a peer may still be reading
        mine = {"copy": copy_row, "stream": stream_row, "gather": gather_row, "inbound": inbound, "wrong": wrong}
        rows = [None] * world
        if world > 1:
            with_deadline(lambda: dist.all_gather_object(rows, mine, group=group),
                          f"rank {rank} of {world} collecting the link-probe rows")
        else:
            rows = [mine]
    finally:
        holder.close()
    bad = [[r, o] for r in range(world) for o in rows[r]["wrong"]]
    if bad:
        raise PeerConnectError(f"link probe: mappings that do not show their owner's memory [reader, owner]: {bad}")
    remote = [rows[r]["gather"][o] for r in range(world) for o in range(world) if o != r]
    return {
        "method": "one rank per GPU; every rank publishes a buffer (hipIpc) and maps its peers'; per pair alone on the "
                  "node: hipMemcpyAsync out of the mapping, a 16-B-per-lane streaming copy kernel reading it in place, and the "
                  "feature store's gather kernel (ggms_gather_scatter_partition, 16-B nt loads) on random rows of it; then all "
                  "ranks at once, each from "
                  "all its peers (slots modulo the peers); matrices are [reader][owner], the diagonal is local HBM",
        "probe_bytes": nbytes, "row_bytes": row_bytes, "rows_per_launch": num_rows, "reps": reps,
        "per_pair_copy_GBps": [r["copy"] for r in rows],
        "per_pair_stream_kernel_GBps": [r["stream"] for r in rows],
        "per_pair_gather_GBps": [r["gather"] for r in rows],
        "inbound_all_peers_gather_GBps": [r["inbound"] for r in rows],
        "inbound_all_peers_min_GBps": min((r["inbound"] for r in rows), default=0.0) if world > 1 else None,
        "per_pair_gather_min_GBps": min(remote) if remote else None,
        "per_pair_gather_max_GBps": max(remote) if remote else None,
        "seconds": round(time.perf_counter() - t0, 2),
    }


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
        if mode == "peer":
            check_num_parts(world, "FeatureShards(peer)")
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


def plan_with_links(num_cached, row_bytes, world, r_budget, inbound_GBps, local_GBps, capacity_bytes, margin=0.8):
    """The placement once the links have been MEASURED (ggms_store.link_probe) -- what the reference's PartitionSolver
    does with its bandwidth matrix (dist_graph.cu:40-222: replicas until no GPU fetches more than its links carry).
    One gather kernel streams a batch's local rows from HBM and its remote rows over xGMI at the same time; the remote
    part stays off the critical path while  remote_bytes / inbound <= local_bytes / local,  i.e. while the share of a
    batch's rows that is remote stays below  f = inbound / (inbound + local)  (`margin` of it is used).  With the sharded
    tail holding a fraction t of the (uniformly requested) slots, (world - 1) / world of its requests are remote:
    t <= f * world / (world - 1).  The replica is the LARGER of the budget plan's (r_budget, plan_replication) and the
    one this bound asks for, capped by what the GPU can really hold (capacity_bytes).
    -> (R, record)"""
    if world <= 1 or not inbound_GBps or not local_GBps:
        return int(r_budget), None
    f = margin * inbound_GBps / (inbound_GBps + local_GBps)
    tail = min(1.0, f * world / (world - 1))
    r_link = min(int(num_cached), int(np.ceil(num_cached * (1.0 - tail))))
    r_cap = plan_replication(num_cached, row_bytes, world, int(capacity_bytes))
    r = int(min(max(int(r_budget), r_link), r_cap))
    return r, {"inbound_GBps": inbound_GBps, "local_gather_GBps": local_GBps, "margin": margin,
               "remote_row_share_the_links_hide": f, "sharded_tail_max_fraction": tail,
               "replicated_rows_budget_plan": int(r_budget), "replicated_rows_link_plan": r_link,
               "replicated_rows_capacity": int(r_cap), "replicated_rows_chosen": r,
               "rule": "remote share <= margin * inbound / (inbound + local): the remote rows of a batch arrive while its "
                       "local rows stream; R = min(max(budget plan, link plan), capacity)"}


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


def topology_shards(indptr, indices, num_part, num_cache_node, edges_per_step=1 << 26, only=None):
    """_DatasetPartition (dist_graph.cu:228-272) on the GPU: shard p = the CSR of the nodes v = p (mod num_part),
    v < num_cache_node, at rows v // num_part.  indptr / indices: int32 device tensors holding the uint32 CSR.
    Returns (part_indptr, part_indices): two lists of int32 device tensors -- the shards of every worker (one process
    that plays all of them: logical shards), or with only=p just worker p's own.
    Scratch is bounded by EDGES, not rows: a block moves at most about edges_per_step neighbour ids (20 B of
    temporaries each: 1.3 GB at the default) whatever the degrees are, and a shard costs two host syncs in all."""
    dev = indptr.device
    P = int(num_part)
    pip, pix = [], []
    for p in (range(P) if only is None else [int(only)]):
        nodes = torch.arange(p, max(p, int(num_cache_node)), P, dtype=torch.int64, device=dev)
        n = nodes.numel()
        start = indptr[nodes].to(torch.int64) & 0xFFFFFFFF
        ip = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        if n:
            torch.cumsum((indptr[nodes + 1].to(torch.int64) & 0xFFFFFFFF) - start, 0, out=ip[1:])
        del nodes
        total = int(ip[-1].item())  # host sync 1: the shard's edge count sizes its allocation
        ix = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        if total:
            # block boundaries: the first row whose list starts at or beyond each multiple of edges_per_step
            targets = torch.arange(1, (total - 1) // edges_per_step + 1, dtype=torch.int64, device=dev) * edges_per_step
            cuts = torch.unique(torch.cat([torch.zeros(1, dtype=torch.int64, device=dev),
                                           torch.searchsorted(ip, targets), torch.full((1,), n, dtype=torch.int64, device=dev)]))
            rows = cuts.tolist()          # host sync 2
            offs = ip[cuts].tolist()
            for (lo, hi), (e_lo, e_hi) in zip(zip(rows[:-1], rows[1:]), zip(offs[:-1], offs[1:])):
                if e_hi == e_lo:
                    continue
                # source slot of shard edge k of row j: start[j] + (k - ip[j])
                src = torch.repeat_interleave(start[lo:hi] - ip[lo:hi], ip[lo + 1:hi + 1] - ip[lo:hi], output_size=e_hi - e_lo)
                src += torch.arange(e_lo, e_hi, dtype=torch.int64, device=dev)
                ix[e_lo:e_hi] = indices[src]
                del src
        del start
        pip.append(ip.to(torch.int32))  # values < 2^32 kept bit for bit (uint32 in an int32 tensor)
        del ip
        pix.append(ix)
    return pip, pix


class TopologyShards:
    """This rank's view of the sharded topology (DistGraph, cuda/dist_graph.cu:228-385): it builds shard `rank` of the
    leading num_cache_node nodes, publishes it (two hipIpc allocations: indptr, indices), maps every peer's, and hands
    out a DeviceGraph whose kernels read a peer's list heads and neighbour lists in place over xGMI.  slot: (indptr,
    indices) tensors of the whole CSR for the nodes beyond num_cache_node (device or registered host memory).

    Every failure is a verdict of the whole group: a rank that cannot build or allocate its shard (out of memory)
    still walks through the handle exchange as a FailedShard, so that EVERY rank raises PeerConnectError together
    instead of the others sitting out a deadline in all_gather_object; whatever this rank did allocate or map before
    the verdict is released before the exception leaves.  shard_alloc (tests): stands in for ops.SharedShard."""

    def __init__(self, indptr, indices, world, rank, num_cache_node, dist, slot, group=None, shard_alloc=None,
                 build=None):
        check_num_parts(world, "TopologyShards")  # before anything is allocated, exported or mapped
        if shard_alloc is None:
            from . import ops
            shard_alloc = ops.SharedShard
        self.holders = []
        self.num_cache_node, self.world, self.rank = int(num_cache_node), world, rank
        self._slot = slot
        names = ("topology indptr shard", "topology indices shard")
        failure = None
        try:
            pip, pix = (build or topology_shards)(indptr, indices, world, num_cache_node, only=rank)
            for t in (pip[0], pix[0]):
                h = shard_alloc((max(1, t.numel()),), torch.int32, t.device)
                self.holders.append(h)
                h.tensor[:t.numel()].copy_(t)
            del pip, pix, t
        except (RuntimeError, MemoryError) as e:  # torch.cuda.OutOfMemoryError, GgmsError (ggms_device_alloc), ...
            failure = f"{type(e).__name__}: {str(e)[:200]}"
            self.close()
            pip = pix = None
            if torch.cuda.is_available() and torch.cuda.is_initialized():
                torch.cuda.empty_cache()
        tabs = []
        try:
            for i, name in enumerate(names):
                # a failed build takes part in the FIRST exchange with its reason: that exchange raises on every rank
                shard = FailedShard(failure) if failure is not None else self.holders[i]
                tabs.append(connect_shared(shard, world, rank, dist, group, what=name))
        except BaseException:
            self.close()  # the indptr holder and its peer mappings when the second exchange is the one that failed
            raise
        from . import ops
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
