"""Tensor-level wrappers over the C ABI (include/ggms.h).

PyTorch-ROCm is used for device memory and streams only; every operator below
is one or more calls into libggms_hip.so with raw device pointers.  Ids are
stored in int32 tensors (the reference hands int32 to PyTorch too,
samgraph/torch/adapter.cc:103,117) and are bit-identical to uint32.
"""
import ctypes as C

import numpy as np

import torch

from . import _lib
from ._lib import Graph as _CGraph, HashTable as _CHashTable, check, lib

EMPTY_KEY = 0xFFFFFFFF

# DataType codes, samgraph/common/common.h:38-46
DTYPE_CODE = {
    torch.float32: 0, torch.float64: 1, torch.float16: 2, torch.uint8: 3,
    torch.int32: 4, torch.int8: 5, torch.int64: 6, torch.int16: 2, torch.bfloat16: 2,
}

KHOP0, KHOP1, WEIGHTED_KHOP, RANDOM_WALK, WEIGHTED_KHOP_PREFIX, KHOP2, WEIGHTED_KHOP_HASH_DEDUP, KHOP3 = range(8)


def _require_gpu(t):
    if not t.is_cuda:
        raise _lib.GgmsError("xgnn_amd operators need device tensors (no CPU fallback)")


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _i32(t):
    assert t.dtype == torch.int32 and t.is_contiguous(), "ids must be contiguous int32 tensors"
    return t


class DeviceGraph:
    """ggms_graph_t over device tensors: DeviceNormalGraph or DeviceDistGraph (dist_graph.h:114-180)."""

    def __init__(self, indptr, indices, part_indptr=None, part_indices=None, num_cache_node=0):
        self.indptr, self.indices = indptr, indices
        self._keep = [indptr, indices]
        num_node = (indptr.numel() - 1) if indptr is not None else 0
        self.c = _CGraph()
        self.c.indptr = indptr.data_ptr() if indptr is not None else None
        self.c.indices = indices.data_ptr() if indices is not None else None
        self.c.num_node = num_node
        self.c.num_part = 0
        self.c.num_cache_node = 0
        if part_indptr is not None:
            # part_* : lists of P+1 tensors (slot P = whole CSR, device memory or pinned / registered host memory);
            # the pointer tables are HOST arrays -- the kernels take the pointers by value (include/ggms.h)
            self._pip = PartTable([t.data_ptr() for t in part_indptr])
            self._pix = PartTable([t.data_ptr() for t in part_indices])
            self._keep += list(part_indptr) + list(part_indices)
            self.c.part_indptr = self._pip.ptr().value
            self.c.part_indices = self._pix.ptr().value
            self.c.num_part = len(part_indptr) - 1
            self.c.num_cache_node = num_cache_node
            self.c.num_node = part_indptr[-1].numel() - 1


def random_states(num_states, seed, device="cuda"):
    """GPURandomStates (cuda_random_states.cu:64-109) with an explicit seed."""
    st = torch.empty((num_states, 6), dtype=torch.int32, device=device)
    _require_gpu(st)
    check(lib().ggms_random_states_init(_ptr(st), num_states, seed, _stream()), "ggms_random_states_init")
    return st


def device_status(clear=False):
    """The device status word (include/ggms.h): 0 = ok, 1 = a scan's look-back gave up, 2 = hashed table full.
    Synchronises the device."""
    st = C.c_uint32(0)
    check(lib().ggms_device_status(C.byref(st), 1 if clear else 0), "ggms_device_status")
    return st.value


def check_device_status(what=""):
    """Raise if a kernel reported a device-side failure (the reference CHECK-aborts there, logging.cc:69-73)."""
    st = device_status(clear=True)
    if st:
        raise _lib.GgmsError(f"{what}: device status {st:#x} "
                             f"({'scan look-back gave up ' if st & 1 else ''}{'hashed dedup table full' if st & 2 else ''})")


def _workspace(nbytes, device):
    return torch.empty(max(16, (nbytes + 3) // 4), dtype=torch.int32, device=device)


def _sample(fn_name, sample_type, graph, inp, fanout, states):
    _require_gpu(inp)
    _i32(inp)
    n = inp.numel()
    dev = inp.device
    out_src = torch.empty(max(1, n * fanout), dtype=torch.int32, device=dev)
    out_dst = torch.empty(max(1, n * fanout), dtype=torch.int32, device=dev)
    num_out = torch.zeros(1, dtype=torch.int64, device=dev)
    wsb = lib().ggms_sample_workspace_bytes(sample_type, n, fanout)
    ws = _workspace(wsb, dev)
    if fn_name in ("ggms_sample_khop3", "ggms_sample_khop2"):
        rc = getattr(lib(), fn_name)(C.byref(graph.c), _ptr(inp), n, fanout, _ptr(out_src), _ptr(out_dst),
                                     _ptr(num_out), _ptr(states), states.shape[0], _ptr(ws), ws.numel() * 4,
                                     _stream())
    else:
        rc = lib().ggms_sample_khop0(C.byref(graph.c), _ptr(inp), n, fanout, _ptr(out_src), _ptr(out_dst),
                                     _ptr(num_out), _ptr(ws), ws.numel() * 4, _stream())
    check(rc, fn_name)
    return out_src, out_dst, num_out


def sample_khop3(graph, inp, fanout, states):
    """GPUSampleKHop3 (cuda_sampling_khop3.cu:234-318).  Returns (out_src, out_dst, num_out[1] on device)."""
    return _sample("ggms_sample_khop3", KHOP3, graph, inp, fanout, states)


def sample_khop0(graph, inp, fanout):
    """GPUSampleKHop0 (cuda_sampling_khop0.cu:243-335)."""
    return _sample("ggms_sample_khop0", KHOP0, graph, inp, fanout, None)


def sample_weighted_khop_prefix(graph, prob_prefix_table, inp, fanout, states):
    """GPUSampleWeightedKHopPrefix (cuda_sampling_weighted_khop_prefix.cu:145-246)."""
    _require_gpu(inp)
    _i32(inp)
    n = inp.numel()
    dev = inp.device
    out_src = torch.empty(max(1, n * fanout), dtype=torch.int32, device=dev)
    out_dst = torch.empty(max(1, n * fanout), dtype=torch.int32, device=dev)
    num_out = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = _workspace(lib().ggms_sample_weighted_workspace_bytes(n, fanout), dev)
    check(lib().ggms_sample_weighted_khop_prefix(C.byref(graph.c), _ptr(prob_prefix_table), _ptr(inp), n, fanout,
                                                 _ptr(out_src), _ptr(out_dst), _ptr(num_out), _ptr(states),
                                                 states.shape[0], _ptr(ws), ws.numel() * 4, _stream()),
          "ggms_sample_weighted_khop_prefix")
    return out_src, out_dst, num_out


def sample_weighted_khop_hash_dedup(graph, prob_table, alias_table, inp, fanout, states):
    """GPUSampleWeightedKHopHashDedup (cuda_sampling_weighted_khop_hash_dedup.cu:196-283)."""
    _require_gpu(inp)
    _i32(inp)
    n = inp.numel()
    dev = inp.device
    out_src = torch.empty(max(1, n * fanout), dtype=torch.int32, device=dev)
    out_dst = torch.empty(max(1, n * fanout), dtype=torch.int32, device=dev)
    num_out = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = _workspace(lib().ggms_sample_workspace_bytes(WEIGHTED_KHOP_HASH_DEDUP, n, fanout), dev)
    check(lib().ggms_sample_weighted_khop_hash_dedup(C.byref(graph.c), _ptr(prob_table), _ptr(alias_table), _ptr(inp),
                                                     n, fanout, _ptr(out_src), _ptr(out_dst), _ptr(num_out),
                                                     _ptr(states), states.shape[0], _ptr(ws), ws.numel() * 4,
                                                     _stream()), "ggms_sample_weighted_khop_hash_dedup")
    return out_src, out_dst, num_out


def sample_khop1(graph, inp, fanout, states):
    """GPUSampleKHop1 (cuda_sampling_khop1.cu:130-236): uniform with replacement, adjacent duplicates dropped."""
    _require_gpu(inp)
    _i32(inp)
    n = inp.numel()
    dev = inp.device
    out_src = torch.empty(max(1, n * fanout), dtype=torch.int32, device=dev)
    out_dst = torch.empty(max(1, n * fanout), dtype=torch.int32, device=dev)
    num_out = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = _workspace(lib().ggms_sample_weighted_workspace_bytes(n, fanout), dev)
    check(lib().ggms_sample_khop1(C.byref(graph.c), _ptr(inp), n, fanout, _ptr(out_src), _ptr(out_dst), _ptr(num_out),
                                  _ptr(states), states.shape[0], _ptr(ws), ws.numel() * 4, _stream()),
          "ggms_sample_khop1")
    return out_src, out_dst, num_out


def sample_khop2(graph, inp, fanout, states):
    """GPUSampleKHop2 (cuda_sampling_khop2.cu:196-262).  Permutes graph.indices in place, like the reference."""
    return _sample("ggms_sample_khop2", KHOP2, graph, inp, fanout, states)


def sample_weighted_khop(graph, prob_table, alias_table, inp, fanout, states):
    """GPUSampleWeightedKHop (cuda_sampling_weighted_khop.cu:132-238)."""
    _require_gpu(inp)
    _i32(inp)
    n = inp.numel()
    dev = inp.device
    out_src = torch.empty(max(1, n * fanout), dtype=torch.int32, device=dev)
    out_dst = torch.empty(max(1, n * fanout), dtype=torch.int32, device=dev)
    num_out = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = _workspace(lib().ggms_sample_weighted_workspace_bytes(n, fanout), dev)
    check(lib().ggms_sample_weighted_khop(C.byref(graph.c), _ptr(prob_table), _ptr(alias_table), _ptr(inp), n, fanout,
                                          _ptr(out_src), _ptr(out_dst), _ptr(num_out), _ptr(states), states.shape[0],
                                          _ptr(ws), ws.numel() * 4, _stream()), "ggms_sample_weighted_khop")
    return out_src, out_dst, num_out


def sample_random_walk(graph, inp, walk_length, restart_prob, num_walk, K, states):
    """GPUSampleRandomWalk + FrequencyHashmap::GetTopK (cuda_sampling_random_walk.cu:116-165)."""
    _require_gpu(inp)
    _i32(inp)
    n = inp.numel()
    dev = inp.device
    outs = [torch.empty(max(1, n * K), dtype=torch.int32, device=dev) for _ in range(3)]
    num_out = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = _workspace(lib().ggms_sample_random_walk_workspace_bytes(n, walk_length, num_walk, K), dev)
    check(lib().ggms_sample_random_walk(C.byref(graph.c), _ptr(inp), n, walk_length, restart_prob, num_walk, K,
                                        _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _ptr(num_out), _ptr(states),
                                        states.shape[0], _ptr(ws), ws.numel() * 4, _stream()),
          "ggms_sample_random_walk")
    return outs[0], outs[1], outs[2], num_out


class OrderedHashTable:
    """OrderedHashTable (cuda_hashtable.h:103-153) over caller-owned device buffers."""

    def __init__(self, capacity, device="cuda", num_node=None):
        """num_node=None: hashed layout sized like the reference (TableSize); num_node=N: direct-mapped
        layout, one 8-byte word per node id (the engine's choice on MI355X)."""
        direct = num_node is not None
        nb = int(num_node) if direct else lib().ggms_hashtable_num_buckets(capacity)
        self.o2n = torch.empty((nb, 2 if direct else 4), dtype=torch.int32, device=device)
        _require_gpu(self.o2n)
        self.n2o = torch.empty(max(1, capacity), dtype=torch.int32, device=device)
        self.num_items_dev = torch.zeros(2, dtype=torch.int32, device=device)  # [item count, batch status word]
        self.c = _CHashTable()
        self.c.o2n = self.o2n.data_ptr()
        self.c.n2o = self.n2o.data_ptr()
        self.c.num_items_dev = self.num_items_dev.data_ptr()
        self.c.o2n_size = nb
        self.c.n2o_size = self.n2o.numel()
        self.c.direct = 1 if direct else 0
        check(lib().ggms_hashtable_init(C.byref(self.c), _stream()), "ggms_hashtable_init")

    def reset(self):
        check(lib().ggms_hashtable_reset(C.byref(self.c), _stream()), "ggms_hashtable_reset")

    def fill_with_duplicates(self, inp, num_input=None, unique_out=None):
        _i32(inp)
        n = inp.numel() if num_input is None else int(num_input)
        ws = _workspace(lib().ggms_hashtable_workspace_bytes(n), inp.device)
        check(lib().ggms_hashtable_fill_with_duplicates(C.byref(self.c), _ptr(inp), n, _ptr(unique_out), _ptr(ws),
                                                        ws.numel() * 4, _stream()),
              "ggms_hashtable_fill_with_duplicates")

    @property
    def num_items(self):
        return int(self.num_items_dev[0].item())

    def unique(self):
        return self.n2o[: self.num_items]

    def map_edges(self, src, dst, num_edges=None):
        n = (src if src is not None else dst).numel() if num_edges is None else int(num_edges)
        dev = (src if src is not None else dst).device
        ns = torch.empty(max(1, n), dtype=torch.int32, device=dev) if src is not None else None
        nd = torch.empty(max(1, n), dtype=torch.int32, device=dev) if dst is not None else None
        check(lib().ggms_map_edges(C.byref(self.c), _ptr(src), _ptr(ns), _ptr(dst), _ptr(nd), n, _stream()),
              "ggms_map_edges")
        return (ns[:n] if ns is not None else None), (nd[:n] if nd is not None else None)


def _dim_of(t):
    return t.shape[1] if t.dim() > 1 else 1


def extract(src, index, out=None):
    """GPUExtract (cuda_extraction.cu:74-117): out[i, :] = src[index[i], :]."""
    _require_gpu(index)
    _i32(index)
    assert src.is_contiguous()
    n = index.numel()
    if out is None:
        out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=index.device)
    check(lib().ggms_extract(_ptr(out), _ptr(src), _ptr(index), n, _dim_of(src), DTYPE_CODE[src.dtype], _stream()),
          "ggms_extract")
    return out


def mock_extract(src, index, mock_bits, out=None):
    """GPUMockExtract (cuda_extraction.cu:119-160): out[i, :] = src[index[i] & (2**mock_bits - 1), :]."""
    _require_gpu(index)
    _i32(index)
    n = index.numel()
    if out is None:
        out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=index.device)
    check(lib().ggms_mock_extract(_ptr(out), _ptr(src), _ptr(index), n, _dim_of(src), DTYPE_CODE[src.dtype], mock_bits,
                                  _stream()), "ggms_mock_extract")
    return out


def get_miss_cache_index(table, nodes):
    """GetMissCacheIndex (cuda_cache_manager_device.cu:355-441).  Counts stay on the device."""
    _require_gpu(nodes)
    n = nodes.numel()
    dev = nodes.device
    outs = [torch.empty(max(1, n), dtype=torch.int32, device=dev) for _ in range(4)]
    num_miss = torch.zeros(1, dtype=torch.int64, device=dev)
    num_hit = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = _workspace(lib().ggms_cache_index_workspace_bytes(n), dev)
    check(lib().ggms_get_miss_cache_index(_ptr(table), _ptr(nodes), n, _ptr(outs[0]), _ptr(outs[1]), _ptr(num_miss),
                                          _ptr(outs[2]), _ptr(outs[3]), _ptr(num_hit), _ptr(ws), ws.numel() * 4,
                                          _stream()), "ggms_get_miss_cache_index")
    return outs[0], outs[1], num_miss, outs[2], outs[3], num_hit


def gather_scatter(out, src, src_index, dst_index, num=None, num_dev=None):
    """combine_cache_data / extract_miss_data / combine_miss_data (cuda_cache_manager_device.cu:209-275)."""
    _require_gpu(out)
    if num is None:
        num = (src_index if src_index is not None else dst_index).numel()
    check(lib().ggms_gather_scatter(_ptr(out), _ptr(src), _ptr(src_index), _ptr(dst_index), num, _ptr(num_dev),
                                    _dim_of(out), DTYPE_CODE[out.dtype], _stream()), "ggms_gather_scatter")
    return out


class PartTable:
    """HOST array of shard base pointers (DeviceDistFeature / DeviceDistGraph, dist_graph.h:114-212): the operators
    hand the pointers to their kernels by value, at most GGMS_MAX_PARTS = 8 shards."""

    def __init__(self, ptrs, keep=None):
        if len(ptrs) > 9:  # GGMS_MAX_PARTS shards + the topology's host slot
            raise ValueError(f"PartTable: {len(ptrs)} pointers, the kernels carry at most GGMS_MAX_PARTS = 8 shards "
                             "(+ the host slot of a graph view)")
        self.n = len(ptrs)
        self.arr = (C.c_void_p * max(1, self.n))(*[int(p) for p in ptrs])
        self._keep = keep

    def ptr(self):
        return C.cast(self.arr, C.c_void_p)


def part_pointer_table(parts, device=None):
    """Pointer table of a list of shard tensors (kept alive by the table)."""
    return PartTable([p.data_ptr() for p in parts], keep=list(parts))


def gather_scatter_partition(out, parts_table, num_part, src_index, dst_index, num=None, num_dev=None):
    """combine_cache_data_for_partition (cuda_cache_manager_device.cu:277-299)."""
    _require_gpu(out)
    if num is None:
        num = src_index.numel()
    check(lib().ggms_gather_scatter_partition(_ptr(out), parts_table.ptr(), num_part, _ptr(src_index),
                                              _ptr(dst_index), num, _ptr(num_dev), _dim_of(out),
                                              DTYPE_CODE[out.dtype], _stream()), "ggms_gather_scatter_partition")
    return out


def extract_cached(out, nodes, table, parts_table, num_part, host_feat, num=None, num_dev=None, num_miss=None):
    """Fused GetMissCacheIndex + GPUExtractMissData + CombineCacheData (dist_loops.cc:1209-1285).
    table=None: full cache kept in node order (slot = node id), no table read per row."""
    _require_gpu(out)
    if num is None:
        num = nodes.numel()
    check(lib().ggms_extract_cached(_ptr(out), _ptr(nodes), num, _ptr(num_dev), _ptr(table), parts_table.ptr(),
                                    num_part, _ptr(host_feat), _dim_of(out), DTYPE_CODE[out.dtype], _ptr(num_miss),
                                    _stream()), "ggms_extract_cached")
    return out


def extract_tiered(out, nodes, table, replica, parts_table, num_part, my_part, host_feat, num=None, num_dev=None,
                   tier_rows=None):
    """Every tier of the GGMS store in one gather (include/ggms.h ggms_extract_tiered): replica of the hottest
    slots, local / peer shards (slot - |replica| modulo num_part), pinned host rows for uncached nodes.
    tier_rows: int64[4] device counters {host, remote shard, local shard, replica}, added to."""
    _require_gpu(out)
    if num is None:
        num = nodes.numel()
    t = _lib.FeatureTiers()
    t.table = table.data_ptr() if table is not None else None
    t.replica = replica.data_ptr() if replica is not None else None
    t.num_replica = replica.shape[0] if replica is not None else 0
    t.parts = parts_table.ptr().value
    t.num_part, t.my_part = num_part, my_part
    t.host_feat = host_feat.data_ptr() if host_feat is not None else None
    check(lib().ggms_extract_tiered(_ptr(out), _ptr(nodes), num, _ptr(num_dev), C.byref(t), _dim_of(out),
                                    DTYPE_CODE[out.dtype], _ptr(tier_rows), _stream()), "ggms_extract_tiered")
    return out


class LaunchTimer:
    """ggms_launch_timer_t (include/ggms.h): the next row gather issued by this thread carries the timer's two events ON
    its dispatch packet -- the kernel's own start / end timestamps, no marker packet on the stream.  `wait(stream)` makes
    another stream wait for that launch; `elapsed_us()` blocks until it has finished."""

    def __init__(self):
        self._h = C.c_void_p()
        check(lib().ggms_launch_timer_create(C.byref(self._h)), "ggms_launch_timer_create")

    def arm(self):
        check(lib().ggms_launch_timer_arm(self._h), "ggms_launch_timer_arm")
        return self

    def wait(self, stream=None):
        s = _stream() if stream is None else C.c_void_p(stream.cuda_stream)
        check(lib().ggms_launch_timer_wait(self._h, s), "ggms_launch_timer_wait")

    def elapsed_us(self):
        us = C.c_double(0)
        check(lib().ggms_launch_timer_elapsed_us(self._h, C.byref(us)), "ggms_launch_timer_elapsed_us")
        return us.value

    def span_us(self, last):
        """From the start of this timer's launch to the end of `last`'s (both waited for)."""
        us = C.c_double(0)
        check(lib().ggms_launch_timer_span_us(self._h, last._h, C.byref(us)), "ggms_launch_timer_span_us")
        return us.value

    def close(self):
        if self._h:
            lib().ggms_launch_timer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


def owner_histogram(table, nodes, num_part, slots_out, counts, num=None, num_dev=None):
    """slots_out[i] = table[nodes[i]]; counts[p] += rows of the batch owned by shard p (p = num_part: host tier)."""
    _require_gpu(nodes)
    n = nodes.numel() if num is None else num
    check(lib().ggms_owner_histogram(_ptr(table), _ptr(nodes), n, _ptr(num_dev), num_part, _ptr(slots_out),
                                     _ptr(counts), _stream()), "ggms_owner_histogram")


def owner_bucket(slots, nodes, num_part, cursor, bucket_row, bucket_pos, num=None, num_dev=None):
    """Group the batch by owning shard: bucket p starts at cursor[p] (exclusive prefix of the histogram)."""
    _require_gpu(nodes)
    n = nodes.numel() if num is None else num
    check(lib().ggms_owner_bucket(_ptr(slots), _ptr(nodes), n, _ptr(num_dev), num_part, _ptr(cursor),
                                  _ptr(bucket_row), _ptr(bucket_pos), _stream()), "ggms_owner_bucket")


class _RawDevice:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


_TYPESTR = {torch.float32: "<f4", torch.float64: "<f8", torch.float16: "<f2", torch.uint8: "|u1",
            torch.int32: "<i4", torch.int8: "|i1", torch.int64: "<i8"}


class RegisteredHost:
    """A host array registered for device access (hipHostRegister, mapped): what DistGraph keeps in its last slot --
    the whole CSR in host memory, read by the sampling kernels over PCIe (dist_graph.cu:367-381) -- and what the
    `gpu_extract` miss tier is (dist_engine.cc:217-241).  `.tensor` aliases the host pages as a device tensor."""

    def __init__(self, array, device="cuda"):
        self.array = np.ascontiguousarray(array)  # kept alive: the registration pins THESE pages
        hip = C.CDLL("libamdhip64.so")
        self._hip = hip
        nbytes = self.array.nbytes
        self._host = C.c_void_p(self.array.ctypes.data)
        with torch.cuda.device(torch.device(device)):
            rc = hip.hipHostRegister(self._host, C.c_size_t(nbytes), C.c_uint(2))  # hipHostRegisterMapped
            if rc != 0:
                raise _lib.GgmsError(f"hipHostRegister({nbytes} bytes) -> {rc}")
            dp = C.c_void_p()
            rc = hip.hipHostGetDevicePointer(C.byref(dp), self._host, C.c_uint(0))
            if rc != 0:
                hip.hipHostUnregister(self._host)
                raise _lib.GgmsError(f"hipHostGetDevicePointer -> {rc}")
        kind = {4: "<i4", 8: "<i8", 1: "|u1", 2: "<i2"}[self.array.dtype.itemsize]
        self.ptr = dp.value
        self.tensor = torch.as_tensor(_RawDevice(self.ptr, self.array.shape, kind), device=torch.device(device))

    def close(self):
        if self._host is not None:
            self.tensor = None
            self._hip.hipHostUnregister(self._host)
            self._host = None


class SharedShard:
    """A GGMS shard that other processes can map: an allocation of its own (hipMalloc), published with
    hipIpcGetMemHandle and opened by peers with hipIpcOpenMemHandle (cuda/dist_graph.cu:228-272)."""

    def __init__(self, shape, dtype, device):
        self.shape, self.dtype, self.device = tuple(int(x) for x in shape), dtype, torch.device(device)
        nbytes = max(1, int(np.prod(self.shape))) * torch.empty(0, dtype=dtype).element_size()
        p = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().ggms_device_alloc(C.byref(p), nbytes), "ggms_device_alloc")
        self.ptr = p.value
        self.tensor = torch.as_tensor(_RawDevice(self.ptr, self.shape, _TYPESTR[dtype]), device=self.device)
        self._imported = []

    def export_handle(self):
        buf = C.create_string_buffer(64)
        check(lib().ggms_ipc_export(C.c_void_p(self.ptr), buf), "ggms_ipc_export")
        return bytes(buf.raw)

    def import_peer(self, handle):
        """Map a peer's shard; returns its device address in this process."""
        p = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().ggms_ipc_import(C.c_char_p(handle), C.byref(p)), "ggms_ipc_import")
        self._imported.append(p.value)
        return p.value

    def release_peers(self):
        """Unmap every peer shard opened through this holder (hipIpcCloseMemHandle)."""
        for p in self._imported:
            lib().ggms_ipc_release(C.c_void_p(p))
        self._imported = []

    def close(self):
        self.release_peers()
        if self.ptr:
            self.tensor = None
            lib().ggms_device_free(C.c_void_p(self.ptr))
            self.ptr = None


class BatchSampler:
    """DoGPUSample (dist_loops.cc:62-368) as one enqueue: buffers sized once, reused every batch.

    num_slots      output slots (the engine's batch slots): a batch's COO / counts / input-node copy stay valid
                   while later batches are being sampled.
    num_pipelines  batches that may be in flight at once, each with its own dedup table and workspace.  Call
                   sample() for consecutive batches from different streams; the shared RNG pool (and khop2's CSR)
                   is still consumed in call order (ggms_sample_extra_t.rng_wait / rng_done)."""

    def __init__(self, graph, fanouts, batch_size, sample_type=KHOP3, seed=0, device="cuda", direct_table=True,
                 prob_table=None, alias_table=None, random_walk_length=0, random_walk_restart_prob=0.0,
                 num_random_walk=0, num_slots=1, num_pipelines=1):
        self.graph, self.fanouts, self.sample_type = graph, [int(f) for f in fanouts], sample_type
        L = len(self.fanouts)
        self.L = L
        self._f = (C.c_size_t * L)(*self.fanouts)
        mi, me, mu = (C.c_size_t * L)(), (C.c_size_t * L)(), C.c_size_t(0)
        # first batch of an epoch may be 1.25x (dist_shuffler_aligned.cc:137-140): size for it
        self.max_seeds = int(batch_size * 1.25) + 1
        check(lib().ggms_sample_batch_capacity(self.max_seeds, self._f, L, mi, me, C.byref(mu)), "capacity")
        self.max_input, self.max_edges, self.max_unique = list(mi), list(me), mu.value
        self.num_pipelines = num_pipelines
        self.hts = [OrderedHashTable(self.max_unique, device, num_node=graph.c.num_node if direct_table else None)
                    for _ in range(num_pipelines)]
        self.ht = self.hts[0]
        nstates = lib().ggms_random_states_count(sample_type, self._f, L, self.max_seeds, num_random_walk)
        nstates = max(nstates, (max(self.max_input) + 127) // 128 * 8, (max(self.max_input) + 1023) // 1024 * 256)
        if sample_type == RANDOM_WALK:
            nstates = max(nstates, lib().ggms_random_walk_num_states(max(self.max_input), num_random_walk))
        self.states = random_states(nstates, seed, device) if sample_type != KHOP0 else None
        self.num_slots = num_slots
        self.rows = [[torch.empty(max(1, e), dtype=torch.int32, device=device) for e in self.max_edges]
                     for _ in range(num_slots)]
        self.cols = [[torch.empty(max(1, e), dtype=torch.int32, device=device) for e in self.max_edges]
                     for _ in range(num_slots)]
        self._rows = [(C.c_void_p * L)(*[t.data_ptr() for t in r]) for r in self.rows]
        self._cols = [(C.c_void_p * L)(*[t.data_ptr() for t in c]) for c in self.cols]
        self.counts_slots = [torch.zeros(3 * L + 2, dtype=torch.int64, device=device) for _ in range(num_slots)]
        self.input_nodes = [torch.empty(self.max_unique, dtype=torch.int32, device=device) for _ in range(num_slots)]
        self.row, self.col, self.counts = self.rows[0], self.cols[0], self.counts_slots[0]
        self.data = None
        self.datas = None
        self._keep = (prob_table, alias_table)
        if sample_type == RANDOM_WALK:
            self.datas = [[torch.empty(max(1, e), dtype=torch.int32, device=device) for e in self.max_edges]
                          for _ in range(num_slots)]
            self._datas = [(C.c_void_p * L)(*[t.data_ptr() for t in d]) for d in self.datas]
            self.data = self.datas[0]
        # one extras struct per (pipeline, slot) use: filled in sample()
        self._extra = _lib.SampleExtra()
        self._extra.prob_table = prob_table.data_ptr() if prob_table is not None else None
        self._extra.alias_table = alias_table.data_ptr() if alias_table is not None else None
        self._extra.random_walk_length = random_walk_length
        self._extra.random_walk_restart_prob = random_walk_restart_prob
        self._extra.num_random_walk = num_random_walk
        if self.datas is not None:
            self._extra.data = C.cast(self._datas[0], C.c_void_p)
        wsb = lib().ggms_sample_batch_workspace_bytes(sample_type, self.max_seeds, self._f, L, C.byref(self._extra))
        self.wss = [_workspace(wsb, device) for _ in range(num_pipelines)]
        self.ws = self.wss[0]
        # batch-order events on the RNG pool; only needed when batches overlap
        self._events = []
        if num_pipelines > 1 and sample_type != KHOP0:
            for _ in range(num_pipelines):
                e = C.c_void_p()
                check(lib().ggms_event_create(C.byref(e)), "ggms_event_create")
                self._events.append(e)
        self._batch_no = 0
        self.active_pipelines = num_pipelines

    def use_pipelines(self, k):
        """Run the following batches on the first k of the allocated pipelines (1 <= k <= num_pipelines).  The caller
        has drained the device (nothing of an earlier batch is in flight): the batch order on the RNG pool starts anew."""
        assert 1 <= k <= self.num_pipelines
        self.active_pipelines = k
        self._batch_no = 0

    def __del__(self):
        try:
            for e in self._events:
                lib().ggms_event_destroy(e)
        except Exception:
            pass

    def sample(self, seeds, slot=0, copy_input_nodes=False, heavy_wait=None, distinct=False):
        """Enqueue one batch into output slot `slot`; read counts / row / col / ht.n2o after a sync.
        copy_input_nodes: also copy the unique list (ht.n2o, reused by a later batch) into the slot.
        distinct: the caller promises pairwise distinct seeds (ggms_sample_extra_t.seeds_distinct) -- a slice of a
        shuffled train set is; the seeds' insert / ordered scan / look-up launches are then skipped.
        Batch b runs on pipeline b % num_pipelines (its table is `self.ht` until the next call)."""
        _i32(seeds)
        n = seeds.numel()
        assert n <= self.max_seeds
        b, K = self._batch_no, self.active_pipelines
        self._batch_no += 1
        pipe = b % K
        self.ht = self.hts[pipe]
        counts = self.counts_slots[slot]
        self.row, self.col, self.counts = self.rows[slot], self.cols[slot], counts
        ex = self._extra
        if self.datas is not None:
            ex.data = C.cast(self._datas[slot], C.c_void_p)
            self.data = self.datas[slot]
        if self._events and K > 1:
            ex.rng_wait = self._events[(b - 1) % K] if b > 0 else None
            ex.rng_done = self._events[pipe]
        else:
            ex.rng_wait = ex.rng_done = None
        # heavy_wait (torch.cuda.Event): the last layer's sampler launch waits for it (ggms_sample_extra_t.heavy_wait)
        ex.heavy_wait = C.c_void_p(heavy_wait.cuda_event) if heavy_wait is not None else None
        ex.seeds_distinct = 1 if distinct else 0
        ws = self.wss[pipe]
        # copy_input_nodes: the slot keeps the batch's unique list.  The table's n2o buffer is the caller's
        # (ggms_hashtable_t is plain data), so the batch simply builds the list IN the slot's buffer -- no copy
        self._n2o = self.input_nodes[slot] if copy_input_nodes else self.ht.n2o
        self.ht.c.n2o = self._n2o.data_ptr()
        self.ht.c.n2o_size = self._n2o.numel()
        check(lib().ggms_sample_batch(self.sample_type, C.byref(self.graph.c), _ptr(seeds), n, self._f, self.L,
                                      C.byref(self.ht.c), _ptr(self.states),
                                      self.states.shape[0] if self.states is not None else 0, self._rows[slot],
                                      self._cols[slot], _ptr(counts), C.byref(ex), _ptr(ws), ws.numel() * 4,
                                      _stream()),
              "ggms_sample_batch")

    def result(self):
        """Sync and slice the outputs (host round trip: for tests and hand-off, not for the hot loop)."""
        c = self.counts.cpu().tolist()
        if c[3 * self.L + 1]:  # a kernel of the batch hit a bound it must not hit: the outputs are invalid
            check_device_status("ggms_sample_batch")
            raise _lib.GgmsError(f"ggms_sample_batch: device status {c[3 * self.L + 1]:#x}")
        layers = []
        for i in range(self.L):
            ne = c[3 * i]
            layers.append(dict(row=self.row[i][:ne], col=self.col[i][:ne], num_src=c[3 * i + 1], num_dst=c[3 * i + 2],
                               data=self.data[i][:ne] if self.data is not None else None))
        return dict(layers=layers, input_nodes=self._n2o[: c[3 * self.L]])
