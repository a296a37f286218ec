"""TEST INFRASTRUCTURE ONLY.

ctypes/numpy front-end of ``oracle/libggms_oracle.so`` (our plain-C restatement
of the reference's GGMS hot path) and, when present, of
``oracle/_ref/libref_cpu.so`` (the reference's own CPU leaf objects).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product package ``xgnn_amd`` never
does; it fails loudly when its HIP library is missing.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

EMPTY_KEY = 0xFFFFFFFF
KHOP0, KHOP1, WEIGHTED_KHOP, RANDOM_WALK, KHOP2, KHOP3, CPU_KHOP0 = 0, 1, 2, 3, 5, 7, 100
WEIGHTED_KHOP_PREFIX, WEIGHTED_KHOP_HASH_DEDUP = 4, 6

u32p = C.POINTER(C.c_uint32)
XORWOW_DTYPE = np.dtype([("d", "<u4"), ("v", "<u4", (5,))])


def build(ref=True):
    """(Re)build the oracle; building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", _HERE])
    if ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libggms_oracle.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(
            os.path.join(_HERE, "ggms_oracle.c")
        ):
            build(ref=False)
        _LIB = C.CDLL(path)
        _LIB.orc_predict_num_nodes.restype = C.c_size_t
        _LIB.orc_xorwow_next.restype = C.c_uint32
        _LIB.orc_xorwow_uniform.restype = C.c_float
        _LIB.orc_xorwow_fold.restype = C.c_uint32
        _LIB.orc_xorwow_uniform_double.restype = C.c_double
        _LIB.orc_mt19937_next.restype = C.c_uint32
        _LIB.orc_mt19937_uniform_u32.restype = C.c_uint32
        _LIB.orc_minstd0_next.restype = C.c_uint64
        _LIB.orc_minstd0_uniform_u64.restype = C.c_uint64
        _LIB.orc_cpu_random_id.restype = C.c_uint32
        _LIB.orc_aligned_pad.restype = C.c_size_t
        _LIB.orc_ht_create.restype = C.c_void_p
        _LIB.orc_ht_fill_with_duplicates.restype = C.c_size_t
        _LIB.orc_ht_num_items.restype = C.c_size_t
        _LIB.orc_ht_unique.restype = u32p
        _LIB.orc_do_sample.restype = C.c_void_p
        _LIB.orc_do_sample_ex.restype = C.c_void_p
        _LIB.orc_partition_feature.restype = C.c_size_t
        _LIB.orc_num_cache_node.restype = C.c_uint32
    return _LIB


def ref_lib():
    """The reference's own CPU leaves (oracle/_ref), or None if not built."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libref_cpu.so")
        if not os.path.exists(path):
            return None
        _REF = C.CDLL(path)
        _REF.ref_random_id.restype = C.c_uint32
        _REF.ref_dtype_code.restype = C.c_int
    return _REF


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _sz(x):
    return C.c_size_t(int(x))


# --------------------------------------------------------------------- RNG
def random_states(num, seed):
    st = np.zeros(int(num), dtype=XORWOW_DTYPE)
    lib().orc_random_states_init(_p(st), _sz(num), C.c_uint64(int(seed)))
    return st


def xorwow_stream(seed, n):
    st = np.zeros(1, dtype=XORWOW_DTYPE)
    lib().orc_xorwow_init(_p(st), C.c_uint64(int(seed)))
    return np.array([lib().orc_xorwow_next(_p(st)) for _ in range(n)], dtype=np.uint32)


def xorwow_draws(state, n):
    """n curand() draws from a 1-element XORWOW_DTYPE array (advanced in place)."""
    out = np.empty(int(n), np.uint32)
    lib().orc_xorwow_draws(_p(state), _sz(n), _p(out))
    return out


def xorwow_uniforms(state, n, double=False):
    """n curand_uniform() / curand_uniform_double() values (state advanced in place)."""
    out = np.empty(int(n), np.float64 if double else np.float32)
    (lib().orc_xorwow_uniform_doubles if double else lib().orc_xorwow_uniforms)(_p(state), _sz(n), _p(out))
    return out


def rocrand_pin():
    """librocrand_pin.so: rocRAND's host XORWOW engine (third-party pin of the recurrence), or None."""
    path = os.path.join(_HERE, "librocrand_pin.so")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "rocrand_pin.cc")):
        if subprocess.call(["make", "-s", "-C", _HERE, "pin"]) != 0 or not os.path.exists(path):
            return None
    p = C.CDLL(path)
    p.pin_xorwow_fold.restype = C.c_uint32
    return p


def xorwow_next(state):
    """Advance a 1-element XORWOW_DTYPE array by one curand() draw; returns the draw."""
    return int(lib().orc_xorwow_next(_p(state)))


def shuffle_minstd0(data, seed):
    d = _u32(data).copy()
    lib().orc_shuffle_minstd0(_p(d), _sz(d.size), C.c_uint64(int(seed)))
    return d


def aligned_pad(train, num_worker):
    """DistAlignedShuffler ctor padding (dist_shuffler_aligned.cc:46-56)."""
    train = _u32(train)
    n = int(lib().orc_aligned_pad(None, _sz(train.size), _sz(num_worker), None))
    out = np.empty(n, np.uint32)
    lib().orc_aligned_pad(_p(train), _sz(train.size), _sz(num_worker), _p(out))
    return out


def mt19937_shuffle(data, seed):
    g = (C.c_uint32 * 625)()
    lib().orc_mt19937_seed(g, C.c_uint32(int(seed)))
    d = _u32(data).copy()
    lib().orc_mt19937_shuffle_u32(g, _p(d), _sz(d.size))
    return d


def predict_num_nodes(batch, fanouts, k):
    f = (C.c_size_t * len(fanouts))(*[int(x) for x in fanouts])
    return int(lib().orc_predict_num_nodes(_sz(batch), f, _sz(k)))


# ---------------------------------------------------------------- samplers
def _alloc_out(n, fanout):
    m = max(1, int(n) * int(fanout))
    return np.empty(m, np.uint32), np.empty(m, np.uint32)


def sample_khop3(indptr, indices, inp, fanout, states):
    indptr, indices, inp = _u32(indptr), _u32(indices), _u32(inp)
    src, dst = _alloc_out(inp.size, fanout)
    n = C.c_size_t(0)
    lib().orc_sample_khop3(_p(indptr), _p(indices), _p(inp), _sz(inp.size), _sz(fanout),
                           _p(states), _sz(states.size), _p(src), _p(dst), C.byref(n))
    return src[: n.value].copy(), dst[: n.value].copy()


def sample_weighted_khop_prefix(indptr, indices, prob_prefix, inp, fanout, states):
    indptr, indices, inp = _u32(indptr), _u32(indices), _u32(inp)
    prob_prefix = np.ascontiguousarray(prob_prefix, np.float32)
    src, dst = _alloc_out(inp.size, fanout)
    n = C.c_size_t(0)
    lib().orc_sample_weighted_khop_prefix(_p(indptr), _p(indices), _p(prob_prefix), _p(inp), _sz(inp.size),
                                          _sz(fanout), _p(states), _sz(states.size), _p(src), _p(dst), C.byref(n))
    return src[: n.value].copy(), dst[: n.value].copy()


def sample_weighted_khop_hash_dedup(indptr, indices, prob, alias, inp, fanout, states):
    indptr, indices, inp, alias = _u32(indptr), _u32(indices), _u32(inp), _u32(alias)
    prob = np.ascontiguousarray(prob, np.float32)
    src, dst = _alloc_out(inp.size, fanout)
    n = C.c_size_t(0)
    lib().orc_sample_weighted_khop_hash_dedup(_p(indptr), _p(indices), _p(prob), _p(alias), _p(inp), _sz(inp.size),
                                              _sz(fanout), _p(states), _sz(states.size), _p(src), _p(dst), C.byref(n))
    return src[: n.value].copy(), dst[: n.value].copy()


def sample_khop1(indptr, indices, inp, fanout, states):
    indptr, indices, inp = _u32(indptr), _u32(indices), _u32(inp)
    src, dst = _alloc_out(inp.size, fanout)
    n = C.c_size_t(0)
    lib().orc_sample_khop1(_p(indptr), _p(indices), _p(inp), _sz(inp.size), _sz(fanout),
                           _p(states), _sz(states.size), _p(src), _p(dst), C.byref(n))
    return src[: n.value].copy(), dst[: n.value].copy()


def sample_khop2(indptr, indices, inp, fanout, states):
    """`indices` must be a writable contiguous uint32 array: it is permuted in place, like the reference's."""
    assert indices.dtype == np.uint32 and indices.flags.c_contiguous and indices.flags.writeable
    indptr, inp = _u32(indptr), _u32(inp)
    src, dst = _alloc_out(inp.size, fanout)
    n = C.c_size_t(0)
    lib().orc_sample_khop2(_p(indptr), _p(indices), _p(inp), _sz(inp.size), _sz(fanout),
                           _p(states), _sz(states.size), _p(src), _p(dst), C.byref(n))
    return src[: n.value].copy(), dst[: n.value].copy()


def sample_khop0(indptr, indices, inp, fanout):
    indptr, indices, inp = _u32(indptr), _u32(indices), _u32(inp)
    src, dst = _alloc_out(inp.size, fanout)
    n = C.c_size_t(0)
    lib().orc_sample_khop0(_p(indptr), _p(indices), _p(inp), _sz(inp.size), _sz(fanout),
                           _p(src), _p(dst), C.byref(n))
    return src[: n.value].copy(), dst[: n.value].copy()


def sample_weighted_khop(indptr, indices, prob, alias, inp, fanout, states):
    indptr, indices, inp, alias = _u32(indptr), _u32(indices), _u32(inp), _u32(alias)
    prob = np.ascontiguousarray(prob, np.float32)
    src, dst = _alloc_out(inp.size, fanout)
    n = C.c_size_t(0)
    lib().orc_sample_weighted_khop(_p(indptr), _p(indices), _p(prob), _p(alias), _p(inp),
                                   _sz(inp.size), _sz(fanout), _p(states), _sz(states.size),
                                   _p(src), _p(dst), C.byref(n))
    return src[: n.value].copy(), dst[: n.value].copy()


def random_walk_raw(indptr, indices, inp, walk_length, restart_prob, num_walk, states):
    indptr, indices, inp = _u32(indptr), _u32(indices), _u32(inp)
    m = max(1, inp.size * walk_length * num_walk)
    src = np.full(m, EMPTY_KEY, np.uint32)
    dst = np.full(m, EMPTY_KEY, np.uint32)
    lib().orc_random_walk_raw(_p(indptr), _p(indices), _p(inp), _sz(inp.size), _sz(walk_length),
                              C.c_double(restart_prob), _sz(num_walk), _p(states),
                              _sz(states.size), _p(src), _p(dst))
    return src, dst


def sample_random_walk(indptr, indices, inp, walk_length, restart_prob, num_walk, K, states):
    indptr, indices, inp = _u32(indptr), _u32(indices), _u32(inp)
    m = max(1, inp.size * K)
    src, dst, data = (np.empty(m, np.uint32) for _ in range(3))
    n = C.c_size_t(0)
    lib().orc_sample_random_walk(_p(indptr), _p(indices), _p(inp), _sz(inp.size),
                                 _sz(walk_length), C.c_double(restart_prob), _sz(num_walk),
                                 _sz(K), _p(states), _sz(states.size), _p(src), _p(dst),
                                 _p(data), C.byref(n))
    return src[: n.value].copy(), dst[: n.value].copy(), data[: n.value].copy()


def cpu_random_reset():
    lib().orc_cpu_random_reset()


def cpu_sample_khop0(indptr, indices, inp, fanout, num_threads=1):
    indptr, indices, inp = _u32(indptr), _u32(indices), _u32(inp)
    src, dst = _alloc_out(inp.size, fanout)
    n = C.c_size_t(0)
    lib().orc_cpu_sample_khop0(_p(indptr), _p(indices), _p(inp), _sz(inp.size), _p(src), _p(dst),
                               C.byref(n), _sz(fanout), C.c_int(num_threads))
    return src[: n.value].copy(), dst[: n.value].copy()


def cpu_sample_khop2(indptr, indices, inp, fanout, num_threads=1):
    """CPUSampleKHop2 restated; `indices` (writable uint32) is permuted in place."""
    assert indices.dtype == np.uint32 and indices.flags.c_contiguous and indices.flags.writeable
    indptr, inp = _u32(indptr), _u32(inp)
    src, dst = _alloc_out(inp.size, fanout)
    n = C.c_size_t(0)
    lib().orc_cpu_sample_khop2(_p(indptr), _p(indices), _p(inp), _sz(inp.size), _p(src), _p(dst),
                               C.byref(n), _sz(fanout), C.c_int(num_threads))
    return src[: n.value].copy(), dst[: n.value].copy()


def extract(src, index, num_threads=1):
    """out[i, :] = src[index[i], :] for a 2-D (or 1-D) array of any dtype."""
    src = np.ascontiguousarray(src)
    index = _u32(index)
    row_bytes = src.strides[0] if src.ndim > 1 else src.itemsize
    out = np.empty((index.size,) + src.shape[1:], dtype=src.dtype)
    lib().orc_extract(_p(out), _p(src), _p(index), _sz(index.size), _sz(row_bytes),
                      C.c_int(num_threads))
    return out


# -------------------------------------------------------------- hash table
class HashTable:
    def __init__(self, num_node, capacity):
        self._h = C.c_void_p(lib().orc_ht_create(_sz(num_node), _sz(capacity)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_ht_destroy(self._h)
            self._h = None

    def reset(self):
        lib().orc_ht_reset(self._h)

    def fill_with_duplicates(self, inp):
        inp = _u32(inp)
        return int(lib().orc_ht_fill_with_duplicates(self._h, _p(inp), _sz(inp.size)))

    @property
    def num_items(self):
        return int(lib().orc_ht_num_items(self._h))

    def unique(self):
        n = self.num_items
        ptr = lib().orc_ht_unique(self._h)
        return np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].copy()

    def map_edges(self, src, dst):
        src, dst = _u32(src), _u32(dst)
        ns, nd = np.empty(max(1, src.size), np.uint32), np.empty(max(1, src.size), np.uint32)
        lib().orc_ht_map_edges(self._h, _p(src), _p(dst), _sz(src.size), _p(ns), _p(nd))
        return ns[: src.size], nd[: src.size]


def create_alias_table(indptr, indices, weights):
    """create_alias_table.cc:105-170 -> (prob_table f32[E], alias_table u32[E])."""
    ip, ix, w = _u32(indptr), _u32(indices), np.ascontiguousarray(weights, np.float32)
    prob, alias = np.empty(ix.size, np.float32), np.empty(ix.size, np.uint32)
    lib().orc_create_alias_table(_p(ip), _p(ix), _sz(ip.size - 1), _p(w), _p(prob), _p(alias))
    return prob, alias


def create_prob_prefix_table(indptr, weights):
    """create_prob_prefix_table.cc:94-123."""
    ip, w = _u32(indptr), np.ascontiguousarray(weights, np.float32)
    out = np.empty(w.size, np.float32)
    lib().orc_create_prob_prefix_table(_p(ip), _sz(ip.size - 1), _p(w), _p(out))
    return out


class CpuHashTable2:
    """CPUHashTable2 with its OpenMP loops (cpu/cpu_hashtable2.cc:35-191); threads=1 == HashTable."""

    def __init__(self, num_node, threads=1):
        self.threads = int(threads)
        lib().orc_cpu_ht2_create.restype = C.c_void_p
        lib().orc_cpu_ht2_populate.restype = C.c_size_t
        lib().orc_cpu_ht2_unique.restype = u32p
        self._h = C.c_void_p(lib().orc_cpu_ht2_create(_sz(num_node), C.c_int(self.threads)))
        self.num_items = 0

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_cpu_ht2_destroy(self._h)
            self._h = None

    def reset(self):
        lib().orc_cpu_ht2_reset(self._h, C.c_int(self.threads))
        self.num_items = 0

    def fill_with_duplicates(self, inp):
        inp = _u32(inp)
        self.num_items = int(lib().orc_cpu_ht2_populate(self._h, _p(inp), _sz(inp.size), C.c_int(self.threads)))
        return self.num_items

    def unique(self):
        ptr = lib().orc_cpu_ht2_unique(self._h)
        return np.ctypeslib.as_array(ptr, shape=(max(self.num_items, 1),))[: self.num_items].copy()

    def map_edges(self, src, dst):
        src, dst = _u32(src), _u32(dst)
        ns, nd = np.empty(max(1, src.size), np.uint32), np.empty(max(1, src.size), np.uint32)
        lib().orc_cpu_ht2_map_edges(self._h, _p(src), _p(dst), _sz(src.size), _p(ns), _p(nd), C.c_int(self.threads))
        return ns[: src.size], nd[: src.size]


class _SampleResult(C.Structure):
    _fields_ = [("num_layer", C.c_size_t), ("num_src", C.POINTER(C.c_size_t)),
                ("num_dst", C.POINTER(C.c_size_t)), ("num_edge", C.POINTER(C.c_size_t)),
                ("row", C.POINTER(u32p)), ("col", C.POINTER(u32p)), ("data", C.POINTER(u32p)),
                ("input_nodes", u32p), ("num_input_nodes", C.c_size_t)]


class _SampleExtra(C.Structure):
    _fields_ = [("prob_table", C.c_void_p), ("alias_table", C.c_void_p), ("walk_length", C.c_size_t),
                ("restart_prob", C.c_double), ("num_walk", C.c_size_t)]


def do_sample(sample_type, indptr, indices, seeds, fanouts, states=None, prob=None, alias=None, walk_length=0,
              restart_prob=0.0, num_walk=0):
    """Multi-layer loop.  Returns dict(layers=[{row,col,data,num_src,num_dst}], input_nodes)."""
    indptr, indices, seeds = _u32(indptr), _u32(indices), _u32(seeds)
    f = (C.c_size_t * len(fanouts))(*[int(x) for x in fanouts])
    if states is None:
        states = np.zeros(1, XORWOW_DTYPE)
    ex = _SampleExtra()
    if prob is not None:  # alias method (prob + alias) or prefix sums (prob only)
        prob = np.ascontiguousarray(prob, np.float32)
        ex.prob_table = prob.ctypes.data
        if alias is not None:
            alias = _u32(alias)
            ex.alias_table = alias.ctypes.data
    if sample_type == RANDOM_WALK and (walk_length < 1 or num_walk < 1):
        raise ValueError("random walk needs walk_length >= 1 and num_walk >= 1 (the block-shape rule never ends on 0)")
    ex.walk_length, ex.restart_prob, ex.num_walk = walk_length, restart_prob, num_walk
    h = lib().orc_do_sample_ex(C.c_int(sample_type), _p(indptr), _p(indices), _sz(indptr.size - 1),
                               _p(seeds), _sz(seeds.size), f, _sz(len(fanouts)), _p(states),
                               _sz(states.size), C.byref(ex))
    r = C.cast(h, C.POINTER(_SampleResult)).contents
    layers = []
    for i in range(len(fanouts)):
        ne = r.num_edge[i]
        row = np.ctypeslib.as_array(r.row[i], shape=(max(ne, 1),))[:ne].copy()
        col = np.ctypeslib.as_array(r.col[i], shape=(max(ne, 1),))[:ne].copy()
        data = None
        if r.data[i]:
            data = np.ctypeslib.as_array(r.data[i], shape=(max(ne, 1),))[:ne].copy()
        layers.append(dict(row=row, col=col, data=data, num_src=int(r.num_src[i]), num_dst=int(r.num_dst[i])))
    n = r.num_input_nodes
    inp = np.ctypeslib.as_array(r.input_nodes, shape=(max(n, 1),))[:n].copy()
    lib().orc_sample_result_free(C.c_void_p(h))
    return dict(layers=layers, input_nodes=inp)


# ------------------------------------------------------------ feature cache
def cache_build(rank_nodes, num_cached, partition_shuffle):
    rank_nodes = _u32(rank_nodes)
    rank_out = np.empty_like(rank_nodes)
    table = np.empty_like(rank_nodes)
    lib().orc_cache_build(_p(rank_nodes), _sz(rank_nodes.size), _sz(num_cached),
                          C.c_int(int(partition_shuffle)), _p(rank_out), _p(table))
    return rank_out, table


def get_miss_cache_index(table, nodes):
    table, nodes = _u32(table), _u32(nodes)
    n = max(1, nodes.size)
    ms, md, hs, hd = (np.empty(n, np.uint32) for _ in range(4))
    nm, nh = C.c_size_t(0), C.c_size_t(0)
    lib().orc_get_miss_cache_index(_p(table), _p(nodes), _sz(nodes.size), _p(ms), _p(md),
                                   C.byref(nm), _p(hs), _p(hd), C.byref(nh))
    return ms[: nm.value].copy(), md[: nm.value].copy(), hs[: nh.value].copy(), hd[: nh.value].copy()


def gather_scatter(out, src, src_index, dst_index):
    src = np.ascontiguousarray(src)
    row_bytes = src.strides[0] if src.ndim > 1 else src.itemsize
    si = _u32(src_index) if src_index is not None else None
    di = _u32(dst_index) if dst_index is not None else None
    n = si.size if si is not None else di.size
    lib().orc_gather_scatter(_p(out), _p(src), _p(si) if si is not None else None,
                             _p(di) if di is not None else None, _sz(n), _sz(row_bytes))
    return out


def gather_scatter_partition(out, parts, src_index, dst_index):
    parts = [np.ascontiguousarray(p) for p in parts]
    row_bytes = out.strides[0] if out.ndim > 1 else out.itemsize
    arr = (C.c_void_p * len(parts))(*[p.ctypes.data for p in parts])
    si, di = _u32(src_index), _u32(dst_index)
    lib().orc_gather_scatter_partition(_p(out), arr, _sz(len(parts)), _p(si), _p(di), _sz(si.size),
                                       _sz(row_bytes))
    return out


# ------------------------------------------------------------- GGMS shards
def partition_graph(indptr, indices, part_id, num_part, num_part_node):
    indptr, indices = _u32(indptr), _u32(indices)
    a, b = C.c_size_t(0), C.c_size_t(0)
    lib().orc_partition_graph(_p(indptr), _p(indices), C.c_uint32(part_id), C.c_uint32(num_part),
                              C.c_uint32(num_part_node), None, None, C.byref(a), C.byref(b))
    pi = np.empty(a.value, np.uint32)
    px = np.empty(max(1, b.value), np.uint32)
    lib().orc_partition_graph(_p(indptr), _p(indices), C.c_uint32(part_id), C.c_uint32(num_part),
                              C.c_uint32(num_part_node), _p(pi), _p(px), C.byref(a), C.byref(b))
    return pi, px[: b.value]


def partition_feature(feat, rank_nodes, num_cache, part_id, num_part):
    feat = np.ascontiguousarray(feat)
    rank_nodes = _u32(rank_nodes)
    row_bytes = feat.strides[0]
    cnt = lib().orc_partition_feature(_p(feat), _sz(row_bytes), _p(rank_nodes), C.c_uint32(num_cache),
                                      C.c_uint32(part_id), C.c_uint32(num_part), None)
    out = np.empty((cnt,) + feat.shape[1:], feat.dtype)
    lib().orc_partition_feature(_p(feat), _sz(row_bytes), _p(rank_nodes), C.c_uint32(num_cache),
                                C.c_uint32(part_id), C.c_uint32(num_part), _p(out))
    return out


def num_cache_node(indptr, percentage):
    indptr = _u32(indptr)
    return int(lib().orc_num_cache_node(_p(indptr), C.c_uint32(indptr.size - 1), C.c_double(percentage)))


# ---------------------------------------------- reference CPU leaves (_ref)
def ref_cpu_sample_khop0(indptr, indices, inp, fanout, num_threads=1):
    r = ref_lib()
    indptr, indices, inp = _u32(indptr), _u32(indices), _u32(inp)
    src, dst = _alloc_out(inp.size, fanout)
    n = C.c_size_t(0)
    r.ref_set_omp_threads(C.c_int(num_threads))
    r.ref_cpu_sample_khop0(_p(indptr), _p(indices), _p(inp), _sz(inp.size), _p(src), _p(dst),
                           C.byref(n), _sz(fanout))
    return src[: n.value].copy(), dst[: n.value].copy()


def ref_cpu_sample_khop2(indptr, indices, inp, fanout, num_threads=1):
    """The reference's CPUSampleKHop2 object; `indices` is permuted in place."""
    assert indices.dtype == np.uint32 and indices.flags.c_contiguous and indices.flags.writeable
    r = ref_lib()
    indptr, inp = _u32(indptr), _u32(inp)
    src, dst = _alloc_out(inp.size, fanout)
    n = C.c_size_t(0)
    r.ref_set_omp_threads(C.c_int(num_threads))
    r.ref_cpu_sample_khop2(_p(indptr), _p(indices), _p(inp), _sz(inp.size), _p(src), _p(dst),
                           C.byref(n), _sz(fanout))
    return src[: n.value].copy(), dst[: n.value].copy()


_REF_DT = {"float32": b"f32", "float64": b"f64", "int16": b"f16", "float16": b"f16",
           "uint8": b"u8", "int32": b"i32", "int64": b"i64"}


def ref_cpu_extract(src, index, num_threads=1):
    r = ref_lib()
    src = np.ascontiguousarray(src)
    index = _u32(index)
    dim = src.shape[1] if src.ndim > 1 else 1
    out = np.empty((index.size,) + src.shape[1:], dtype=src.dtype)
    code = r.ref_dtype_code(_REF_DT[src.dtype.name])
    r.ref_set_omp_threads(C.c_int(num_threads))
    r.ref_cpu_extract(_p(out), _p(src), _p(index), _sz(index.size), _sz(dim), C.c_int(code))
    return out
