"""ctypes binding of include/ggms.h.  Fails loudly when the library is absent."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libggms_hip.so")
_LIB = None


class GgmsError(RuntimeError):
    pass


class Graph(C.Structure):
    """ggms_graph_t"""
    _fields_ = [("indptr", C.c_void_p), ("indices", C.c_void_p),
                ("part_indptr", C.c_void_p), ("part_indices", C.c_void_p),
                ("num_part", C.c_uint32), ("num_cache_node", C.c_uint32),
                ("num_node", C.c_uint32), ("_pad", C.c_uint32)]


class SampleExtra(C.Structure):
    """ggms_sample_extra_t"""
    _fields_ = [("prob_table", C.c_void_p), ("alias_table", C.c_void_p), ("random_walk_length", C.c_size_t),
                ("random_walk_restart_prob", C.c_double), ("num_random_walk", C.c_size_t), ("data", C.c_void_p),
                ("rng_wait", C.c_void_p), ("rng_done", C.c_void_p), ("heavy_wait", C.c_void_p),
                ("seeds_distinct", C.c_uint32), ("_pad", C.c_uint32)]


class FeatureTiers(C.Structure):
    """ggms_feature_tiers_t"""
    _fields_ = [("table", C.c_void_p), ("replica", C.c_void_p), ("num_replica", C.c_uint64),
                ("parts", C.c_void_p), ("num_part", C.c_uint32), ("my_part", C.c_uint32),
                ("host_feat", C.c_void_p), ("host_row_mask", C.c_uint32), ("_pad", C.c_uint32)]


class Topology(C.Structure):
    """ggms_topology_t"""
    _fields_ = [("num_device", C.c_int32), ("_pad", C.c_int32), ("can_access", (C.c_int32 * 16) * 16),
                ("copy_GBps", (C.c_double * 16) * 16)]


class HashTable(C.Structure):
    """ggms_hashtable_t"""
    _fields_ = [("o2n", C.c_void_p), ("n2o", C.c_void_p), ("num_items_dev", C.c_void_p),
                ("o2n_size", C.c_uint64), ("n2o_size", C.c_uint64),
                ("version", C.c_uint32), ("direct", C.c_uint32)]


# name -> (restype, argtypes); every symbol declared in include/ggms.h
_vp, _sz, _u64, _u32, _i = C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_int
SYMBOLS = {
    "ggms_abi_version": (_i, []),
    "ggms_last_error": (C.c_char_p, []),
    "ggms_device_status": (_i, [C.POINTER(_u32), _i]),
    "ggms_debug_poison_next_scan": (None, []),
    "ggms_debug_set_scan_patience": (None, [C.c_uint32]),
    "ggms_debug_delay_next_scan": (None, [C.c_uint32]),
    "ggms_debug_set_knob": (None, [_i, C.c_longlong]),
    "ggms_fabric_probe": (_i, [_i, _vp, _sz, _sz, _u32, _vp, _vp]),
    "ggms_dtype_bytes": (_sz, [_i]),
    "ggms_random_states_init": (_i, [_vp, _sz, _u64, _vp]),
    "ggms_random_states_count": (_sz, [_i, C.POINTER(_sz), _sz, _sz, _sz]),
    "ggms_sample_workspace_bytes": (_sz, [_i, _sz, _sz]),
    "ggms_sample_khop3": (_i, [C.POINTER(Graph), _vp, _sz, _sz, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "ggms_sample_khop0": (_i, [C.POINTER(Graph), _vp, _sz, _sz, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ggms_sample_khop1": (_i, [C.POINTER(Graph), _vp, _sz, _sz, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "ggms_sample_khop2": (_i, [C.POINTER(Graph), _vp, _sz, _sz, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "ggms_sample_weighted_workspace_bytes": (_sz, [_sz, _sz]),
    "ggms_sample_weighted_khop": (_i, [C.POINTER(Graph), _vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _sz, _vp, _sz,
                                       _vp]),
    "ggms_sample_weighted_khop_prefix": (_i, [C.POINTER(Graph), _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _sz, _vp, _sz,
                                              _vp]),
    "ggms_sample_weighted_khop_hash_dedup": (_i, [C.POINTER(Graph), _vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _sz,
                                                  _vp, _sz, _vp]),
    "ggms_sample_random_walk_workspace_bytes": (_sz, [_sz, _sz, _sz, _sz]),
    "ggms_random_walk_num_states": (_sz, [_sz, _sz]),
    "ggms_sample_random_walk": (_i, [C.POINTER(Graph), _vp, _sz, _sz, C.c_double, _sz, _sz, _vp, _vp, _vp, _vp, _vp,
                                     _sz, _vp, _sz, _vp]),
    "ggms_device_alloc": (_i, [C.POINTER(C.c_void_p), _sz]),
    "ggms_device_free": (_i, [_vp]),
    "ggms_ipc_export": (_i, [_vp, _vp]),
    "ggms_ipc_import": (_i, [_vp, C.POINTER(C.c_void_p)]),
    "ggms_ipc_release": (_i, [_vp]),
    "ggms_owner_histogram": (_i, [_vp, _vp, _sz, _vp, _u32, _vp, _vp, _vp]),
    "ggms_owner_bucket": (_i, [_vp, _vp, _sz, _vp, _u32, _vp, _vp, _vp, _vp]),
    "ggms_event_create": (_i, [C.POINTER(C.c_void_p)]),
    "ggms_event_destroy": (_i, [_vp]),
    "ggms_sample_batch_capacity": (_i, [_sz, C.POINTER(_sz), _u32, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz)]),
    "ggms_sample_batch_workspace_bytes": (_sz, [_i, _sz, C.POINTER(_sz), _u32, C.POINTER(SampleExtra)]),
    "ggms_sample_batch": (_i, [_i, C.POINTER(Graph), _vp, _sz, C.POINTER(_sz), _u32, C.POINTER(HashTable), _vp, _sz,
                               C.POINTER(_vp), C.POINTER(_vp), _vp, C.POINTER(SampleExtra), _vp, _sz, _vp]),
    "ggms_hashtable_num_buckets": (_sz, [_sz]),
    "ggms_hashtable_init": (_i, [C.POINTER(HashTable), _vp]),
    "ggms_hashtable_reset": (_i, [C.POINTER(HashTable), _vp]),
    "ggms_hashtable_workspace_bytes": (_sz, [_sz]),
    "ggms_hashtable_fill_with_duplicates": (_i, [C.POINTER(HashTable), _vp, _sz, _vp, _vp, _sz, _vp]),
    "ggms_map_edges": (_i, [C.POINTER(HashTable), _vp, _vp, _vp, _vp, _sz, _vp]),
    "ggms_extract": (_i, [_vp, _vp, _vp, _sz, _sz, _i, _vp]),
    "ggms_count_nodes": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "ggms_cache_index_workspace_bytes": (_sz, [_sz]),
    "ggms_get_miss_cache_index": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ggms_get_miss_cache_index_dev": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ggms_gather_scatter": (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _sz, _i, _vp]),
    "ggms_gather_scatter_partition": (_i, [_vp, _vp, _u32, _vp, _vp, _sz, _vp, _sz, _i, _vp]),
    "ggms_extract_cached": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _u32, _vp, _sz, _i, _vp, _vp]),
    "ggms_build_alias_table_host": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _i]),
    "ggms_build_prob_prefix_table_host": (_i, [_vp, _sz, _vp, _vp, _i]),
    "ggms_mock_extract": (_i, [_vp, _vp, _vp, _sz, _sz, _i, _u32, _vp]),
    "ggms_gather_scatter_masked": (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _sz, _i, _u32, _vp]),
    "ggms_ipc_safe_bytes": (_sz, [_sz]),
    "ggms_extract_tiered": (_i, [_vp, _vp, _sz, _vp, C.POINTER(FeatureTiers), _sz, _i, _vp, _vp]),
    "ggms_device_count": (_i, [C.POINTER(_i)]),
    "ggms_peer_access": (_i, [_i, _i, C.POINTER(_i)]),
    "ggms_detect_topology": (_i, [C.POINTER(Topology), _sz, _i]),
    "ggms_topology_write_host": (_i, [C.POINTER(Topology), C.c_char_p, C.c_char_p]),
    "ggms_topology_read_host": (_i, [C.POINTER(Topology), C.c_char_p]),
    "ggms_link_probe_copy": (_i, [_vp, _vp, _sz, _i, _i, C.POINTER(C.c_double), _vp]),
    "ggms_link_probe_gather": (_i, [_vp, _vp, _u32, _sz, _sz, _sz, _u32, _i, _vp, C.POINTER(C.c_double), _vp]),
    "ggms_launch_timer_create": (_i, [C.POINTER(C.c_void_p)]),
    "ggms_launch_timer_destroy": (_i, [_vp]),
    "ggms_launch_timer_arm": (_i, [_vp]),
    "ggms_launch_timer_wait": (_i, [_vp, _vp]),
    "ggms_launch_timer_elapsed_us": (_i, [_vp, C.POINTER(C.c_double)]),
    "ggms_launch_timer_span_us": (_i, [_vp, _vp, C.POINTER(C.c_double)]),
}

ABI_VERSION = 3  # include/ggms.h as this binding declares it (struct layouts, host / device pointer conventions)


def lib():
    """Load libggms_hip.so (once).  Raises GgmsError if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise GgmsError(
                f"{LIB_PATH} is missing: build it with `make -C xgnn_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        h = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(h, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        # a stale build (another struct layout, pointer tables on the other side of the bus) must not be called at all
        if h.ggms_abi_version() != ABI_VERSION:
            raise GgmsError(f"{LIB_PATH} has ABI version {h.ggms_abi_version()}, this binding is written for "
                            f"{ABI_VERSION}: rebuild it (`make -C xgnn_amd/csrc`)")
        _LIB = h
    return _LIB


def check(rc, what=""):
    if rc != 0:
        msg = lib().ggms_last_error().decode(errors="replace")
        raise GgmsError(f"{what} failed (status {rc}): {msg}")
