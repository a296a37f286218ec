"""Argument checking at the C ABI, on a host without a GPU: an entry point given nonsense returns GGMS_ERR_INVALID and a
message (ggms_last_error) BEFORE it touches the device -- the reference CHECK-aborts in these places; a leaf operator
linked into somebody else's engine must not."""
import ctypes as C

import pytest

from xgnn_amd import _lib
from xgnn_amd._lib import Graph, HashTable, lib

INVALID = -1  # GGMS_ERR_INVALID, include/ggms.h


def _err():
    return lib().ggms_last_error().decode()


def test_error_codes_of_the_header():
    text = open(__file__.replace("tests/test_abi_negative.py", "include/ggms.h")).read()
    assert "GGMS_ERR_INVALID = -1" in text.replace("  ", " ")


def test_extract_family_refuses_empty_rows_and_null_pointers():
    l = lib()
    p = C.c_void_p(1 << 20)  # never dereferenced: the checks come first
    assert l.ggms_extract(p, p, p, 10, 0, 0, None) == INVALID and "invalid argument" in _err()   # dim 0
    assert l.ggms_extract(p, p, p, 10, 4, 99, None) == INVALID                                   # unknown dtype
    assert l.ggms_extract(None, p, p, 10, 4, 0, None) == INVALID                                 # no destination
    assert l.ggms_extract(p, p, p, 0, 4, 0, None) == 0                                           # nothing to do is fine
    assert l.ggms_gather_scatter(None, p, p, p, 5, None, 4, 0, None) == INVALID
    assert l.ggms_gather_scatter_partition(p, p, 0, p, p, 5, None, 4, 0, None) == INVALID        # num_part 0
    assert l.ggms_mock_extract(p, p, p, 5, 4, 0, 0, None) == INVALID                             # mock_bits out of range
    assert l.ggms_mock_extract(p, p, p, 5, 4, 0, 33, None) == INVALID


def test_owner_split_refuses_bad_partition_counts():
    l = lib()
    p = C.c_void_p(1 << 20)
    for bad in (0, 65):
        assert l.ggms_owner_histogram(p, p, 10, None, bad, p, p, None) == INVALID
        assert l.ggms_owner_bucket(p, p, 10, None, bad, p, p, p, None) == INVALID
    assert l.ggms_owner_histogram(p, p, 10, None, 8, p, None, None) == INVALID  # no counters


def test_hashtable_refuses_bad_geometry():
    l = lib()
    ht = HashTable()
    assert l.ggms_hashtable_init(C.byref(ht), None) == INVALID          # no buffers
    ht.o2n, ht.n2o, ht.num_items_dev = 1 << 20, 1 << 21, 1 << 22
    ht.o2n_size, ht.n2o_size, ht.direct = 1000, 10, 0                    # hashed layout wants a power of two
    assert l.ggms_hashtable_init(C.byref(ht), None) == INVALID
    ht.o2n_size = 0
    assert l.ggms_hashtable_init(C.byref(ht), None) == INVALID
    ht.o2n_size, ht.version = 1024, 0                                    # fill before init / reset
    assert l.ggms_hashtable_fill_with_duplicates(C.byref(ht), C.c_void_p(8), 4, None, C.c_void_p(8), 1 << 20, None) == INVALID
    assert l.ggms_map_edges(C.byref(ht), None, None, None, None, 4, None) == INVALID
    assert l.ggms_hashtable_num_buckets(0) == 4 and l.ggms_hashtable_num_buckets(8_448_000) == 1 << 25  # TableSize


def test_batch_entry_points_refuse_bad_layer_counts():
    l = lib()
    f = (C.c_size_t * 17)(*([5] * 17))
    out = (C.c_size_t * 17)()
    edges = (C.c_size_t * 17)()
    mu = C.c_size_t(0)
    assert l.ggms_sample_batch_capacity(8000, f, 0, out, out, C.byref(mu)) == INVALID
    assert l.ggms_sample_batch_capacity(8000, f, 17, out, out, C.byref(mu)) == INVALID
    assert l.ggms_sample_batch_capacity(8000, None, 2, out, out, C.byref(mu)) == INVALID
    assert l.ggms_sample_batch_capacity(8000, f, 3, out, edges, C.byref(mu)) == 0
    assert list(out)[:3] == [8000 * 6 * 6, 8000 * 6, 8000] and mu.value == 8000 * 6 * 6 * 6  # PredictNumNodes, common.cc:488-497
    assert list(edges)[:3] == [5 * n for n in list(out)[:3]]
    assert l.ggms_sample_batch_workspace_bytes(7, 8000, f, 0, None) == 0
    assert l.ggms_sample_batch_workspace_bytes(7, 8000, f, 3, None) > 0
    g, ht = Graph(), HashTable()
    p = C.c_void_p(1 << 20)
    rows = (C.c_void_p * 3)(p, p, p)
    assert l.ggms_sample_batch(7, C.byref(g), p, 10, f, 0, C.byref(ht), p, 100, rows, rows, p, None, p, 1 << 30, None) == INVALID
    assert l.ggms_sample_batch(42, C.byref(g), p, 10, f, 2, C.byref(ht), p, 100, rows, rows, p, None, p, 1 << 30, None) == INVALID
    assert l.ggms_sample_batch(7, C.byref(g), p, 10, f, 2, C.byref(ht), p, 100, rows, rows, p, None, p, 16, None) == INVALID  # workspace too small


def test_probe_and_samplers_refuse_nonsense():
    l = lib()
    p = C.c_void_p(1 << 20)
    assert l.ggms_fabric_probe(7, p, 1024, 100, 1, p, None) == INVALID        # unknown kind
    assert l.ggms_fabric_probe(0, None, 1024, 100, 1, p, None) == INVALID     # no table
    assert l.ggms_fabric_probe(0, p, 0, 100, 1, p, None) == INVALID
    g = Graph()
    g.num_part = 2  # sharded view: the weighted / khop2 samplers refuse it, as dist_loops.cc:167-228
    nout = C.c_void_p(1 << 21)
    assert l.ggms_sample_khop2(C.byref(g), p, 10, 5, p, p, nout, p, 1000, p, 1 << 20, None) == INVALID
    assert l.ggms_sample_weighted_khop(C.byref(g), p, p, p, 10, 5, p, p, nout, p, 1 << 20, p, 1 << 20, None) == INVALID
    g.num_part = 0
    assert l.ggms_sample_weighted_khop_hash_dedup(C.byref(g), p, p, p, 10, 50, p, p, nout, p, 1 << 20, p, 1 << 20, None) == INVALID  # fanout < 50
    assert l.ggms_sample_khop3(C.byref(g), p, 10, 128, p, p, nout, p, 1 << 20, p, 1 << 30, None) == INVALID                         # fanout < 128
