"""C-ABI library loads on a CPU-only host and exports every symbol include/*.h declares."""
import ctypes
import os
import re

import xgnn_amd
from xgnn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:ggms|samgraph)_[a-z0-9_]+)\s*\(", text)))


def test_ggms_header_symbols_exported_and_bound():
    names = _declared("ggms.h")
    assert len(names) >= 20
    h = ctypes.CDLL(xgnn_amd.LIB_PATH)
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/ggms.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes binding in xgnn_amd/_lib.py"
    for n in _lib.SYMBOLS:
        assert n in names, f"{n} bound but not declared in include/ggms.h"


def test_samgraph_header_symbols_exported_and_bound():
    """Every samgraph_* entry of include/samgraph.h (operation.h:30-115 + adapter.cc hand-off) is exported
    by the same shared object and has a ctypes binding in the samgraph.common mirror."""
    from xgnn_amd import common
    names = _declared("samgraph.h")
    assert len(names) >= 45
    h = ctypes.CDLL(xgnn_amd.LIB_PATH)
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/samgraph.h but not exported"
        assert n in common.SAMGRAPH_SYMBOLS, f"{n} has no binding in xgnn_amd/common.py"
    for n in common.SAMGRAPH_SYMBOLS:
        assert n in names, f"{n} bound but not declared in include/samgraph.h"


def test_reference_python_surface_present():
    """Names the reference's example scripts use through `import samgraph.torch as sam`
    (samgraph/torch/adapter.py:63-218, common/__init__.py:47-276)."""
    import samgraph.torch as sam
    for n in ("config init data_init sample_init train_init extract_start sample_once get_next_batch get_dgl_blocks "
              "get_graph_feat get_graph_label get_graph_row get_graph_col get_graph_num_src get_graph_num_dst "
              "num_epoch steps_per_epoch num_local_step num_class feat_dim report_step report_epoch_average "
              "log_step log_epoch_add get_log_epoch_value wait_one_child shutdown cpu gpu sample_types "
              "builtin_archs cache_policies kKHop3 kArch6 kCacheByDegree kLogEpochSampleTime kLogEpochNumSample "
              "kLogEpochCopyTime kLogEpochFeatureBytes kLogEpochMissBytes kLogL1CopyTime kL1Event_Sample").split():
        assert hasattr(sam, n), n
    assert (sam.kLogEpochSampleTime, sam.kLogEpochCopyTime, sam.kLogEpochNumSample) == (0, 8, 15)
    assert (sam.kLogL1NumSample, sam.kLogL1CopyTime, sam.kLogL1MissBytes, sam.kNumLogStepItems) == (0, 6, 13, 50)


def test_host_only_entry_points():
    l = xgnn_amd.lib()
    assert l.ggms_abi_version() == 3
    # shards a peer can open (ROCm 7.2: sizes with bit 31 set never open): rounded up to the next multiple of 4 GiB
    assert l.ggms_ipc_safe_bytes(1 << 20) == 1 << 20 and l.ggms_ipc_safe_bytes(3000 << 20) == 4 << 30
    assert l.ggms_ipc_safe_bytes(28_431_348_736) == 7 << 32 and l.ggms_ipc_safe_bytes(6000 << 20) == 6000 << 20
    # TableSize(num, 2), cuda_hashtable.cu:146-149
    for cap, want in [(2, 8), (3, 8), (4, 16), (1000, 2048), (2288000, 8388608), (8448000, 33554432)]:
        assert l.ggms_hashtable_num_buckets(cap) == want, cap
    assert [l.ggms_dtype_bytes(c) for c in range(7)] == [4, 8, 2, 1, 4, 1, 8]
    # GPURandomStates sizing, cuda_random_states.cu:70-97 with PredictNumNodes (common.cc:488-497)
    f = (ctypes.c_size_t * 2)(25, 10)
    assert l.ggms_random_states_count(7, f, 2, 8000, 0) == 8000 + 8000 * 25
    assert l.ggms_random_states_count(2, f, 2, 8000, 0) == 512 * 1024
    f3 = (ctypes.c_size_t * 3)(5, 5, 5)
    assert l.ggms_random_states_count(3, f3, 3, 100, 4) == ((100 * 6 * 6 + 63) // 64) * 256
    assert l.ggms_random_states_count(3, f3, 3, 100, 0) == 0  # no walks: no states (and no spin in the block-shape rule)
    # PredictRandomWalkMaxThreads (cuda_random_states.cu:48-60): the block-shape rule, which must not spin on 0 walks
    l.ggms_random_walk_num_states.restype = ctypes.c_size_t
    assert [l.ggms_random_walk_num_states(ctypes.c_size_t(100), ctypes.c_size_t(w)) for w in (0, 1, 4, 5, 300)] == \
        [0, 256, 512, 4 * 256, 100 * 256]


def test_no_cpu_fallback():
    import pytest
    import torch
    from xgnn_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception):
        ops.extract(torch.zeros(4, 4), torch.zeros(2, dtype=torch.int32))
