"""Mirror of the reference's ``samgraph.torch`` adapter (/root/reference/samgraph/torch/adapter.py:63-218):
the same function names; tensors are zero-copy views of engine-owned device buffers.

The reference's pybind11 module returned torch::from_blob tensors whose deleter captured the buffer
(adapter.cc:62-77).  Here the C ABI hands back {pointer, shape, dtype, device}; the view goes through
``__cuda_array_interface__`` and the wrapper object holds a retain on the batch until the tensor dies.
"""
import ctypes as C

import numpy as np
import torch

from .common import *  # noqa: F401,F403  (enum mirrors, like `from samgraph.common import *`)
from .common import _basics, Tensor as _CTensor

for _name in ("config init start num_class feat_dim num_epoch steps_per_epoch get_next_batch get_graph_num_src "
              "get_graph_num_dst get_graph_num_edge shutdown sample_once log_step log_step_add log_epoch_add "
              "get_log_init_value get_log_step_value get_log_epoch_value report_init report_step report_step_average "
              "report_epoch report_epoch_average report_node_access trace_step_begin trace_step_end "
              "trace_step_begin_now trace_step_end_now dump_trace forward_barrier wait_one_child log_step_by_key "
              "get_log_step_value_by_key data_init sample_init train_init extract_start num_local_step "
              "um_sample_init switch_init").split():
    globals()[_name] = getattr(_basics, _name)

# DataType code -> (numpy typestr, torch dtype); common/common.h:38-46, adapter.cc:33-53
_DT = {0: ("<f4", torch.float32), 1: ("<f8", torch.float64), 2: ("<f2", torch.float16), 3: ("|u1", torch.uint8),
       4: ("<i4", torch.int32), 5: ("|i1", torch.int8), 6: ("<i8", torch.int64)}


class _DeviceView(object):
    """Zero-copy view of an engine buffer; keeps the batch alive (adapter.cc:70-73 deleter capture)."""

    def __init__(self, t, key):
        shape = tuple(int(t.shape[i]) for i in range(t.ndim))
        self._key = key
        self.__cuda_array_interface__ = {"shape": shape, "typestr": _DT[t.dtype][0],
                                         "data": (int(t.data or 0), False), "version": 2, "strides": None}
        if key is not None:
            _basics.C_LIB_CTYPES.samgraph_batch_retain(key)

    def __del__(self):
        if getattr(self, "_key", None) is not None:
            try:
                _basics.C_LIB_CTYPES.samgraph_batch_release(self._key)
            except Exception:
                pass


def _wrap(t, key):
    n = 1
    for i in range(t.ndim):
        n *= int(t.shape[i])
    if t.device_type == 2:
        dev = torch.device("cuda", t.device_id)
        if n == 0:
            return torch.empty(tuple(int(t.shape[i]) for i in range(t.ndim)), dtype=_DT[t.dtype][1], device=dev)
        return torch.as_tensor(_DeviceView(t, key), device=dev)
    # host memory: numpy view over the mapped dataset file (GetDatasetFeature, adapter.cc:136-152)
    shape = tuple(int(t.shape[i]) for i in range(t.ndim))
    if n == 0 or not t.data:
        return torch.empty(shape, dtype=_DT[t.dtype][1])
    buf = (C.c_char * (n * np.dtype(_DT[t.dtype][0]).itemsize)).from_address(t.data)
    a = np.frombuffer(buf, dtype=_DT[t.dtype][0]).reshape(shape)
    # a batch of the CPU deployment (arch0, host trainer) lives in a slot the engine reuses: hand out a copy;
    # the dataset tensors (key None) stay zero-copy views of the mapped files
    return torch.from_numpy(a.copy() if key is not None else a)


def _get(fn, key, *args):
    t = _CTensor()
    getattr(_basics.C_LIB_CTYPES, fn)(*((key,) if key is not None else ()), *args, C.byref(t))
    return _wrap(t, key)


def get_graph_feat(batch_key):
    batch_feat = _get("samgraph_get_graph_feat", batch_key)
    if batch_feat.dtype != torch.float32:
        batch_feat = batch_feat.float()
    return batch_feat


def get_graph_label(batch_key): return _get("samgraph_get_graph_label", batch_key)
def get_graph_row(batch_key, layer_idx): return _get("samgraph_get_graph_row", batch_key, layer_idx)
def get_graph_col(batch_key, layer_idx): return _get("samgraph_get_graph_col", batch_key, layer_idx)
def get_graph_data(batch_key, layer_idx): return _get("samgraph_get_graph_data", batch_key, layer_idx)
def get_dataset_feat(): return _get("samgraph_get_dataset_feat", None)
def get_dataset_label(): return _get("samgraph_get_dataset_label", None)
def get_graph_input_nodes(batch_key): return _get("samgraph_get_graph_input_nodes", batch_key)
def get_graph_output_nodes(batch_key): return _get("samgraph_get_graph_output_nodes", batch_key)


def _create_dgl_block(data, num_src_nodes, num_dst_nodes):
    import dgl  # DGL-on-ROCm is needed only here, exactly where the reference needs it (adapter.py:131-136)
    from dgl.heterograph import DGLBlock
    row, col = data
    gidx = dgl.heterograph_index.create_unitgraph_from_coo(2, num_src_nodes, num_dst_nodes, row, col,
                                                           ['coo', 'csr', 'csc'])
    return DGLBlock(gidx, (['_N'], ['_N']), ['_E'])


def get_graph_coo(batch_key, num_layers):
    """(row, col, num_src, num_dst) per layer without DGL -- what get_dgl_blocks feeds to DGL."""
    return [(get_graph_row(batch_key, i), get_graph_col(batch_key, i),
             get_graph_num_src(batch_key, i), get_graph_num_dst(batch_key, i)) for i in range(num_layers)]  # noqa: F405


def get_dgl_blocks(batch_key, num_layers, with_feat=True):
    feat = get_graph_feat(batch_key) if with_feat else None
    label = get_graph_label(batch_key) if with_feat else None
    blocks = [_create_dgl_block((row, col), ns, nd) for row, col, ns, nd in get_graph_coo(batch_key, num_layers)]
    return blocks, feat, label


def get_dgl_blocks_with_weights(batch_key, num_layers, with_feat=True):
    blocks, feat, label = get_dgl_blocks(batch_key, num_layers, with_feat)
    for i, block in enumerate(blocks):
        block.edata['weights'] = get_graph_data(batch_key, i)
    return blocks, feat, label


def notify_sampler_ready(barrier): barrier.wait()
def wait_for_sampler_ready(barrier): barrier.wait()


def load_subtensor(batch_key, feat, label, device):
    input_nodes = get_graph_input_nodes(batch_key).to(feat.device)
    output_nodes = get_graph_output_nodes(batch_key).to(label.device)
    batch_inputs = torch.index_select(feat, 0, input_nodes.long()).to(device, dtype=torch.float32)
    batch_labels = torch.index_select(label, 0, output_nodes.long()).to(device)
    return batch_inputs, batch_labels
