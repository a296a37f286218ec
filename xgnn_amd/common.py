"""Mirror of the reference's ``samgraph.common`` Python package
(/root/reference/samgraph/common/__init__.py): enum values, context helpers and the
ctypes front-end ``_basics`` over the ``samgraph_*`` C ABI (include/samgraph.h), which
lives in the same libggms_hip.so as the kernels.  Same names, same argument meaning.
"""
import ctypes as C

from . import _lib

# ---- enum mirrors (common/__init__.py:40-76, common.h:38-95) ----------------------------------
kCPU, kMMAP, kGPU = 0, 1, 2
kKHop0, kKHop1, kWeightedKHop, kRandomWalk, kWeightedKHopPrefix, kKHop2, kWeightedKHopHashDedup, kKHop3 = range(8)
kArch0, kArch1, kArch2, kArch3, kArch4, kArch5, kArch6, kArch7 = range(8)
(kCacheByDegree, kCacheByHeuristic, kCacheByPreSample, kCacheByDegreeHop, kCacheByPreSampleStatic,
 kCacheByFakeOptimal, kDynamicCache, kCacheByRandom) = range(8)


def cpu(device_id=0):
    return 'cpu:{:}'.format(device_id)


def gpu(device_id=0):
    return 'cuda:{:}'.format(device_id)


sample_types = {
    'khop0': kKHop0, 'khop1': kKHop1, 'khop2': kKHop2, 'khop3': kKHop3, 'random_walk': kRandomWalk,
    'weighted_khop': kWeightedKHop, 'weighted_khop_prefix': kWeightedKHopPrefix,
    'weighted_khop_hash_dedup': kWeightedKHopHashDedup,
}
builtin_archs = {
    'arch0': {'arch': kArch0, 'sampler_ctx': cpu(), 'trainer_ctx': gpu(0)},
    'arch1': {'arch': kArch1, 'sampler_ctx': gpu(0), 'trainer_ctx': gpu(0)},
    'arch2': {'arch': kArch2, 'sampler_ctx': gpu(0), 'trainer_ctx': gpu(0)},
    'arch3': {'arch': kArch3, 'sampler_ctx': gpu(0), 'trainer_ctx': gpu(1)},
    'arch4': {'arch': kArch4, 'sampler_ctx': gpu(1), 'trainer_ctx': gpu(0)},
    'arch5': {'arch': kArch5}, 'arch6': {'arch': kArch6}, 'arch7': {'arch': kArch7},
}
cache_policies = {
    'degree': kCacheByDegree, 'heuristic': kCacheByHeuristic, 'pre_sample': kCacheByPreSample,
    'degree_hop': kCacheByDegreeHop, 'presample_static': kCacheByPreSampleStatic,
    'fake_optimal': kCacheByFakeOptimal, 'dynamic_cache': kDynamicCache, 'random': kCacheByRandom,
}

# ---- profiler item codes, by name, in the order of common/profiler.h:30-163 ---------------------
_INIT_ITEMS = """kLogInitL1Common kLogInitL1Sampler kLogInitL1Trainer kLogInitL1GraphMemory kLogInitL1FeatMemory
kLogInitL1WorkspaceTotalMemory kLogInitL2LoadDataset kLogInitL2DistQueue kLogInitL2Presample kLogInitL2InternalState
kLogInitL2BuildCache kLogInitL3LoadDatasetMMap kLogInitL3LoadDatasetCopy kLogInitL3DistQueueAlloc kLogInitL3DistQueuePin
kLogInitL3DistQueuePush kLogInitL3PresampleInit kLogInitL3PresampleSample kLogInitL3PresampleCopy kLogInitL3PresampleCount
kLogInitL3PresampleSort kLogInitL3PresampleReset kLogInitL3PresampleGetRank kLogInitL3InternalStateCreateCtx
kLogInitL3InternalStateCreateStream kNumLogInitItems""".split()
_STEP_ITEMS = """kLogL1NumSample kLogL1NumNode kLogL1SampleTotalTime kLogL1SampleTime kLogL1SendTime kLogL1RecvTime
kLogL1CopyTime kLogL1ConvertTime kLogL1TrainTime kLogL1FeatureBytes kLogL1LabelBytes kLogL1IdBytes kLogL1GraphBytes
kLogL1MissBytes kLogL1PrefetchAdvanced kLogL1GetNeighbourTime kLogL1SamplerId kLogL2ShuffleTime kLogL2LastLayerTime
kLogL2LastLayerSize kLogL2CoreSampleTime kLogL2IdRemapTime kLogL2GraphCopyTime kLogL2IdCopyTime kLogL2ExtractTime
kLogL2FeatCopyTime kLogL2CacheCopyTime kLogL3KHopSampleCooTime kLogL3KHopSampleSortCooTime kLogL3KHopSampleCountEdgeTime
kLogL3KHopSampleCompactEdgesTime kLogL3RandomWalkSampleCooTime kLogL3RandomWalkTopKTime kLogL3RandomWalkTopKStep1Time
kLogL3RandomWalkTopKStep2Time kLogL3RandomWalkTopKStep3Time kLogL3RandomWalkTopKStep4Time kLogL3RandomWalkTopKStep5Time
kLogL3RandomWalkTopKStep6Time kLogL3RandomWalkTopKStep7Time kLogL3RemapFillUniqueTime kLogL3RemapPopulateTime
kLogL3RemapMapNodeTime kLogL3RemapMapEdgeTime kLogL3CacheGetIndexTime KLogL3CacheCopyIndexTime kLogL3CacheExtractMissTime
kLogL3CacheCopyMissTime kLogL3CacheCombineMissTime kLogL3CacheCombineCacheTime kNumLogStepItems""".split()
_EPOCH_ITEMS = """kLogEpochSampleTime KLogEpochSampleGetCacheMissIndexTime kLogEpochSampleSendTime kLogEpochSampleTotalTime
kLogEpochCoreSampleTime kLogEpochSampleCooTime kLogEpochIdRemapTime kLogEpochShuffleTime kLogEpochCopyTime
kLogEpochConvertTime kLogEpochTrainTime kLogEpochTotalTime kLogEpochFeatureBytes kLogEpochMissBytes
kLogEpochLocalCacheBytes kLogEpochNumSample kNumLogEpochItems""".split()
_EVENT_ITEMS = """kL0Event_Train_Step kL1Event_Sample kL2Event_Sample_Shuffle kL2Event_Sample_Core kL2Event_Sample_IdRemap
kL3Event_Sample_Core_Coo kL1Event_Copy kL2Event_Copy_Id kL2Event_Copy_Graph kL2Event_Copy_Extract kL2Event_Copy_FeatCopy
kL2Event_Copy_CacheCopy kL3Event_Copy_CacheCopy_GetIndex kL3Event_Copy_CacheCopy_CopyIndex
kL3Event_Copy_CacheCopy_ExtractMiss kL3Event_Copy_CacheCopy_CopyMiss kL3Event_Copy_CacheCopy_CombineMiss
kL3Event_Copy_CacheCopy_CombineCache kL1Event_Convert kL1Event_Train""".split()
for _names in (_INIT_ITEMS, _STEP_ITEMS, _EPOCH_ITEMS, _EVENT_ITEMS):
    for _i, _n in enumerate(_names):
        globals()[_n] = _i


class Tensor(C.Structure):
    """samgraph_tensor_t (include/samgraph.h)"""
    _fields_ = [("data", C.c_void_p), ("shape", C.c_int64 * 2), ("ndim", C.c_int32), ("dtype", C.c_int32),
                ("device_type", C.c_int32), ("device_id", C.c_int32)]


_u64, _i, _d, _sz = C.c_uint64, C.c_int, C.c_double, C.c_size_t
_TP = C.POINTER(Tensor)
SAMGRAPH_SYMBOLS = {
    "samgraph_config": (None, [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), _sz]),
    "samgraph_init": (None, []), "samgraph_start": (None, []), "samgraph_shutdown": (None, []),
    "samgraph_data_init": (None, []), "samgraph_sample_init": (None, [_i, C.c_char_p]),
    "samgraph_train_init": (None, [_i, C.c_char_p]), "samgraph_extract_start": (None, [_i]),
    "samgraph_um_sample_init": (None, [_i]), "samgraph_switch_init": (None, [_i, C.c_char_p, C.c_double]),
    "samgraph_wait_one_child": (_i, []), "samgraph_forward_barrier": (None, []),
    "samgraph_num_epoch": (_sz, []), "samgraph_steps_per_epoch": (_sz, []), "samgraph_num_local_step": (_sz, []),
    "samgraph_num_class": (_sz, []), "samgraph_feat_dim": (_sz, []), "samgraph_get_next_batch": (_u64, []),
    "samgraph_sample_once": (None, []),
    "samgraph_get_graph_num_src": (_sz, [_u64, _i]), "samgraph_get_graph_num_dst": (_sz, [_u64, _i]),
    "samgraph_get_graph_num_edge": (_sz, [_u64, _i]),
    "samgraph_log_step": (None, [_u64, _u64, _i, _d]), "samgraph_log_step_by_key": (None, [_u64, _i, _d]),
    "samgraph_log_step_add": (None, [_u64, _u64, _i, _d]), "samgraph_log_epoch_add": (None, [_u64, _i, _d]),
    "samgraph_get_log_init_value": (_d, [_i]), "samgraph_get_log_step_value": (_d, [_u64, _u64, _i]),
    "samgraph_get_log_step_value_by_key": (_d, [_u64, _i]), "samgraph_get_log_epoch_value": (_d, [_u64, _i]),
    "samgraph_report_init": (None, []), "samgraph_report_step": (None, [_u64, _u64]),
    "samgraph_report_step_average": (None, [_u64, _u64]), "samgraph_report_epoch": (None, [_u64]),
    "samgraph_report_epoch_average": (None, [_u64]), "samgraph_report_node_access": (None, []),
    "samgraph_trace_step_begin": (None, [_u64, _i, _u64]), "samgraph_trace_step_end": (None, [_u64, _i, _u64]),
    "samgraph_trace_step_begin_now": (None, [_u64, _i]), "samgraph_trace_step_end_now": (None, [_u64, _i]),
    "samgraph_dump_trace": (None, []),
    "samgraph_get_graph_feat": (None, [_u64, _TP]), "samgraph_get_graph_label": (None, [_u64, _TP]),
    "samgraph_get_graph_row": (None, [_u64, _i, _TP]), "samgraph_get_graph_col": (None, [_u64, _i, _TP]),
    "samgraph_get_graph_data": (None, [_u64, _i, _TP]), "samgraph_get_dataset_feat": (None, [_TP]),
    "samgraph_get_dataset_label": (None, [_TP]), "samgraph_get_graph_input_nodes": (None, [_u64, _TP]),
    "samgraph_get_graph_output_nodes": (None, [_u64, _TP]),
    "samgraph_batch_retain": (None, [_u64]), "samgraph_batch_release": (None, [_u64]),
}


class SamGraphBasics(object):
    """ctypes front-end, method for method the reference's SamGraphBasics (common/__init__.py:279-537)."""

    def __init__(self):
        self._h = None

    @property
    def C_LIB_CTYPES(self):
        if self._h is None:
            h = _lib.lib()
            for name, (res, args) in SAMGRAPH_SYMBOLS.items():
                fn = getattr(h, name)
                fn.restype = res
                fn.argtypes = args
            self._h = h
        return self._h

    def config(self, run_config: dict):
        keys = [str(k).encode() for k in run_config.keys()]
        vals = []
        for value in run_config.values():
            vals.append((' '.join(str(v) for v in value) if isinstance(value, list) else str(value)).encode())
        n = len(keys)
        return self.C_LIB_CTYPES.samgraph_config((C.c_char_p * n)(*keys), (C.c_char_p * n)(*vals), n)

    def init(self): return self.C_LIB_CTYPES.samgraph_init()
    def data_init(self): return self.C_LIB_CTYPES.samgraph_data_init()
    def sample_init(self, worker_id, ctx): return self.C_LIB_CTYPES.samgraph_sample_init(worker_id, ctx.encode())
    def train_init(self, worker_id, ctx): return self.C_LIB_CTYPES.samgraph_train_init(worker_id, ctx.encode())
    def extract_start(self, count): return self.C_LIB_CTYPES.samgraph_extract_start(count)
    def um_sample_init(self, num_workers): return self.C_LIB_CTYPES.samgraph_um_sample_init(num_workers)

    def switch_init(self, worker_id, ctx, cache_percentage):
        return self.C_LIB_CTYPES.samgraph_switch_init(worker_id, ctx.encode(), cache_percentage)

    def num_local_step(self): return self.C_LIB_CTYPES.samgraph_num_local_step()
    def start(self): return self.C_LIB_CTYPES.samgraph_start()
    def shutdown(self): return self.C_LIB_CTYPES.samgraph_shutdown()
    def num_class(self): return self.C_LIB_CTYPES.samgraph_num_class()
    def feat_dim(self): return self.C_LIB_CTYPES.samgraph_feat_dim()
    def num_epoch(self): return self.C_LIB_CTYPES.samgraph_num_epoch()
    def steps_per_epoch(self): return self.C_LIB_CTYPES.samgraph_steps_per_epoch()
    def get_next_batch(self): return self.C_LIB_CTYPES.samgraph_get_next_batch()
    def get_graph_num_src(self, key, graph_id): return self.C_LIB_CTYPES.samgraph_get_graph_num_src(key, graph_id)
    def get_graph_num_dst(self, key, graph_id): return self.C_LIB_CTYPES.samgraph_get_graph_num_dst(key, graph_id)
    def get_graph_num_edge(self, key, graph_id): return self.C_LIB_CTYPES.samgraph_get_graph_num_edge(key, graph_id)
    def sample_once(self): return self.C_LIB_CTYPES.samgraph_sample_once()
    def log_step(self, epoch, step, item, val): return self.C_LIB_CTYPES.samgraph_log_step(epoch, step, item, val)
    def log_step_by_key(self, key, item, val): return self.C_LIB_CTYPES.samgraph_log_step_by_key(key, item, val)
    def log_step_add(self, epoch, step, item, val): return self.C_LIB_CTYPES.samgraph_log_step_add(epoch, step, item, val)
    def log_epoch_add(self, epoch, item, val): return self.C_LIB_CTYPES.samgraph_log_epoch_add(epoch, item, val)
    def get_log_init_value(self, item): return self.C_LIB_CTYPES.samgraph_get_log_init_value(item)
    def get_log_step_value(self, epoch, step, item): return self.C_LIB_CTYPES.samgraph_get_log_step_value(epoch, step, item)
    def get_log_step_value_by_key(self, key, item): return self.C_LIB_CTYPES.samgraph_get_log_step_value_by_key(key, item)
    def get_log_epoch_value(self, epoch, item): return self.C_LIB_CTYPES.samgraph_get_log_epoch_value(epoch, item)
    def report_init(self): return self.C_LIB_CTYPES.samgraph_report_init()
    def report_step(self, epoch, step): return self.C_LIB_CTYPES.samgraph_report_step(epoch, step)
    def report_step_average(self, epoch, step): return self.C_LIB_CTYPES.samgraph_report_step_average(epoch, step)
    def report_epoch(self, epoch): return self.C_LIB_CTYPES.samgraph_report_epoch(epoch)
    def report_epoch_average(self, epoch): return self.C_LIB_CTYPES.samgraph_report_epoch_average(epoch)
    def report_node_access(self): return self.C_LIB_CTYPES.samgraph_report_node_access()
    def trace_step_begin(self, key, item, us): return self.C_LIB_CTYPES.samgraph_trace_step_begin(key, item, us)
    def trace_step_end(self, key, item, us): return self.C_LIB_CTYPES.samgraph_trace_step_end(key, item, us)
    def trace_step_begin_now(self, key, item): return self.C_LIB_CTYPES.samgraph_trace_step_begin_now(key, item)
    def trace_step_end_now(self, key, item): return self.C_LIB_CTYPES.samgraph_trace_step_end_now(key, item)
    def dump_trace(self): return self.C_LIB_CTYPES.samgraph_dump_trace()
    def forward_barrier(self): return self.C_LIB_CTYPES.samgraph_forward_barrier()
    def wait_one_child(self): return self.C_LIB_CTYPES.samgraph_wait_one_child()


_basics = SamGraphBasics()
