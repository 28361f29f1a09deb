"""ctypes binding of the VecSim C ABI served by redisearch_amd/lib/libVectorSimilarity.so.

This is the host-side mirror of the reference interface for the path: same function names, argument
meaning and error behaviour as deps/VectorSimilarity's C API as RediSearch calls it (reference
src/iterators/hybrid_reader.c, src/vector_index.c, src/redisearch_rs/c_wrappers/vecsim/src/*.rs).
It is a thin binding only -- every distance, selection and scoring operation happens in the HIP
library; if the library (or a GPU) is missing the calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

# enums (include/VecSim/vec_sim_common.h)
VecSimType_FLOAT32, VecSimType_FLOAT64, VecSimType_BFLOAT16, VecSimType_FLOAT16 = 0, 1, 2, 3
VecSimType_INT8, VecSimType_UINT8 = 4, 5
VecSimAlgo_BF, VecSimAlgo_HNSWLIB, VecSimAlgo_TIERED, VecSimAlgo_SVS = 0, 1, 2, 3
VecSimMetric_L2, VecSimMetric_IP, VecSimMetric_Cosine = 0, 1, 2
BY_SCORE, BY_ID = 0, 1
VecSim_QueryReply_OK, VecSim_QueryReply_TimedOut = 0, 1
EMPTY_MODE, STANDARD_KNN, HYBRID_ADHOC_BF, HYBRID_BATCHES, HYBRID_BATCHES_TO_ADHOC_BF, RANGE_QUERY = range(6)
QUERY_TYPE_NONE, QUERY_TYPE_KNN, QUERY_TYPE_HYBRID, QUERY_TYPE_RANGE = range(4)
(VecSim_OK, VecSimParamResolverErr_AlreadySet, VecSimParamResolverErr_UnknownParam,
 VecSimParamResolverErr_BadValue, VecSimParamResolverErr_InvalidPolicy_NExits,
 VecSimParamResolverErr_InvalidPolicy_NHybrid, VecSimParamResolverErr_InvalidPolicy_NRange,
 VecSimParamResolverErr_InvalidPolicy_AdHoc_With_BatchSize,
 VecSimParamResolverErr_InvalidPolicy_AdHoc_With_EfRuntime) = range(9)

TYPE_NP = {VecSimType_FLOAT32: np.float32, VecSimType_FLOAT64: np.float64, VecSimType_FLOAT16: np.float16,
           VecSimType_BFLOAT16: np.uint16, VecSimType_INT8: np.int8, VecSimType_UINT8: np.uint8}

_sz, _vp, _dbl, _i, _b = C.c_size_t, C.c_void_p, C.c_double, C.c_int, C.c_bool


class BFParams(C.Structure):
    _fields_ = [("type", _i), ("dim", _sz), ("metric", _i), ("multi", _b), ("initialCapacity", _sz),
                ("blockSize", _sz)]


class HNSWParams(C.Structure):
    _fields_ = [("type", _i), ("dim", _sz), ("metric", _i), ("multi", _b), ("initialCapacity", _sz),
                ("blockSize", _sz), ("M", _sz), ("efConstruction", _sz), ("efRuntime", _sz), ("epsilon", _dbl)]


class SVSParams(C.Structure):
    _fields_ = [("type", _i), ("dim", _sz), ("metric", _i), ("multi", _b), ("blockSize", _sz), ("quantBits", _i),
                ("alpha", C.c_float), ("graph_max_degree", _sz), ("construction_window_size", _sz),
                ("max_candidate_pool_size", _sz), ("prune_to", _sz), ("use_search_history", _i),
                ("num_threads", _sz), ("search_window_size", _sz), ("search_buffer_capacity", _sz),
                ("leanvec_dim", _sz), ("epsilon", _dbl)]


class _TieredSpecific(C.Union):
    _fields_ = [("swapJobThreshold", _sz), ("svs", _sz * 3)]


class TieredIndexParams(C.Structure):
    _fields_ = [("jobQueue", _vp), ("jobQueueCtx", _vp), ("submitCb", _vp), ("flatBufferLimit", _sz),
                ("primaryIndexParams", _vp), ("specificParams", _TieredSpecific)]


class AlgoParams(C.Union):
    _fields_ = [("hnswParams", HNSWParams), ("bfParams", BFParams), ("tieredParams", TieredIndexParams),
                ("svsParams", SVSParams)]


class VecSimParams(C.Structure):
    _fields_ = [("algo", _i), ("algoParams", AlgoParams), ("logCtx", _vp)]


class _RuntimeUnion(C.Union):
    _fields_ = [("hnsw", _sz * 2), ("svs", _sz * 4), ("disk", _i)]


class VecSimQueryParams(C.Structure):
    _anonymous_ = ("u",)
    _fields_ = [("u", _RuntimeUnion), ("batchSize", _sz), ("searchMode", _i), ("timeoutCtx", _vp)]


class VecSimRawParam(C.Structure):
    _fields_ = [("name", C.c_char_p), ("nameLen", _sz), ("value", C.c_char_p), ("valLen", _sz)]


class VecSimIndexBasicInfo(C.Structure):
    _fields_ = [("algo", _i), ("metric", _i), ("type", _i), ("isMulti", _b), ("isTiered", _b), ("isDisk", _b),
                ("blockSize", _sz), ("dim", _sz)]


class VecSimIndexStatsInfo(C.Structure):
    _fields_ = [("memory", _sz), ("numberOfMarkedDeleted", _sz), ("directHNSWInsertions", _sz),
                ("flatBufferSize", _sz)]


class FieldValue(C.Union):
    _fields_ = [("floatingPointValue", _dbl), ("integerValue", C.c_int64), ("uintegerValue", C.c_uint64),
                ("stringValue", C.c_char_p), ("iteratorValue", _vp)]


class VecSim_InfoField(C.Structure):
    _fields_ = [("fieldName", C.c_char_p), ("fieldType", _i), ("fieldValue", FieldValue)]


class VecSimMemoryFunctions(C.Structure):
    _fields_ = [("allocFunction", _vp), ("callocFunction", _vp), ("reallocFunction", _vp), ("freeFunction", _vp)]


TIMEOUT_CB = C.CFUNCTYPE(_i, _vp)
LOG_CB = C.CFUNCTYPE(None, _vp, C.c_char_p, C.c_char_p)

# every symbol declared in include/VecSim/*.h and include/rsgpu_ext.h: (restype, argtypes)
ABI = {
    # vec_sim.h
    "VecSimIndex_New": (_vp, [C.POINTER(VecSimParams)]),
    "VecSimIndex_NewDisk": (_vp, [_vp]),
    "VecSimIndex_Free": (None, [_vp]),
    "VecSimIndex_EstimateInitialSize": (_sz, [C.POINTER(VecSimParams)]),
    "VecSimIndex_EstimateElementSize": (_sz, [C.POINTER(VecSimParams)]),
    "VecSimIndex_AddVector": (_i, [_vp, _vp, _sz]),
    "VecSimIndex_DeleteVector": (_i, [_vp, _sz]),
    "VecSimIndex_IndexSize": (_sz, [_vp]),
    "VecSimIndex_BasicInfo": (VecSimIndexBasicInfo, [_vp]),
    "VecSimIndex_StatsInfo": (VecSimIndexStatsInfo, [_vp]),
    "VecSimIndex_DebugInfoIterator": (_vp, [_vp]),
    "VecSimDebugInfoIterator_NumberOfFields": (_sz, [_vp]),
    "VecSimDebugInfoIterator_HasNextField": (_b, [_vp]),
    "VecSimDebugInfoIterator_NextField": (C.POINTER(VecSim_InfoField), [_vp]),
    "VecSimDebugInfoIterator_Free": (None, [_vp]),
    "VecSimIndex_ResolveParams": (_i, [_vp, C.POINTER(VecSimRawParam), _i, C.POINTER(VecSimQueryParams), _i]),
    "VecSimIndex_TopKQuery": (_vp, [_vp, _vp, _sz, C.POINTER(VecSimQueryParams), _i]),
    "VecSimIndex_RangeQuery": (_vp, [_vp, _vp, _dbl, C.POINTER(VecSimQueryParams), _i]),
    "VecSimIndex_GetDistanceFrom_Unsafe": (_dbl, [_vp, _sz, _vp]),
    "VecSimIndex_PreferAdHocSearch": (_b, [_vp, _sz, _sz, _b]),
    "VecSimBatchIterator_New": (_vp, [_vp, _vp, C.POINTER(VecSimQueryParams)]),
    "VecSimIndex_AdhocBfCtx_New": (_vp, [_vp, _vp]),
    "VecSimIndex_AdhocBfCtx_GetDistanceFrom": (_dbl, [_vp, _sz]),
    "VecSimIndex_AdhocBfCtx_GetExactDistances": (None, [_vp, _vp, _vp, _sz]),
    "VecSimIndex_AdhocBfCtx_Free": (None, [_vp]),
    "VecSimTieredIndex_AcquireSharedLocks": (None, [_vp]),
    "VecSimTieredIndex_ReleaseSharedLocks": (None, [_vp]),
    "VecSimTieredIndex_GC": (None, [_vp]),
    "VecSimDebug_GetElementNeighborsInHNSWGraph": (_i, [_vp, _sz, _vp]),
    "VecSimDebug_ReleaseElementNeighborsInHNSWGraph": (None, [_vp]),
    "VecSim_Normalize": (None, [_vp, _sz, _i]),
    "VecSimParams_GetQueryBlobSize": (_sz, [_i, _sz, _i]),
    "VecSim_SetMemoryFunctions": (None, [VecSimMemoryFunctions]),
    "VecSim_SetTimeoutCallbackFunction": (None, [TIMEOUT_CB]),
    "VecSim_SetLogCallbackFunction": (None, [LOG_CB]),
    "VecSim_SetWriteMode": (None, [_i]),
    "VecSim_UpdateThreadPoolSize": (None, [_sz]),
    "VecSim_GetSharedMemory": (_sz, []),
    # query_results.h
    "VecSimQueryResult_GetId": (_sz, [_vp]),
    "VecSimQueryResult_GetScore": (_dbl, [_vp]),
    "VecSimQueryReply_Len": (_sz, [_vp]),
    "VecSimQueryReply_GetCode": (_i, [_vp]),
    "VecSimQueryReply_Free": (None, [_vp]),
    "VecSimQueryReply_GetIterator": (_vp, [_vp]),
    "VecSimQueryReply_IteratorNext": (_vp, [_vp]),
    "VecSimQueryReply_IteratorHasNext": (_b, [_vp]),
    "VecSimQueryReply_IteratorReset": (None, [_vp]),
    "VecSimQueryReply_IteratorFree": (None, [_vp]),
    "VecSimBatchIterator_Next": (_vp, [_vp, _sz, _i]),
    "VecSimBatchIterator_HasNext": (_b, [_vp]),
    "VecSimBatchIterator_Free": (None, [_vp]),
    "VecSimBatchIterator_Reset": (None, [_vp]),
    # rsgpu_ext.h
    "RSGPU_LastError": (C.c_char_p, []),
    "RSGPU_DeviceCount": (_i, []),
    "RSGPU_FlatIndex_Reserve": (_i, [_vp, _sz]),
    "RSGPU_FlatIndex_AddDeviceRows": (_i, [_vp, _vp, _sz, _sz]),
    "RSGPU_FlatIndex_AddPhiloxRows": (C.c_long, [_vp, C.c_uint64, C.c_uint64, _sz, _sz]),
    "RSGPU_FlatIndex_ReadRows": (_i, [_vp, _sz, _sz, _vp]),
    "RSGPU_FlatIndex_LabelTable": (_i, [_vp]),
    "RSGPU_FlatIndex_TopKDevice": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "RSGPU_ShardedIndex_New": (_vp, [C.POINTER(VecSimParams), _i, _vp, _i]),
    "RSGPU_ShardedIndex_Free": (None, [_vp]),
    "RSGPU_ShardedIndex_NumShards": (_i, [_vp]),
    "RSGPU_ShardedIndex_FromHandle": (_vp, [_vp]),
    "RSGPU_ShardedIndex_ShardDevice": (_i, [_vp, _i]),
    "RSGPU_ShardedIndex_Shard": (_vp, [_vp, _i]),
    "RSGPU_ShardedIndex_IndexSize": (_sz, [_vp]),
    "RSGPU_ShardedIndex_AddVector": (_i, [_vp, _vp, _sz]),
    "RSGPU_ShardedIndex_DeleteVector": (_i, [_vp, _sz]),
    "RSGPU_ShardedIndex_GetDistanceFrom": (_dbl, [_vp, _sz, _vp]),
    "RSGPU_ShardedIndex_TopKQuery": (_vp, [_vp, _vp, _sz, C.POINTER(VecSimQueryParams), _i]),
    "RSGPU_ShardedIndex_RangeQuery": (_vp, [_vp, _vp, _dbl, C.POINTER(VecSimQueryParams), _i]),
    "RSGPU_FlatIndex_TopKBatch": (_i, [_vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "RSGPU_MergeTopK": (_i, [_i, _vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "RSGPU_MergeTopKHost": (_i, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "RSGPU_SetProfiling": (None, [_i]),
    "RSGPU_ResetProfile": (None, []),
    "RSGPU_GetScanProfile": (None, [C.POINTER(C.c_uint64), C.POINTER(_dbl), C.POINTER(C.c_uint64)]),
    "RSGPU_GetLastScanKernel": (C.c_char_p, [C.c_char_p, _sz]),
    "RSGPU_GetTwoStageStats": (None, [C.POINTER(C.c_uint64)]),
    "RSGPU_ShardedIndex_GetExchangeStats": (None, [_vp, C.POINTER(C.c_uint64), _i]),
    "RSGPU_ShardedIndex_GetRcclStats": (None, [_vp, C.POINTER(C.c_uint64), _i]),
    "RSGPU_GetCoalesceStats": (None, [C.POINTER(C.c_uint64)]),
    "RSGPU_LastBatchRoute": (C.c_int, []),
    "RSGPU_GetWidePassStats": (None, [C.POINTER(C.c_uint64)]),
    "RSGPU_GetCoalesceTimeouts": (C.c_uint64, []),
    "RSGPU_ShardComm_GetUniqueId": (_i, [_vp]),
    "RSGPU_ShardComm_Init": (_vp, [_i, _i, _vp, _i]),
    "RSGPU_ShardComm_Free": (None, [_vp]),
    "RSGPU_ShardComm_World": (_i, [_vp]),
    "RSGPU_ShardComm_TopK": (C.c_long, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "RSGPU_ShardComm_GetStats": (None, [_vp, C.POINTER(C.c_uint64), _i]),
    "RSGPU_MergeTopKDevice": (C.c_long, [_i, _vp, _vp, _sz, _sz, _vp, _vp]),
    "RSGPU_ResetCoalesceStats": (None, []),
    "RSGPU_GetLastMqScanKernel": (C.c_char_p, [C.c_char_p, _sz]),
    "RSGPU_ResetTwoStageStats": (None, []),
    "RSGPU_SetTuning": (_i, [C.c_char_p, _i]),
    "RSGPU_ReleaseWorkspaces": (None, []),
}

_lib = None


def load(path=None):
    """dlopen the engine (never builds, never falls back)."""
    global _lib
    if _lib is None:
        path = path or _build.lib_path()
        if not os.path.exists(path):
            raise RuntimeError("%s is missing: run `python -m redisearch_amd.build` (hipcc, gfx950)" % path)
        lib = C.CDLL(path)
        for name, (res, args) in ABI.items():
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        _lib = lib
    return _lib


def last_error():
    return (load().RSGPU_LastError() or b"").decode()


def _p(a):
    return a.ctypes.data_as(_vp)


def to_blob(vec, vtype):
    if vtype == VecSimType_BFLOAT16:
        f = np.ascontiguousarray(vec, dtype=np.float32)
        u = f.view(np.uint32)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)  # round to nearest even
        return r
    return np.ascontiguousarray(vec, dtype=TYPE_NP[vtype])


def flat_params(vtype, dim, metric, multi=False, initial_capacity=0, block_size=1024, algo=VecSimAlgo_BF):
    p = VecSimParams()
    p.algo = algo
    p.algoParams.bfParams = BFParams(vtype, dim, metric, multi, initial_capacity, block_size)
    p.logCtx = None
    return p


class QueryReply:
    """Owned VecSimQueryReply (reference c_wrappers/vecsim/src/reply.rs)."""

    def __init__(self, ptr):
        self.lib = load()
        self.ptr = ptr

    @property
    def code(self):
        return self.lib.VecSimQueryReply_GetCode(self.ptr)

    def __len__(self):
        return self.lib.VecSimQueryReply_Len(self.ptr)

    def results(self):
        it = self.lib.VecSimQueryReply_GetIterator(self.ptr)
        ids, scores = [], []
        while self.lib.VecSimQueryReply_IteratorHasNext(it):
            r = self.lib.VecSimQueryReply_IteratorNext(it)
            ids.append(self.lib.VecSimQueryResult_GetId(r))
            scores.append(self.lib.VecSimQueryResult_GetScore(r))
        assert self.lib.VecSimQueryReply_IteratorNext(it) is None
        self.lib.VecSimQueryReply_IteratorFree(it)
        return np.array(ids, dtype=np.uint64), np.array(scores, dtype=np.float64)

    def free(self):
        if self.ptr:
            self.lib.VecSimQueryReply_Free(self.ptr)
            self.ptr = None

    def __del__(self):
        self.free()


class VecSimIndex:
    """Owned FLAT index handle; methods are the C entry points with the `VecSimIndex_` prefix dropped."""

    def __init__(self, vtype, dim, metric, multi=False, initial_capacity=0, block_size=1024):
        self.lib = load()
        self.vtype, self.dim, self.metric, self.multi = vtype, dim, metric, multi
        params = flat_params(vtype, dim, metric, multi, initial_capacity, block_size)
        self.ptr = self.lib.VecSimIndex_New(C.byref(params))
        if not self.ptr:
            raise RuntimeError("VecSimIndex_New failed: " + last_error())

    def free(self):
        if getattr(self, "ptr", None):
            self.lib.VecSimIndex_Free(self.ptr)
            self.ptr = None

    __del__ = free

    def _q(self, q):
        b = to_blob(q, self.vtype)
        assert b.size == self.dim, "query blob does not match the index dimensionality"
        return b

    # writes
    def add_vector(self, vec, label):
        return self.lib.VecSimIndex_AddVector(self.ptr, _p(self._q(vec)), label)

    def add_bulk(self, mat, first_label=1):
        m = to_blob(mat, self.vtype)
        for i in range(m.shape[0]):
            self.lib.VecSimIndex_AddVector(self.ptr, _p(m[i]), first_label + i)

    def delete_vector(self, label):
        return self.lib.VecSimIndex_DeleteVector(self.ptr, label)

    def reserve(self, rows):
        if self.lib.RSGPU_FlatIndex_Reserve(self.ptr, rows) != 0:
            raise RuntimeError(last_error())

    def add_device_rows(self, dev_ptr, n, first_label=1):
        r = self.lib.RSGPU_FlatIndex_AddDeviceRows(self.ptr, dev_ptr, n, first_label)
        if r < 0:
            raise RuntimeError(last_error())
        return r

    def add_philox_rows(self, seed, first_index, n, first_label=1):
        r = self.lib.RSGPU_FlatIndex_AddPhiloxRows(self.ptr, seed, first_index, n, first_label)
        if r < 0:
            raise RuntimeError(last_error())
        return r

    def label_table(self):
        """0 identity labels, 1 direct device table, 2 device hash table (labels far apart; round 6): RSGPU_FlatIndex_LabelTable"""
        return self.lib.RSGPU_FlatIndex_LabelTable(self.ptr)

    def read_rows(self, row_begin, n):
        out = np.zeros((n, self.dim), dtype=TYPE_NP[self.vtype])
        if self.lib.RSGPU_FlatIndex_ReadRows(self.ptr, row_begin, n, _p(out)) != 0:
            raise RuntimeError(last_error())
        return out

    # info
    def index_size(self):
        return self.lib.VecSimIndex_IndexSize(self.ptr)

    def basic_info(self):
        return self.lib.VecSimIndex_BasicInfo(self.ptr)

    def stats_info(self):
        return self.lib.VecSimIndex_StatsInfo(self.ptr)

    def debug_info(self):
        it = self.lib.VecSimIndex_DebugInfoIterator(self.ptr)
        out = []
        n = self.lib.VecSimDebugInfoIterator_NumberOfFields(it)
        while self.lib.VecSimDebugInfoIterator_HasNextField(it):
            f = self.lib.VecSimDebugInfoIterator_NextField(it).contents
            v = f.fieldValue.stringValue.decode() if f.fieldType == 0 else int(f.fieldValue.uintegerValue)
            out += [f.fieldName.decode(), v]
        self.lib.VecSimDebugInfoIterator_Free(it)
        assert len(out) == 2 * n
        return out

    # queries
    def resolve_params(self, raw, query_type):
        arr = (VecSimRawParam * max(len(raw), 1))()
        keep = []
        for i, (k, v) in enumerate(raw):
            kb, vb = k.encode(), str(v).encode()
            keep += [kb, vb]
            arr[i] = VecSimRawParam(kb, len(kb), vb, len(vb))
        qp = VecSimQueryParams()
        code = self.lib.VecSimIndex_ResolveParams(self.ptr, arr, len(raw), C.byref(qp), query_type)
        return code, qp

    def topk_query(self, q, k, params=None, order=BY_SCORE):
        r = self.lib.VecSimIndex_TopKQuery(self.ptr, _p(self._q(q)), k, C.byref(params) if params else None, order)
        if not r:
            raise RuntimeError("VecSimIndex_TopKQuery failed: " + last_error())
        return QueryReply(r)

    def range_query(self, q, radius, params=None, order=BY_ID):
        r = self.lib.VecSimIndex_RangeQuery(self.ptr, _p(self._q(q)), radius, C.byref(params) if params else None, order)
        if not r:
            raise RuntimeError("VecSimIndex_RangeQuery failed: " + last_error())
        return QueryReply(r)

    def normalized_query(self, q):
        """The blob hybrid_reader.c:295-305 hands to GetDistanceFrom_Unsafe."""
        n = self.lib.VecSimParams_GetQueryBlobSize(self.vtype, self.dim, self.metric)
        buf = np.zeros(n, dtype=np.uint8)
        raw = self._q(q).view(np.uint8).ravel()
        buf[: raw.size] = raw
        if self.metric == VecSimMetric_Cosine:
            self.lib.VecSim_Normalize(_p(buf), self.dim, self.vtype)
        return buf

    def get_distance_from_unsafe(self, label, blob):
        return self.lib.VecSimIndex_GetDistanceFrom_Unsafe(self.ptr, label, _p(blob))

    def prefer_adhoc_search(self, subset, k, initial_check=True):
        return bool(self.lib.VecSimIndex_PreferAdHocSearch(self.ptr, subset, k, initial_check))

    def batch_iterator(self, q, params=None):
        return BatchIterator(self, q, params)

    def adhoc_ctx(self, q):
        return AdhocBfCtx(self, q)

    def topk_batch(self, queries, k):
        """B queries -> (ids [B,k] uint64, scores [B,k] float64, counts [B])."""
        qm = to_blob(queries, self.vtype)
        assert qm.ndim == 2 and qm.shape[1] == self.dim
        b = qm.shape[0]
        ids = np.zeros((b, k), dtype=np.uint64)
        sc = np.full((b, k), np.inf, dtype=np.float64)
        cnt = np.zeros(b, dtype=np.uint64)
        if self.lib.RSGPU_FlatIndex_TopKBatch(self.ptr, _p(qm), b, k, _p(ids), _p(sc), _p(cnt)) != 0:
            raise RuntimeError(last_error())
        return ids, sc, cnt.astype(np.int64)

    def topk_device(self, q, k, dev_scores_ptr, dev_labels_ptr):
        r = self.lib.RSGPU_FlatIndex_TopKDevice(self.ptr, _p(self._q(q)), k, dev_scores_ptr, dev_labels_ptr)
        if r < 0:
            raise RuntimeError(last_error())
        return r


class _ShardView(VecSimIndex):
    """Borrowed handle of one shard of a ShardedIndex (never freed on its own)."""

    def __init__(self, parent, ptr):
        self.lib, self.ptr, self._parent = parent.lib, ptr, parent
        self.vtype, self.dim, self.metric, self.multi = parent.vtype, parent.dim, parent.metric, parent.multi

    def free(self):
        self.ptr = None

    __del__ = free


class ShardedIndex:
    """RSGPU_ShardedIndex_*: one FLAT index row-partitioned over several GPUs (or replicas) of this process."""

    def __init__(self, vtype, dim, metric, n_shards, devices=None, replicas=False, multi=False, block_size=1024):
        self.lib = load()
        self.vtype, self.dim, self.metric, self.multi = vtype, dim, metric, multi
        params = flat_params(vtype, dim, metric, multi, 0, block_size)
        dv = (C.c_int * n_shards)(*devices) if devices is not None else None
        self.ptr = self.lib.RSGPU_ShardedIndex_New(C.byref(params), n_shards, dv, int(replicas))
        if not self.ptr:
            raise RuntimeError("RSGPU_ShardedIndex_New failed: " + last_error())

    def free(self):
        if getattr(self, "ptr", None):
            self.lib.RSGPU_ShardedIndex_Free(self.ptr)
            self.ptr = None

    __del__ = free

    _q = VecSimIndex._q
    normalized_query = VecSimIndex.normalized_query

    def num_shards(self):
        return self.lib.RSGPU_ShardedIndex_NumShards(self.ptr)

    def shard_device(self, i):
        return self.lib.RSGPU_ShardedIndex_ShardDevice(self.ptr, i)

    def shard(self, i):
        p = self.lib.RSGPU_ShardedIndex_Shard(self.ptr, i)
        if not p:
            raise IndexError(i)
        return _ShardView(self, p)

    def index_size(self):
        return self.lib.RSGPU_ShardedIndex_IndexSize(self.ptr)

    def add_vector(self, vec, label):
        return self.lib.RSGPU_ShardedIndex_AddVector(self.ptr, _p(self._q(vec)), label)

    def delete_vector(self, label):
        return self.lib.RSGPU_ShardedIndex_DeleteVector(self.ptr, label)

    def get_distance_from_unsafe(self, label, blob):
        return self.lib.RSGPU_ShardedIndex_GetDistanceFrom(self.ptr, label, _p(blob))

    def topk_query(self, q, k, params=None, order=BY_SCORE):
        r = self.lib.RSGPU_ShardedIndex_TopKQuery(self.ptr, _p(self._q(q)), k, C.byref(params) if params else None, order)
        if not r:
            raise RuntimeError("RSGPU_ShardedIndex_TopKQuery failed: " + last_error())
        return QueryReply(r)

    def range_query(self, q, radius, params=None, order=BY_ID):
        r = self.lib.RSGPU_ShardedIndex_RangeQuery(self.ptr, _p(self._q(q)), radius, C.byref(params) if params else None, order)
        if not r:
            raise RuntimeError("RSGPU_ShardedIndex_RangeQuery failed: " + last_error())
        return QueryReply(r)


class BatchIterator:
    def __init__(self, index, q, params=None):
        self.index, self.lib = index, index.lib
        self._params = params
        self.ptr = self.lib.VecSimBatchIterator_New(index.ptr, _p(index._q(q)), C.byref(params) if params else None)
        if not self.ptr:
            raise RuntimeError("VecSimBatchIterator_New failed: " + last_error())

    def has_next(self):
        return bool(self.lib.VecSimBatchIterator_HasNext(self.ptr))

    def next(self, n, order=BY_ID):
        r = self.lib.VecSimBatchIterator_Next(self.ptr, n, order)
        if not r:
            raise RuntimeError("VecSimBatchIterator_Next failed: " + last_error())
        return QueryReply(r)

    def reset(self):
        self.lib.VecSimBatchIterator_Reset(self.ptr)

    def free(self):
        if getattr(self, "ptr", None):
            self.lib.VecSimBatchIterator_Free(self.ptr)
            self.ptr = None

    __del__ = free


class AdhocBfCtx:
    def __init__(self, index, q):
        self.index, self.lib = index, index.lib
        self.ptr = self.lib.VecSimIndex_AdhocBfCtx_New(index.ptr, _p(index._q(q)))
        if not self.ptr:
            raise RuntimeError("VecSimIndex_AdhocBfCtx_New failed: " + last_error())

    def get_distance_from(self, label):
        return self.lib.VecSimIndex_AdhocBfCtx_GetDistanceFrom(self.ptr, label)

    def get_exact_distances(self, labels):
        lab = np.ascontiguousarray(labels, dtype=np.uint64)
        out = np.zeros(lab.size, dtype=np.float64)
        self.lib.VecSimIndex_AdhocBfCtx_GetExactDistances(self.ptr, _p(lab), _p(out), lab.size)
        return out

    def free(self):
        if getattr(self, "ptr", None):
            self.lib.VecSimIndex_AdhocBfCtx_Free(self.ptr)
            self.ptr = None

    __del__ = free


def set_timeout_callback(fn):
    """fn(ctx) -> int; keep the returned object alive while installed."""
    cb = TIMEOUT_CB(fn) if fn else C.cast(None, TIMEOUT_CB)
    load().VecSim_SetTimeoutCallbackFunction(cb)
    return cb


def set_log_callback(fn):
    cb = LOG_CB(fn) if fn else C.cast(None, LOG_CB)
    load().VecSim_SetLogCallbackFunction(cb)
    return cb


def last_scan_kernel():
    buf = C.create_string_buffer(256)
    return load().RSGPU_GetLastScanKernel(buf, 256).decode()


def coalesce_stats(reset=False):
    """RSGPU_GetCoalesceStats as a dict."""
    a = (C.c_uint64 * 8)()
    load().RSGPU_GetCoalesceStats(a)
    w = (C.c_uint64 * 2)()
    load().RSGPU_GetWidePassStats(w)
    left = int(load().RSGPU_GetCoalesceTimeouts())
    if reset:
        load().RSGPU_ResetCoalesceStats()
    names = ("passes", "queries", "mq_passes", "mq_queries", "lingers", "linger_ns", "mq_device_ns", "mq_redo")
    out = {n: int(a[i]) for i, n in enumerate(names)}
    out["wide_passes"], out["wide_queries"] = int(w[0]), int(w[1])
    out["left_queue_on_timeout"] = left
    return out


def last_mq_scan_kernel():
    buf = C.create_string_buffer(256)
    return load().RSGPU_GetLastMqScanKernel(buf, 256).decode()


TWO_STAGE_STAT_NAMES = ("attempts", "two_stage", "fallback_shape", "fallback_query_or_band_not_finite",
                        "fallback_first_pass_overflow", "fallback_band_overflow", "fallback_fewer_than_k_in_band",
                        "fallback_final_select")


def two_stage_stats(reset=False):
    """RSGPU_GetTwoStageStats as a dict (+ "fallbacks" = every way out to the plain fp32 scan)."""
    a = (C.c_uint64 * 8)()
    load().RSGPU_GetTwoStageStats(a)
    if reset:
        load().RSGPU_ResetTwoStageStats()
    d = {n: int(a[i]) for i, n in enumerate(TWO_STAGE_STAT_NAMES)}
    d["fallbacks"] = d["attempts"] - d["two_stage"]
    return d


def scan_profile():
    n, ms, by = C.c_uint64(0), _dbl(0), C.c_uint64(0)
    load().RSGPU_GetScanProfile(C.byref(n), C.byref(ms), C.byref(by))
    return n.value, ms.value, by.value
