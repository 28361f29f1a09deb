"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the headers
declare, fails loudly without a GPU, and its pure-host entry points behave as the reference pins.
No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from redisearch_amd import build as B
from redisearch_amd import vecsim as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(B.lib_path()):
        B.build()
    return V.load()


def declared_symbols():
    names = set()
    for h in ("include/VecSim/vec_sim.h", "include/VecSim/query_results.h", "include/VecSim/info_iterator.h",
              "include/VecSim/vec_sim_debug.h", "include/rsgpu_ext.h", "include/rsgpu_search.h"):
        p = os.path.join(ROOT, h)
        if not os.path.exists(p):
            continue
        src = re.sub(r"/\*.*?\*/", "", open(p).read(), flags=re.S)
        names |= set(re.findall(r"\b((?:VecSim|RSGPU)\w*)\s*\(", src))
    return names


def test_every_declared_symbol_is_exported(lib):
    missing = [n for n in sorted(declared_symbols()) if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_covers_headers():
    from redisearch_amd import search as S
    assert declared_symbols() <= set(V.ABI) | set(S.ABI)


def test_idf_host_entry_points(lib):
    # reference src/redisearch_rs/idf/tests/tests.rs:23-80 through the C ABI (pure host functions)
    from redisearch_amd import search as S
    assert S.calculate_idf(100, 10) == 3.0 and S.calculate_idf(100, 0) == S.calculate_idf(100, 1)
    assert S.calculate_idf(0, 1) == 1.0 and S.calculate_idf(0, 0) == 1.0 and S.calculate_idf(1, 1) == 1.0
    assert S.calculate_idf(1000, 1) == 9.0 and S.calculate_idf(1000, 500) == 1.0 and S.calculate_idf(1000, 1000) == 1.0
    assert abs(S.calculate_idf_bm25(5, 10) - S.calculate_idf_bm25(10, 10)) < 1e-15
    assert S.calculate_idf_bm25(1000, 1) > S.calculate_idf_bm25(1000, 500)


def test_struct_layout_matches_c(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "VecSim/vec_sim.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(VecSimParams), sizeof(VecSimQueryParams),'
                   'offsetof(VecSimParams, logCtx), offsetof(VecSimQueryParams, batchSize),'
                   'offsetof(VecSimQueryParams, timeoutCtx), sizeof(VecSimIndexBasicInfo), sizeof(VecSim_InfoField));}')
    exe = tmp_path / "sz"
    os.system("gcc -I%s/include %s -o %s" % (ROOT, src, exe))
    got = [int(x) for x in os.popen(str(exe)).read().split()]
    assert got == [C.sizeof(V.VecSimParams), C.sizeof(V.VecSimQueryParams), V.VecSimParams.logCtx.offset,
                   V.VecSimQueryParams.batchSize.offset, V.VecSimQueryParams.timeoutCtx.offset,
                   C.sizeof(V.VecSimIndexBasicInfo), C.sizeof(V.VecSim_InfoField)]


def test_headers_compile_as_c_and_cpp(tmp_path):
    for comp, ext in (("gcc -std=c99", "c"), ("g++ -std=c++11", "cpp")):
        f = tmp_path / ("t." + ext)
        f.write_text('#include "VecSim/vec_sim.h"\n#include "rsgpu_ext.h"\nint main(void){return 0;}\n')
        assert os.system("%s -Wall -Werror -I%s/include -c %s -o %s.o" % (comp, ROOT, f, f)) == 0


def test_enum_values_are_the_rdb_contract():
    # reference src/vector_index.c:486-495 persists algo/type/metric as unsigned ints
    assert (V.VecSimType_FLOAT32, V.VecSimType_FLOAT64, V.VecSimType_BFLOAT16, V.VecSimType_FLOAT16) == (0, 1, 2, 3)
    assert (V.VecSimAlgo_BF, V.VecSimAlgo_HNSWLIB, V.VecSimAlgo_TIERED, V.VecSimAlgo_SVS) == (0, 1, 2, 3)
    assert (V.VecSimMetric_L2, V.VecSimMetric_IP, V.VecSimMetric_Cosine) == (0, 1, 2)
    assert (V.BY_SCORE, V.BY_ID) == (0, 1)


def test_null_safety(lib):
    # reference src/iterators/hybrid_reader.c:543-544,579-580 frees never-assigned fields
    lib.VecSimQueryReply_Free(None)
    lib.VecSimQueryReply_IteratorFree(None)
    lib.VecSimBatchIterator_Free(None)
    lib.VecSimIndex_Free(None)
    lib.VecSimIndex_AdhocBfCtx_Free(None)
    assert lib.VecSimQueryReply_Len(None) == 0
    assert lib.VecSimIndex_IndexSize(None) == 0


def test_non_flat_algorithms_return_null(lib):
    p = V.flat_params(V.VecSimType_FLOAT32, 4, V.VecSimMetric_L2, algo=V.VecSimAlgo_HNSWLIB)
    assert not lib.VecSimIndex_New(C.byref(p))
    assert "FLAT" in V.last_error()
    assert not lib.VecSimIndex_NewDisk(None)


def test_blob_size_and_normalize(lib):
    # reference src/iterators/hybrid_reader.c:298-304
    assert lib.VecSimParams_GetQueryBlobSize(V.VecSimType_FLOAT32, 128, V.VecSimMetric_Cosine) == 512
    assert lib.VecSimParams_GetQueryBlobSize(V.VecSimType_FLOAT16, 128, V.VecSimMetric_L2) == 256
    assert lib.VecSimParams_GetQueryBlobSize(V.VecSimType_INT8, 128, V.VecSimMetric_Cosine) == 132
    assert lib.VecSimParams_GetQueryBlobSize(V.VecSimType_UINT8, 128, V.VecSimMetric_IP) == 128
    v = np.array([3.0, 4.0], dtype=np.float32)
    lib.VecSim_Normalize(V._p(v), 2, V.VecSimType_FLOAT32)
    assert np.allclose(v, [0.6, 0.8], atol=1e-7)
    h = np.array([3.0, 4.0], dtype=np.float16)
    lib.VecSim_Normalize(V._p(h), 2, V.VecSimType_FLOAT16)
    assert np.allclose(h.astype(np.float32), [0.6, 0.8], atol=1e-3)
    d = np.array([0.0, 5.0, 12.0], dtype=np.float64)
    lib.VecSim_Normalize(V._p(d), 3, V.VecSimType_FLOAT64)
    assert np.allclose(d, [0, 5 / 13, 12 / 13], atol=1e-15)


def resolve(lib, raw, qtype):
    arr = (V.VecSimRawParam * max(len(raw), 1))()
    keep = []
    for i, (k, v) in enumerate(raw):
        kb, vb = k.encode(), str(v).encode()
        keep += [kb, vb]
        arr[i] = V.VecSimRawParam(kb, len(kb), vb, len(vb))
    qp = V.VecSimQueryParams()
    return lib.VecSimIndex_ResolveParams(None, arr, len(raw), C.byref(qp), qtype), qp


def test_resolve_params_flat_pins(lib):
    # reference tests/pytests/test_vecsim.py:692-765 (FLAT rows) -> VecSimResolveCode via vector_index.c:749-786
    H, K, R = V.QUERY_TYPE_HYBRID, V.QUERY_TYPE_KNN, V.QUERY_TYPE_RANGE
    assert resolve(lib, [("EF_RUNTIME", 30)], K)[0] == V.VecSimParamResolverErr_UnknownParam          # :707
    assert resolve(lib, [("RERANK", "TRUE")], K)[0] == V.VecSimParamResolverErr_UnknownParam          # :704
    assert resolve(lib, [("EF_FUNTIME", 30)], K)[0] == V.VecSimParamResolverErr_UnknownParam          # :695
    assert resolve(lib, [("BATCH_SIZE", 100)], K)[0] == V.VecSimParamResolverErr_InvalidPolicy_NHybrid  # :710
    assert resolve(lib, [("HYBRID_POLICY", "ADHOC_BF")], K)[0] == V.VecSimParamResolverErr_InvalidPolicy_NHybrid
    assert resolve(lib, [("HYBRID_POLICY", "BATCHES")], R)[0] == V.VecSimParamResolverErr_InvalidPolicy_NHybrid  # :750
    for bad in (0, -6, "34_not_a_number"):                                                             # :716-718
        assert resolve(lib, [("BATCH_SIZE", bad)], H)[0] == V.VecSimParamResolverErr_BadValue
    assert resolve(lib, [("BATCH_SIZE", 8), ("BATCH_SIZE", 0)], H)[0] == V.VecSimParamResolverErr_AlreadySet  # :719
    assert resolve(lib, [("HYBRID_POLICY", "bad_policy")], H)[0] == V.VecSimParamResolverErr_InvalidPolicy_NExits
    assert resolve(lib, [("HYBRID_POLICY", "ADHOC_BF"), ("BATCH_SIZE", 100)], H)[0] == \
        V.VecSimParamResolverErr_InvalidPolicy_AdHoc_With_BatchSize                                    # :724
    assert resolve(lib, [("EPSILON", 2.71828)], K)[0] == V.VecSimParamResolverErr_InvalidPolicy_NRange  # :760
    assert resolve(lib, [("EPSILON", 0.1)], H)[0] == V.VecSimParamResolverErr_InvalidPolicy_NRange     # :762
    assert resolve(lib, [("epsilon", 0.1)], R)[0] == V.VecSimParamResolverErr_UnknownParam             # :765 (FLAT)
    code, qp = resolve(lib, [("hybrid_policy", "batches"), ("batch_size", 10)], H)                     # :955
    assert code == V.VecSim_OK and qp.searchMode == V.HYBRID_BATCHES and qp.batchSize == 10
    code, qp = resolve(lib, [("HYBRID_POLICY", "ADHOC_BF")], H)
    assert code == V.VecSim_OK and qp.searchMode == V.HYBRID_ADHOC_BF
    assert resolve(lib, [], K)[0] == V.VecSim_OK


def test_estimates(lib):
    p = V.flat_params(V.VecSimType_FLOAT32, 768, V.VecSimMetric_Cosine)
    assert lib.VecSimIndex_EstimateElementSize(C.byref(p)) >= 768 * 4   # reference src/spec.c:613
    assert lib.VecSimIndex_EstimateInitialSize(C.byref(p)) > 0          # reference src/spec.c:623


def test_fails_loudly_without_gpu(lib, has_gpu):
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device|No HIP|device"):
        V.VecSimIndex(V.VecSimType_FLOAT32, 4, V.VecSimMetric_L2)


def test_search_seam_argument_errors_are_loud_and_exception_free(lib):
    """Host-side validation of include/rsgpu_search.h runs before any device work: -1 / NULL plus RSGPU_LastError(),
    never an exception, never a crash (SURVEY.md 8b "Errors")."""
    from redisearch_amd import search as S
    sl = S.load()
    ids = (C.c_uint64 * 3)(1, 2, 3)
    sc = (C.c_double * 3)(0.3, 0.2, 0.1)
    w = (C.c_double * 2)(0.5, 0.5)
    out_i, out_s = (C.c_uint64 * 8)(), (C.c_double * 8)()
    fuse = lambda scoring, weights, a_ids, n_a, window: sl.RSGPU_HybridFuse(
        scoring, 60.0, weights, -1, a_ids, C.cast(sc, C.c_void_p), n_a, C.cast(ids, C.c_void_p), C.cast(sc, C.c_void_p), 3,
        window, 8, C.cast(out_i, C.c_void_p), C.cast(out_s, C.c_void_p))
    assert fuse(9, None, C.cast(ids, C.c_void_p), 3, 20) == -1 and "unknown scoring" in V.last_error()
    assert fuse(S.LINEAR, None, C.cast(ids, C.c_void_p), 3, 20) == -1 and "weights" in V.last_error()
    assert fuse(S.RRF, None, C.cast(ids, C.c_void_p), 3, 5000) == -1 and "window" in V.last_error()
    assert fuse(S.RRF, None, None, 3, 20) == -1 and "NULL list" in V.last_error()
    assert fuse(S.LINEAR, C.cast(w, C.c_void_p), None, 0, 0) == 0          # nothing to fuse: 0 results, no device needed
    assert not sl.RSGPU_Intersect(None, 2) and "lists" in V.last_error()
    assert not sl.RSGPU_Union(None, 0)
    lists33 = (C.c_void_p * 33)()
    assert not sl.RSGPU_Intersect(C.cast(lists33, C.c_void_p), 33) and "1..32" in V.last_error()
    assert not sl.RSGPU_Not(None, None, 10) and "NULL child" in V.last_error()
    assert sl.RSGPU_Hits_TopN(None, 5, None, None) == -1
    assert sl.RSGPU_Hits_Len(None) == 0 and sl.RSGPU_Postings_NumEntries(None) == 0
    sl.RSGPU_Hits_Free(None)
    sl.RSGPU_Postings_Free(None)
    sl.RSGPU_DocTable_Free(None)
    assert lib.RSGPU_SetTuning(b"no_such_knob", 1) == -1 and lib.RSGPU_SetTuning(None, 1) == -1
    # round 4: the hybrid tree query -- an empty tree / a NULL list is refused before any device work, and its knobs exist
    args = S.HybridQueryArgs()
    assert sl.RSGPU_HybridTreeQuery(None, C.byref(args)) == -1 and "empty tree" in V.last_error()
    first = (C.c_size_t * 3)(0, 1, 2)
    nul = (C.c_void_p * 2)()
    tq = S.TreeQuery(S.OP_INTERSECT, 2, C.cast(first, C.c_void_p), None, None, C.cast(nul, C.c_void_p), -1, 0)
    assert sl.RSGPU_HybridTreeQuery(C.byref(tq), C.byref(args)) == -1 and "NULL list" in V.last_error()
    tq.root_op = 7
    assert sl.RSGPU_HybridTreeQuery(C.byref(tq), C.byref(args)) == -1 and "root_op" in V.last_error()
    assert sl.RSGPU_HybridQueryPath() in (0, 1, 2)
    for knob in (b"hybrid_tree_tiles", b"hybrid_force_general", b"prioritize_union_children"):
        assert lib.RSGPU_SetTuning(knob, 0) == 0
    assert lib.RSGPU_SetTuning(b"hybrid_tree_tiles", 1) == 0


def test_posting_upload_refuses_what_cannot_be_a_block_list(lib):
    """RSGPU_Postings_Upload reads the arrays it is given: NULL arrays, a descending byte_offset and a list of 2^32 entries
    are refused with a message before anything is read or any device is touched (so this runs without a GPU)."""
    from redisearch_amd import search as S
    import oracle as O
    sl = S.load()
    up = lambda codec, n, first, last, nent, off, data: sl.RSGPU_Postings_Upload(codec, n, first, last, nent, off, data)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    first, last = np.array([1, 50], np.uint64), np.array([40, 90], np.uint64)
    nent, off, data = np.array([3, 2], np.uint32), np.array([0, 9, 15], np.uint64), np.zeros(15, np.uint8)
    assert not up(99, 2, p(first), p(last), p(nent), p(off), p(data)) and "codec" in V.last_error()
    assert not up(O.C_FREQS_ONLY, 2, None, p(last), p(nent), p(off), p(data)) and "NULL" in V.last_error()
    assert not up(O.C_FREQS_ONLY, 2, p(first), p(last), p(nent), p(off), None) and "bytes are NULL" in V.last_error()
    bad = np.array([0, 9, 5], np.uint64)
    assert not up(O.C_FREQS_ONLY, 2, p(first), p(last), p(nent), p(bad), p(data)) and "ascending" in V.last_error()
    huge = np.array([0xFFFFFFF0, 0x20], np.uint32)
    assert not up(O.C_FREQS_ONLY, 2, p(first), p(last), p(huge), p(off), p(data)) and "2^32 entries" in V.last_error()
    h = up(O.C_FREQS_ONLY, 2, p(first), p(last), p(nent), p(off), p(data))     # well-formed: only the device can be missing
    if h:
        sl.RSGPU_Postings_Free(h)
    else:
        assert "device" in V.last_error().lower() or "hip" in V.last_error().lower()
