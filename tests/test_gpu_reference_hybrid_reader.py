"""The REFERENCE's own HybridIterator on the MI355X engine.

oracle/_ref/libref_hybrid_reader.so is the reference's src/iterators/hybrid_reader.c + src/util/minmax_heap.c compiled
in place (`make -C oracle ref`) with oracle/hybrid_harness.c supplying what is Rust in the module (result constructors,
a sorted-id-list child iterator).  Its VecSim calls bind to redisearch_amd/lib/libVectorSimilarity.so -- so this is
RediSearch's hybrid `(filter)=>[KNN k @v $blob]` iterator, unmodified, driving the GPU index through the seam it uses
in production: mode selection via VecSimIndex_PreferAdHocSearch, the batches loop over VecSimBatchIterator_Next(BY_ID)
with the merge-join and policy review, the ad-hoc loop over VecSimIndex_GetDistanceFrom_Unsafe, the K-bounded heap.
Expectations: the reference's end-to-end known answers (tests/pytests/test_vecsim.py:963-1038, :1362-1396) and, on random
data, the pure-Python replay of the same algorithm over the CPU oracle (tests/hybrid_replay.py).

libref_hybrid_reader_batched.so is the same file with computeDistances_RAM replaced by THIS repository's batching shim
(integration/hybrid_reader_batched.inc.c): identical results, one gather launch per 4096 candidates."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import oracle as O
from redisearch_amd import build as B
from redisearch_amd import vecsim as V
from tests import hybrid_replay as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBS = {"reference": os.path.join(ROOT, "oracle", "_ref", "libref_hybrid_reader.so"),
        "batched": os.path.join(ROOT, "oracle", "_ref", "libref_hybrid_reader_batched.so")}


class Opts(C.Structure):
    _fields_ = [("search_mode_in", C.c_int), ("batch_size", C.c_size_t), ("can_trim", C.c_int), ("read_twice", C.c_int),
                ("child_estimate", C.c_size_t), ("search_mode_out", C.c_int), ("num_iterations", C.c_size_t),
                ("max_batch_size", C.c_size_t), ("child_reads", C.c_size_t), ("child_skips", C.c_size_t),
                ("child_rewinds", C.c_size_t), ("timed_out", C.c_int), ("second_pass_identical", C.c_int)]


_loaded = {}


def harness(kind):
    if not os.path.exists(LIBS[kind]):
        pytest.skip("%s not built (needs /root/reference at build time)" % os.path.basename(LIBS[kind]))
    if kind not in _loaded:
        V.load()
        C.CDLL(B.lib_path(), mode=C.RTLD_GLOBAL)      # promote the engine's VecSim symbols: the harness binds to them
        L = C.CDLL(LIBS[kind])
        L.xhr_run.restype = C.c_long
        L.xhr_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                              C.POINTER(Opts), C.c_void_p, C.c_void_p, C.c_size_t]
        _loaded[kind] = L
    return _loaded[kind]


def run(kind, g, q, k, child, policy=0, batch_size=0, can_trim=False, estimate=0, twice=False):
    L = harness(kind)
    blob = V.to_blob(q, g.vtype)
    o = Opts(search_mode_in=policy, batch_size=batch_size, can_trim=int(can_trim), read_twice=int(twice), child_estimate=estimate)
    cap = max(k, 1)
    ids, sc = np.zeros(cap, np.uint64), np.zeros(cap, np.float64)
    if child is None:
        cp, cn = None, 0
    else:
        ch = np.ascontiguousarray(child, dtype=np.uint64)
        keep = np.zeros(1, np.uint64) if ch.size == 0 else ch
        cp, cn = keep.ctypes.data_as(C.c_void_p), ch.size
    n = L.xhr_run(g.ptr, g.vtype, g.metric, g.dim, blob.ctypes.data_as(C.c_void_p), k, cp, cn, C.byref(o),
                  ids.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p), cap)
    assert n >= 0 and not o.timed_out
    return list(zip(ids[:n].tolist(), sc[:n].tolist())), o


def ramp_index(n, dim, vtype, tdt):
    g = V.VecSimIndex(vtype, dim, V.VecSimMetric_L2)
    t = torch.arange(1, n + 1, dtype=tdt, device="cuda")[:, None].repeat(1, dim).contiguous()
    torch.cuda.synchronize()
    g.add_device_rows(t.data_ptr(), n, 1)
    return g


@pytest.mark.parametrize("kind", ["reference", "batched"])
def test_batches_mode_with_text_kats(kind):
    """reference tests/pytests/test_vecsim.py:963-1038 (N = 6000, d = 2, FLOAT64 L2, doc i = [i, i], q = [N, N])."""
    n, dim, k = 6000, 2, 10
    g = ramp_index(n, dim, V.VecSimType_FLOAT64, torch.float64)
    q = np.full(dim, float(n))
    res, o = run(kind, g, q, k, range(1, n + 1))
    assert o.search_mode_out == H.HYBRID_BATCHES and res == [(n - i, float(dim * i * i)) for i in range(k)]
    res, o = run(kind, g, q, k, range(5, n + 1, 5))
    assert res == [(n - 5 * i, float(dim * (5 * i) ** 2)) for i in range(k)] and o.search_mode_out == H.HYBRID_BATCHES
    # an empty intersection whose estimate is 1200: the first batch finds nothing, the policy flips to ad-hoc BF
    res, o = run(kind, g, q, k, [], estimate=1200)
    assert res == [] and o.search_mode_out == H.HYBRID_BATCHES_TO_ADHOC_BF and o.num_iterations == 1
    keep = [i for i in range(1, n + 1) if i % 5]
    res, _ = run(kind, g, q, k, keep)
    assert res == [(n - i, float(dim * i * i)) for i in range(13) if (n - i) % 5][:k]
    res, o = run(kind, g, q, k, None)                       # no child: STANDARD_KNN through VecSimIndex_TopKQuery
    assert o.search_mode_out == H.STANDARD_KNN and res == [(n - i, float(dim * i * i)) for i in range(k)]
    res, o = run(kind, g, q, k, range(1, n + 1), twice=True)  # Rewind + second pass
    assert o.second_pass_identical == 1


@pytest.mark.parametrize("kind", ["reference", "batched"])
def test_adhoc_bf_mode_kat(kind):
    """reference tests/pytests/test_vecsim.py:1362-1396: 100 docs of dim 128, every 10th passes the filter."""
    n, dim, k = 100, 128, 10
    g = ramp_index(n, dim, V.VecSimType_FLOAT32, torch.float32)
    q = np.full(dim, float(n), dtype=np.float32)
    res, o = run(kind, g, q, k, range(10, n + 1, 10), policy=H.HYBRID_ADHOC_BF)
    assert res == [(n - 10 * j, float(dim * (10 * j) ** 2)) for j in range(k)] and o.search_mode_out == H.HYBRID_ADHOC_BF
    assert o.child_reads == 11                       # 10 candidates + the EOF read
    res2, _ = run(kind, g, q, k, range(10, n + 1, 10), policy=H.HYBRID_ADHOC_BF, can_trim=True)
    assert res2 == res                               # Metric results instead of HybridMetric(vector, child)


@pytest.mark.parametrize("metric,om", [(V.VecSimMetric_L2, O.L2), (V.VecSimMetric_Cosine, O.COSINE), (V.VecSimMetric_IP, O.IP)])
def test_reference_iterator_on_gpu_equals_replay_on_oracle(metric, om):
    """Random data, every policy: the reference's compiled iterator over the GPU index == the Python replay of the same
    algorithm over the CPU oracle index; and the batching shim == the per-candidate original, id for id."""
    rng = np.random.default_rng(1583)
    n, dim, k = 9000, 24, 10
    data = rng.standard_normal((n, dim)).astype(np.float32)
    g = V.VecSimIndex(V.VecSimType_FLOAT32, dim, metric)
    t = torch.from_numpy(data).cuda()
    torch.cuda.synchronize()
    g.add_device_rows(t.data_ptr(), n, 1)
    o = O.FlatIndex(O.F32, dim, om)
    o.add_bulk(data)
    for trial in range(4):
        qv = rng.standard_normal(dim).astype(np.float32)
        m = int(rng.choice([40, 700, 5000]))
        child = sorted(rng.choice(np.arange(1, n + 400), m, replace=False).tolist())   # some ids have no vector
        for policy, bs in ((0, 0), (H.HYBRID_BATCHES, 7), (H.HYBRID_BATCHES, 0), (H.HYBRID_ADHOC_BF, 0)):
            ref, ro = run("reference", g, qv, k, child, policy=policy, batch_size=bs)
            bat, bo = run("batched", g, qv, k, child, policy=policy, batch_size=bs)
            rep = H.HybridReplay(H.OracleIndex(o), qv, k, H.IdListChild(child), policy=policy, batch_size=bs)
            want = rep.results()
            assert [i for i, _ in ref] == [i for i, _ in want], (trial, policy, bs)
            assert np.allclose([d for _, d in ref], [d for _, d in want], rtol=1e-5, atol=1e-5)
            assert ro.search_mode_out == rep.mode and ro.num_iterations == rep.num_iterations
            assert bat == ref and bo.search_mode_out == ro.search_mode_out


def test_batching_shim_issues_one_gather_per_chunk():
    """5000 candidates, forced ad-hoc: the original makes 5000 GetDistanceFrom calls, the shim two GetExactDistances
    calls (chunks of 4096) -- observable as wall time; results identical."""
    import time
    rng = np.random.default_rng(7)
    n, dim, k = 20000, 64, 10
    data = rng.standard_normal((n, dim)).astype(np.float32)
    g = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2)
    t = torch.from_numpy(data).cuda()
    torch.cuda.synchronize()
    g.add_device_rows(t.data_ptr(), n, 1)
    child = sorted(rng.choice(np.arange(1, n + 1), 5000, replace=False).tolist())
    qv = rng.standard_normal(dim).astype(np.float32)
    out = {}
    for kind in ("reference", "batched"):
        run(kind, g, qv, k, child, policy=H.HYBRID_ADHOC_BF)
        t0 = time.perf_counter()
        out[kind], _ = run(kind, g, qv, k, child, policy=H.HYBRID_ADHOC_BF)
        out[kind + "_s"] = time.perf_counter() - t0
    assert out["batched"] == out["reference"]
    assert out["batched_s"] * 5 < out["reference_s"], out


# ---- Boundary 3 under the reference's iterator: childIt = the GPU-backed intersection (include/rs_iterator.h) -------------
def run_with_gpu_child(kind, g, q, k, lists, policy=0, batch_size=0, can_trim=False, max_slop=-1, in_order=False):
    """The whole reference pipeline of `(@t:a @t:b)=>[KNN k @v $blob]` with both halves on the device: the reference's
    compiled HybridIterator pulls its filter through librsgpu_iterators.so's vtable and its vectors through the VecSim
    seam.  The harness library supplies the module's RSIndexResult constructors (oracle/hybrid_harness.c)."""
    from redisearch_amd import search as S
    L = harness(kind)
    if not hasattr(L, "_child_ready"):
        L.xhr_run_child.restype = C.c_long
        L.xhr_run_child.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.POINTER(Opts), C.c_void_p, C.c_void_p, C.c_size_t]
        L._child_ready = True
    S.load_iterators(L._handle)          # results are built with THIS harness's constructors
    child = S.new_iterator("and", lists, max_slop=max_slop, in_order=in_order)
    blob = V.to_blob(q, g.vtype)
    o = Opts(search_mode_in=policy, batch_size=batch_size, can_trim=int(can_trim))
    ids, sc = np.zeros(max(k, 1), np.uint64), np.zeros(max(k, 1), np.float64)
    n = L.xhr_run_child(g.ptr, g.vtype, g.metric, g.dim, blob.ctypes.data_as(C.c_void_p), k, child, C.byref(o),
                        ids.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p), max(k, 1))
    assert n >= 0 and not o.timed_out
    return list(zip(ids[:n].tolist(), sc[:n].tolist())), o


@pytest.mark.parametrize("kind", ["reference", "batched"])
def test_reference_hybrid_iterator_over_the_gpu_intersection_iterator(kind):
    from redisearch_amd import search as S
    from tests.test_gpu_proximity import gpu, rand_list
    rng = np.random.default_rng(2024)
    n, dim, k = 9000, 24, 10
    data = rng.standard_normal((n, dim)).astype(np.float32)
    g = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2)
    t = torch.from_numpy(data).cuda()
    torch.cuda.synchronize()
    g.add_device_rows(t.data_ptr(), n, 1)
    try:
        for sizes, slop in (((6000, 5000), -1), ((900, 700), -1), ((6000, 5500), 3)):
            lists = [rand_list(rng, O.C_FULL, m, n + 300, max_pos=15) for m in sizes]     # some ids have no vector
            gl = [gpu(l) for l in lists]
            want_child = (O.intersect_ex(lists, max_slop=slop)[0] if slop >= 0 else O.intersect(lists)[0]).tolist()
            assert len(want_child) > 30
            qv = rng.standard_normal(dim).astype(np.float32)
            for policy, bs, trim in ((0, 0, False), (H.HYBRID_BATCHES, 7, False), (H.HYBRID_ADHOC_BF, 0, True),
                                     (H.HYBRID_BATCHES, 0, True)):
                got, go = run_with_gpu_child(kind, g, qv, k, gl, policy=policy, batch_size=bs, can_trim=trim, max_slop=slop)
                # the same iterator over the harness's own sorted-id-list child holding the oracle's intersection; the
                # GPU child estimates like the reference's intersection does (its smallest list), so hand that over
                ref, ro = run(kind, g, qv, k, want_child, policy=policy, batch_size=bs, can_trim=trim,
                              estimate=min(l.unique_docs for l in lists))
                assert got == ref, (sizes, slop, policy, bs)
                assert (go.search_mode_out, go.num_iterations, go.max_batch_size) == (ro.search_mode_out, ro.num_iterations,
                                                                                      ro.max_batch_size)
    finally:
        S.load_iterators()   # (binding stays; test_gpu_iterators re-binds to its own harness at module start)
