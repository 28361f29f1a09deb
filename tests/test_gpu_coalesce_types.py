"""The multi-query scan for the REST of the element types (round 3): INT8 / UINT8 (exact integer sums; IP, L2 and the
two-norm cosine), FLOAT64 (64-bit keys) and multi-value indexes (one key array per query, then the label walk).

Reference shape: B queries = B VecSimIndex_TopKQuery calls (src/iterators/hybrid_reader.c:374) issued from the worker pool
(src/util/workers.c:58,104).  Contract as in test_gpu_coalesce.py: a reply that came out of a shared pass -- through the
coalescer or through RSGPU_FlatIndex_TopKBatch -- is BIT-IDENTICAL (ids and scores) to the reply of the same query issued
alone, and the CPU oracle agrees with both."""
import threading

import numpy as np
import pytest

import oracle as O
from redisearch_amd import vecsim as V
from tests.util import assert_topk_parity

pytestmark = pytest.mark.gpu

I8, U8, F64, F32, F16 = V.VecSimType_INT8, V.VecSimType_UINT8, V.VecSimType_FLOAT64, V.VecSimType_FLOAT32, V.VecSimType_FLOAT16
METRICS = [V.VecSimMetric_L2, V.VecSimMetric_IP, V.VecSimMetric_Cosine]


def _index(vtype, dim, metric, n, seed=5, multi=False):
    idx = V.VecSimIndex(vtype, dim, metric, multi=multi)
    assert idx.add_philox_rows(seed, 0, n, 1) == n
    return idx


def _queries(vtype, dim, nq, seed=6):
    s = V.VecSimIndex(vtype, dim, V.VecSimMetric_L2)
    try:
        assert s.add_philox_rows(seed, 1 << 40, nq, 1) == nq
        return s.read_rows(0, nq)
    finally:
        s.free()


def _same_as_single(idx, qs, k, ids, sc, cnt):
    for i in range(len(qs)):
        si, ss = idx.topk_query(qs[i], k).results()
        assert cnt[i] == len(si)
        assert ids[i][:len(si)].tolist() == si.tolist(), (i, k)
        assert sc[i][:len(si)].tolist() == ss.tolist(), (i, k)


INT_SHAPES = [  # every (G, ITERS) shape of scan_mq_int_kernel, exact and masked: dim bytes per row
    128, 100, 256, 200, 512, 1024, 1000, 768, 1536, 2048, 1900, 3072, 4096,
]


@pytest.mark.parametrize("vtype", [I8, U8])
@pytest.mark.parametrize("dim", INT_SHAPES)
@pytest.mark.parametrize("metric", METRICS)
def test_integer_rows_multi_query_pass_is_bit_identical(vtype, dim, metric):
    n = 70_000
    idx = _index(vtype, dim, metric, n)
    try:
        qs = _queries(vtype, dim, 19)     # passes of 16 (two launches of eight) and 3
        before = V.coalesce_stats()
        for k in (1, 10, 100):
            ids, sc, cnt = idx.topk_batch(qs, k)
            assert (cnt == k).all()
            _same_as_single(idx, qs, k, ids, sc, cnt)
        after = V.coalesce_stats()
        assert after["mq_queries"] - before["mq_queries"] == 3 * 19
    finally:
        idx.free()


F64_SHAPES = [64, 50, 128, 192, 256, 384, 300, 512, 768, 700]


@pytest.mark.parametrize("dim", F64_SHAPES)
@pytest.mark.parametrize("metric", METRICS)
def test_float64_rows_multi_query_pass_is_bit_identical(dim, metric):
    n = 40_000
    idx = _index(F64, dim, metric, n)
    try:
        qs = _queries(F64, dim, 11)       # a pass of 11 = launches of 4, 4 and 3
        before = V.coalesce_stats()
        for k in (1, 10, 100):
            ids, sc, cnt = idx.topk_batch(qs, k)
            assert (cnt == k).all()
            _same_as_single(idx, qs, k, ids, sc, cnt)
        after = V.coalesce_stats()
        assert after["mq_queries"] - before["mq_queries"] == 3 * 11
    finally:
        idx.free()


@pytest.mark.parametrize("vtype,otype,dim", [(I8, O.I8, 768), (U8, O.U8, 512), (F64, O.F64, 384)])
@pytest.mark.parametrize("metric", METRICS)
def test_shared_pass_against_the_oracle(vtype, otype, dim, metric):
    n = 20_000
    rng = np.random.default_rng(dim + metric)
    if vtype == I8:
        data = rng.integers(-128, 128, (n, dim)).astype(np.int8)
        qs = rng.integers(-128, 128, (6, dim)).astype(np.int8)
    elif vtype == U8:
        data = rng.integers(0, 256, (n, dim)).astype(np.uint8)
        qs = rng.integers(0, 256, (6, dim)).astype(np.uint8)
    else:
        data = rng.uniform(-1, 1, (n, dim))
        qs = rng.uniform(-1, 1, (6, dim))
    g = V.VecSimIndex(vtype, dim, metric)
    o = O.FlatIndex(otype, dim, metric)
    try:
        g.add_bulk(data)
        o.add_bulk(data)
        for k in (5, 50):
            ids, sc, cnt = g.topk_batch(qs, k)
            for i, q in enumerate(qs):
                gi, gs = assert_topk_parity(g, o, q, k)
                assert ids[i].tolist() == gi.tolist() and sc[i].tolist() == gs.tolist()
    finally:
        g.free()


@pytest.mark.parametrize("vtype,dim", [(F32, 128), (F32, 768), (F16, 768), (I8, 768), (F64, 128)])
@pytest.mark.parametrize("metric", [V.VecSimMetric_L2, V.VecSimMetric_Cosine])
def test_multi_value_index_shares_passes(vtype, dim, metric):
    """a multi-value index: 60 000 vectors under 20 000 labels; a reply holds every label once, at its closest vector
    (tests/pytests/test_vecsim.py:1903-1991) -- the shared pass walks each query's key array exactly as a single query does"""
    n, labels = 60_000, 20_000
    idx = V.VecSimIndex(vtype, dim, metric, multi=True)
    try:
        for part in range(3):       # label l holds rows l, l + 20 000, l + 40 000
            assert idx.add_philox_rows(9, part * labels, labels, 1) == labels
        assert idx.index_size() == n
        qs = _queries(vtype, dim, 9)
        before = V.coalesce_stats()
        for k in (1, 10, 40):
            ids, sc, cnt = idx.topk_batch(qs, k)
            assert (cnt == k).all()
            for i in range(len(qs)):
                assert len(set(ids[i][:k].tolist())) == k      # one hit per label
            _same_as_single(idx, qs, k, ids, sc, cnt)
        after = V.coalesce_stats()
        assert after["mq_queries"] - before["mq_queries"] == 3 * 9
    finally:
        idx.free()


def test_multi_value_shared_pass_against_the_oracle():
    n, dim = 9_000, 64
    rng = np.random.default_rng(21)
    data = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    labels = rng.integers(1, 2_000, n)
    g = V.VecSimIndex(F32, dim, V.VecSimMetric_L2, multi=True)
    o = O.FlatIndex(O.F32, dim, O.L2, multi=True)
    try:
        for i in range(n):
            g.add_vector(data[i], int(labels[i]))
            o.add(data[i], int(labels[i]))
        qs = rng.uniform(-1, 1, (5, dim)).astype(np.float32)
        ids, sc, cnt = g.topk_batch(qs, 25)
        for i, q in enumerate(qs):
            gi, gs = assert_topk_parity(g, o, q, 25)
            assert ids[i].tolist() == gi.tolist() and sc[i].tolist() == gs.tolist()
    finally:
        g.free()


@pytest.mark.parametrize("vtype,dim,metric", [(I8, 768, V.VecSimMetric_Cosine), (U8, 1024, V.VecSimMetric_L2),
                                              (F64, 256, V.VecSimMetric_IP)])
def test_concurrent_callers_share_passes_on_every_type(vtype, dim, metric):
    """eight threads through VecSimIndex_TopKQuery: the coalescer (corpus >= 64 MiB) puts their queries into shared passes;
    every reply equals the serial reply of the same query, bit for bit"""
    lib = V.load()
    n = 200_000 if vtype != F64 else 80_000
    assert n * dim * (8 if vtype == F64 else 1) >= 64 << 20
    idx = _index(vtype, dim, metric, n)
    try:
        qs = _queries(vtype, dim, 24)
        ks = [1 + (i * 7) % 40 for i in range(len(qs))]
        assert lib.RSGPU_SetTuning(b"coalesce", 0) == 0
        want = [idx.topk_query(q, k).results() for q, k in zip(qs, ks)]
        assert lib.RSGPU_SetTuning(b"coalesce", 1) == 0
        before = V.coalesce_stats()
        errors, barrier = [], threading.Barrier(8)

        def worker(t):
            try:
                barrier.wait()
                for rep in range(3):
                    for i in range(t, len(qs), 8):
                        gi, gs = idx.topk_query(qs[i], ks[i]).results()
                        if gi.tolist() != want[i][0].tolist() or gs.tolist() != want[i][1].tolist():
                            errors.append((t, rep, i))
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))
        th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors[:5]
        after = V.coalesce_stats()
        assert after["mq_passes"] > before["mq_passes"]            # some passes were shared ...
        assert after["mq_queries"] - before["mq_queries"] >= 2 * (after["mq_passes"] - before["mq_passes"])
    finally:
        lib.RSGPU_SetTuning(b"coalesce", 1)
        idx.free()
