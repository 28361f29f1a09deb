"""GPU parity tests of the remaining FLAT element types -- FLOAT64 (fp64 distances, 64-bit keys), INT8 and
UINT8 (exact integer sums; cosine divides by the two norms) -- through the VecSim C ABI against the CPU
oracle.  The reference runs its sanity tests over every data type (tests/pytests/common.py:33
VECSIM_DATA_TYPES incl. FLOAT64, tolerance 1e-9: test_vecsim.py:14); INT8/UINT8 are VecSim element types
RediSearch accepts in FT.CREATE (src/vector_index.c:292-318).
"""
import math

import numpy as np
import pytest
import torch

import oracle as O
from redisearch_amd import vecsim as V
from tests.util import TYPE_TO_ORACLE, build_pair, quantize

pytestmark = pytest.mark.gpu

F64, I8, U8 = V.VecSimType_FLOAT64, V.VecSimType_INT8, V.VecSimType_UINT8
L2, IP, COS = V.VecSimMetric_L2, V.VecSimMetric_IP, V.VecSimMetric_Cosine
# FLOAT64: the reference's own tolerance; integer types: L2/IP are exact integers, cosine is one fp32 divide
TOL = {F64: 1e-9, I8: 1e-6, U8: 1e-6}


def near(a, b, vtype):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= TOL[vtype] * np.maximum(1.0, np.abs(b)))


def random_rows(rng, n, dim, vtype):
    return quantize(rng.uniform(-1, 1, (n, dim)), vtype)


def test_sanity_cosine_and_l2_float64():
    # test_vecsim.py:65-212 at FLOAT64 / EPSILONS 1e-9, incl. delete-then-requery and range
    from scipy.spatial import distance as sdist
    vecs = [[0.1, 0.1], [0.1, 0.2], [0.1, 0.3], [0.1, 0.4]]
    q = np.array([0.1, 0.1])
    for metric, fn in ((COS, sdist.cosine), (L2, sdist.sqeuclidean)):
        idx = V.VecSimIndex(F64, 2, metric)
        for i, v in enumerate(vecs):
            idx.add_vector(np.array(v), i + 1)
        ids, sc = idx.topk_query(q, 4).results()
        assert ids.tolist() == [1, 2, 3, 4]
        for i, s in zip(ids, sc):
            assert abs(s - fn(np.array(vecs[i - 1]), q)) <= 1e-9
        rid, _ = idx.range_query(q, fn(np.array([0.1, 0.4]), q) + 1e-9, order=V.BY_ID).results()
        assert rid.tolist() == [1, 2, 3, 4]
        assert idx.delete_vector(1) == 1 and idx.delete_vector(1) == 0
        ids, _ = idx.topk_query(q, 4).results()
        assert ids.tolist() == [2, 3, 4] and idx.index_size() == 3


def test_float64_keeps_what_fp32_cannot():
    # test_vecsim.py:1428-1487: doc i=[i/N,1,1,1], q=[1]^4 -- FLOAT64 must return ids N..N-9 EXACTLY
    n = 6000
    idx = V.VecSimIndex(F64, 4, COS)
    rows = np.ones((n, 4))
    rows[:, 0] = np.arange(1, n + 1) / n
    t = torch.from_numpy(rows).cuda()
    torch.cuda.synchronize()
    idx.add_device_rows(t.data_ptr(), n, 1)
    ids, sc = idx.topk_query(np.ones(4), 10).results()
    assert ids.tolist() == list(range(n, n - 10, -1))
    assert np.all(np.diff(sc) > 0) and sc[0] < 1e-15 and sc[9] < 1e-6   # distances fp32 would round to 0


@pytest.mark.parametrize("vtype", [I8, U8])
def test_integer_types_kat(vtype):
    # L2 = dim * d^2 exactly, IP = 1 - dot, cosine of parallel vectors = 0
    dim = 20                                                            # not a multiple of 16: padded chunk
    idx = V.VecSimIndex(vtype, dim, L2)
    for i in range(1, 101):
        idx.add_vector(np.full(dim, i), i)
    ids, sc = idx.topk_query(np.full(dim, 100), 10).results()
    assert ids.tolist() == list(range(100, 90, -1)) and sc.tolist() == [float(dim * d * d) for d in range(10)]
    ids, sc = idx.range_query(np.full(dim, 100), dim * 9.0, order=V.BY_ID).results()
    assert ids.tolist() == [97, 98, 99, 100] and sc.tolist() == [dim * 9.0, dim * 4.0, dim * 1.0, 0.0]
    ip = V.VecSimIndex(vtype, dim, IP)
    for i in range(1, 11):
        ip.add_vector(np.full(dim, i), i)
    ids, sc = ip.topk_query(np.full(dim, 3), 3).results()
    assert ids.tolist() == [10, 9, 8] and sc.tolist() == [1.0 - dim * 3 * i for i in (10, 9, 8)]
    cs = V.VecSimIndex(vtype, dim, COS)
    v = np.arange(1, dim + 1)
    cs.add_vector(v, 1)
    cs.add_vector(v[::-1].copy(), 2)
    cs.add_vector(3 * v, 3)
    ids, sc = cs.topk_query(2 * v, 3).results()
    assert set(ids[:2].tolist()) == {1, 3} and ids[2] == 2 and abs(sc[0]) <= 1e-6 and abs(sc[1]) <= 1e-6
    expect = 1.0 - float(v @ v[::-1]) / float(v @ v)
    assert abs(sc[2] - expect) <= 1e-6
    # query blob size carries the norm slot for cosine (hybrid_reader.c:298-301)
    assert V.load().VecSimParams_GetQueryBlobSize(vtype, dim, COS) == dim + 4
    nq = cs.normalized_query(2 * v)
    assert abs(cs.get_distance_from_unsafe(2, nq) - expect) <= 1e-6
    assert math.isnan(cs.get_distance_from_unsafe(99, nq))


@pytest.mark.parametrize("vtype", [F64, I8, U8])
@pytest.mark.parametrize("metric", [L2, IP, COS])
@pytest.mark.parametrize("dim,n", [(1, 70), (3, 257), (16, 1000), (33, 900), (128, 2000), (768, 600), (1100, 300),
                                   (5000, 40)])
def test_topk_range_parity_random(vtype, metric, dim, n):
    rng = np.random.default_rng(dim * 31 + n + metric)
    data = random_rows(rng, n, dim, vtype)
    if metric == COS:
        data[np.all(data == 0, axis=1)] = 1                              # zero rows have no direction
    g = V.VecSimIndex(vtype, dim, metric)
    o = O.FlatIndex(TYPE_TO_ORACLE[vtype], dim, metric)
    t = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    torch.cuda.synchronize()
    g.add_device_rows(t.data_ptr(), n, 1)
    o.add_bulk(data, 1)
    for qi in range(3):
        q = random_rows(rng, 1, dim, vtype)[0]
        if metric == COS and not np.any(q):
            q[0] = 1
        k = min(n, (1, 10, 64)[qi])
        gi, gs = g.topk_query(q, k).results()
        oi, os_ = o.topk(q, k)
        assert near(gs, os_, vtype)
        if gi.tolist() != oi.tolist():                                   # only rounding-level near-ties may swap
            od = dict(zip(oi.tolist(), os_.tolist()))
            nq = o.normalized_query(q)
            for i in set(gi.tolist()) ^ set(oi.tolist()):
                assert near(o.distance_from(int(i), nq), os_[-1], vtype)
            assert vtype == F64 or metric == COS                        # integer L2/IP sums are exact: ids identical
        if vtype != F64 and metric != COS:
            assert gs.tolist() == os_.tolist()
        radius = float(os_[-1])
        ri, rs = g.range_query(q, radius, order=V.BY_ID).results()
        oi2, os2 = o.range(q, radius, order=O.BY_ID)
        if vtype != F64 and metric != COS:
            assert ri.tolist() == oi2.tolist() and rs.tolist() == os2.tolist()
        else:
            assert abs(len(ri) - len(oi2)) <= 2 and set(oi.tolist()[:-2]) <= set(ri.tolist())


@pytest.mark.parametrize("vtype", [F64, I8, U8])
def test_batch_iterator_adhoc_and_delete(vtype):
    rng = np.random.default_rng(5)
    n, dim = 700, 24
    data = random_rows(rng, n, dim, vtype)
    g, o = build_pair(vtype, dim, L2, data)
    q = random_rows(rng, 1, dim, vtype)[0]
    git, oit = g.batch_iterator(q), o.batches(q)
    seen = []
    while git.has_next():
        assert oit.has_next()
        gi, gs = git.next(97, V.BY_ID).results()
        oi, os_ = oit.next(97, O.BY_ID)
        assert gi.tolist() == sorted(gi.tolist()) and near(np.sort(gs), np.sort(os_), vtype)
        if vtype != F64:
            assert gi.tolist() == oi.tolist()
        seen += gi.tolist()
    assert sorted(seen) == list(range(1, n + 1))
    nq = g.normalized_query(q)
    labels = [1, 5, 699, 700, 12345]
    got = [g.get_distance_from_unsafe(l, nq) for l in labels]
    want = [o.distance_from(l, o.normalized_query(q)) for l in labels]
    assert math.isnan(got[-1]) and math.isnan(want[-1]) and near(got[:-1], want[:-1], vtype)
    for lab in (700, 3, 350):
        assert g.delete_vector(lab) == o.delete(lab) == 1
    gi, gs = g.topk_query(q, 20).results()
    oi, os_ = o.topk(q, 20)
    assert near(gs, os_, vtype) and (vtype == F64 or gi.tolist() == oi.tolist())


def test_float64_many_rows_device_path():
    # enough rows for the 12-level radix select over 64-bit keys and every scan-grid block
    dev = torch.device("cuda", 0)
    n, dim, k = 300_000, 48, 25
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    x = torch.rand((n, dim), device=dev, dtype=torch.float64, generator=gen) * 2 - 1
    idx = V.VecSimIndex(F64, dim, L2)
    torch.cuda.synchronize()
    idx.add_device_rows(x.data_ptr(), n, 1)
    q = np.random.default_rng(8).uniform(-1, 1, dim)
    ids, sc = idx.topk_query(q, k).results()
    d = ((x - torch.from_numpy(q).to(dev)) ** 2).sum(1)
    rs, ri = torch.topk(d, k, largest=False)
    assert ids.tolist() == (ri.cpu().numpy() + 1).tolist()
    assert np.allclose(sc, rs.cpu().numpy(), rtol=1e-12, atol=0)
