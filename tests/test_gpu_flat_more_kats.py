"""GPU twins (through the VecSim C ABI) of tests/test_oracle_flat_more_kats.py: the reference's multi-value FLAT
known answers over every float type (tests/pytests/test_vecsim.py:1903-1991), the INT8/UINT8 cosine ad-hoc self match
(:2649-2692), delete-all / reuse (:246-287), the one-entry index and an absurd K (:1344-1360, :2601-2614: the module
rejects K = 2^59 before VecSim sees it; the library itself must simply clamp)."""
import numpy as np
import pytest

from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu
L2, COS = V.VecSimMetric_L2, V.VecSimMetric_Cosine
FLOAT_TYPES = [V.VecSimType_FLOAT32, V.VecSimType_FLOAT64, V.VecSimType_FLOAT16, V.VecSimType_BFLOAT16]


@pytest.mark.parametrize("vtype", FLOAT_TYPES)
def test_multi_value_json_kat(vtype):
    n, dim, per_doc, scale, k = 100, 4, 5, 8.0, 10
    g = V.VecSimIndex(vtype, dim, L2, multi=True)
    for i in range(n):
        for j in range(per_doc):
            g.add_vector(np.full(dim, (i + j) / scale), i)
    assert g.index_size() == n * per_doc
    ids, sc = g.topk_query(np.zeros(dim), k).results()
    assert ids.tolist() == list(range(k))
    assert sc.tolist() == [i * i * dim / (scale * scale) for i in range(k)]
    radius = (dim * k ** 2 + 40) / (scale * scale)
    rid, rsc = g.range_query(np.full(dim, n / scale), radius, order=V.BY_ID).results()
    exp_ids = list(range(n - k - per_doc + 1, n))
    exp_sc = [dim * (n - per_doc - i + 1) ** 2 / (scale * scale) for i in range(n - k - per_doc + 1, n - per_doc + 1)]
    exp_sc += [0.0] * (per_doc - 1)
    assert rid.tolist() == exp_ids and rsc.tolist() == exp_sc
    # batches: every label once, at its best vector, in score order
    it = g.batch_iterator(np.zeros(dim))
    seen = []
    while it.has_next() and len(seen) < 30:
        bi, bs = it.next(7).results()
        seen += list(zip(bi.tolist(), bs.tolist()))
    assert [s[0] for s in seen[:28]] == list(range(28))
    assert [s[1] for s in seen[:28]] == [i * i * dim / (scale * scale) for i in range(28)]


@pytest.mark.parametrize("vtype,limit,npdt", [(V.VecSimType_INT8, 127, np.int8), (V.VecSimType_UINT8, 255, np.uint8)])
def test_int8_uint8_cosine_adhoc_self_match_is_zero(vtype, limit, npdt):
    dim, qty, k = 4, 10, 3
    g = V.VecSimIndex(vtype, dim, COS)
    vecs = [np.array([min(limit, i + j) for j in range(dim)], dtype=npdt) for i in range(1, qty + 1)]
    for i, v in enumerate(vecs):
        g.add_vector(v, i + 1)
    q = vecs[-1]
    ctx = g.adhoc_ctx(q)
    d = ctx.get_exact_distances(np.arange(1, qty + 1))
    assert d[qty - 1] == 0.0 and int(np.argmin(d)) == qty - 1
    nq = g.normalized_query(q)
    assert g.get_distance_from_unsafe(qty, nq) == 0.0
    ids, sc = g.topk_query(q, k).results()
    order = sorted(range(qty), key=lambda i: (d[i], i))[:k]
    assert ids.tolist() == [o + 1 for o in order] and sc[0] == 0.0


def test_delete_all_and_reuse_labels_three_times():
    rng = np.random.default_rng(246)
    g = V.VecSimIndex(V.VecSimType_FLOAT32, 2, L2)
    q = rng.standard_normal(2).astype(np.float32)
    for _ in range(3):
        for l in (1, 2, 3, 4):
            g.delete_vector(l)
        assert g.index_size() == 0 and g.topk_query(q, 4).results()[0].tolist() == []
        vecs = rng.standard_normal((4, 2)).astype(np.float32)
        for l, v in zip((1, 2, 3, 4), vecs):
            assert g.add_vector(v, l) == 1
        ids, sc = g.topk_query(q, 4).results()
        exp = sorted(range(4), key=lambda i: (float(np.float32(((vecs[i] - q) ** 2).sum())), i))
        assert sorted(ids.tolist()) == [1, 2, 3, 4]
        assert np.allclose(sc, [float(((vecs[e] - q) ** 2).sum()) for e in exp], rtol=1e-5, atol=1e-6)


def test_single_entry_and_oversized_k():
    g = V.VecSimIndex(V.VecSimType_FLOAT32, 128, L2)
    v = np.random.default_rng(1344).random(128).astype(np.float32)
    g.add_vector(v, 7)
    ids, sc = g.topk_query(v, 10).results()
    assert ids.tolist() == [7] and sc.tolist() == [0.0]
    ids, _ = g.topk_query(v, 2 ** 59).results()
    assert ids.tolist() == [7]
