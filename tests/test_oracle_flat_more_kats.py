"""More known-answer tests of the reference for the FLAT path, against the CPU oracle (the GPU twins live in
tests/test_gpu_flat_more_kats.py):
  * multi-value FLAT over every float type -- tests/pytests/test_vecsim.py:1903-1991 (TestIndexMultiValueJsonReload):
    doc i holds the vectors [(i+j)/8]*4, j<5; KNN 10 of 0 -> ids 0..9 with i^2*dim/64; a range query around [n/8]*4
    yields one hit per doc at its CLOSEST vector.  This pins min-over-a-label's-vectors and label de-duplication.
  * INT8 / UINT8 cosine ad-hoc distances -- test_vecsim.py:2649-2692: the self-match distance is EXACTLY 0 and
    the closest three docs come back in order.
  * delete-all / re-insert with the same ids three times -- test_vecsim.py:246-287.
  * one-entry index, K larger than the index -- test_vecsim.py:1344-1360."""
import numpy as np
import pytest

import oracle as O

FLOAT_TYPES = [O.F32, O.F64, O.F16, O.BF16]


def multi_value_corpus(n=100, dim=4, per_doc=5, scale=8.0):
    rows, labels = [], []
    for i in range(n):
        for j in range(per_doc):
            rows.append([(i + j) / scale] * dim)
            labels.append(i)
    return np.array(rows, dtype=np.float64), labels


@pytest.mark.parametrize("vtype", FLOAT_TYPES)
def test_multi_value_json_kat(vtype):
    n, dim, per_doc, scale, k = 100, 4, 5, 8.0, 10
    rows, labels = multi_value_corpus(n, dim, per_doc, scale)
    idx = O.FlatIndex(vtype, dim, O.L2, multi=True)
    for r, l in zip(rows, labels):
        idx.add(r, l)
    ids, sc = idx.topk(np.zeros(dim), k)
    assert ids.tolist() == list(range(k))
    assert sc.tolist() == [i * i * dim / (scale * scale) for i in range(k)]
    radius = (dim * k ** 2 + 40) / (scale * scale)
    rid, rsc = idx.range(np.full(dim, n / scale), radius, order=O.BY_ID)
    exp_ids = list(range(n - k - per_doc + 1, n))
    exp_sc = [dim * (n - per_doc - i + 1) ** 2 / (scale * scale) for i in range(n - k - per_doc + 1, n - per_doc + 1)]
    exp_sc += [0.0] * (per_doc - 1)
    assert rid.tolist() == exp_ids and rsc.tolist() == exp_sc


@pytest.mark.parametrize("vtype,limit,npdt", [(O.I8, 127, np.int8), (O.U8, 255, np.uint8)])
def test_int8_uint8_cosine_adhoc_self_match_is_zero(vtype, limit, npdt):
    dim, qty, k = 4, 10, 3
    idx = O.FlatIndex(vtype, dim, O.COSINE)
    vecs = [np.array([min(limit, i + j) for j in range(dim)], dtype=npdt) for i in range(1, qty + 1)]
    for i, v in enumerate(vecs):
        idx.add(v, i + 1)
    q = vecs[-1]
    nq = idx.normalized_query(q)
    d = [idx.distance_from(l, nq) for l in range(1, qty + 1)]
    order = sorted(range(qty), key=lambda i: (d[i], i))[:k]
    assert order[0] == qty - 1 and d[qty - 1] == 0.0
    ids, sc = idx.topk(q, k)
    assert ids.tolist() == [o + 1 for o in order] and sc[0] == 0.0


def test_delete_all_and_reuse_labels_three_times():
    rng = np.random.default_rng(246)
    idx = O.FlatIndex(O.F32, 2, O.L2)
    q = rng.standard_normal(2).astype(np.float32)
    for _ in range(3):
        for l in (1, 2, 3, 4):
            idx.delete(l)
        assert len(idx) == 0 and idx.topk(q, 4)[0].tolist() == []
        vecs = rng.standard_normal((4, 2)).astype(np.float32)
        for l, v in zip((1, 2, 3, 4), vecs):
            assert idx.add(v, l) == 1
        ids, sc = idx.topk(q, 4)
        exp = sorted(range(4), key=lambda i: (float(np.float32(((vecs[i] - q) ** 2).sum())), i))
        assert sorted(ids.tolist()) == [1, 2, 3, 4] and ids.tolist() == [e + 1 for e in exp]


def test_single_entry_and_oversized_k():
    idx = O.FlatIndex(O.F32, 128, O.L2)
    v = np.random.default_rng(1344).random(128).astype(np.float32)
    idx.add(v, 7)
    ids, sc = idx.topk(v, 10)
    assert ids.tolist() == [7] and sc.tolist() == [0.0]
    ids, _ = idx.topk(v, 2 ** 59)
    assert ids.tolist() == [7]
