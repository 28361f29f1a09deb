"""The reference's HybridIterator (tests/hybrid_replay.py: a faithful replay of src/iterators/hybrid_reader.c, with the
reference's own min-max heap where oracle/_ref provides it) driving the CPU oracle's FLAT index, against the
reference's end-to-end expectations for `(filter)=>[KNN k @v $blob]`:
  test_vecsim.py:963-1038  test_hybrid_query_batches_mode_with_text (N = 6000, d = 2, FLOAT64 L2, doc i = [i, i],
                           q = [N, N]): every doc / every 5th doc / an empty intersection that flips the policy to
                           ad-hoc BF after the first batch / all but every 5th doc
  test_vecsim.py:1362-1396 test_hybrid_query_adhoc_bf_mode (d = 128: scores 128 * (10 j)^2)
  test_vecsim.py:1583-1643 batches and forced ad-hoc return the same list"""
import numpy as np
import pytest

import oracle as O
from tests import hybrid_replay as H


def ramp_index(n, dim, vtype=O.F64):
    idx = O.FlatIndex(vtype, dim, O.L2)
    idx.add_bulk(np.repeat(np.arange(1, n + 1, dtype=np.float64)[:, None], dim, axis=1))
    return idx


@pytest.fixture(scope="module")
def idx6000():
    return ramp_index(6000, 2)


def test_batches_mode_with_text_kats(idx6000):
    n, dim, k = 6000, 2, 10
    q = np.full(dim, float(n))
    index = H.OracleIndex(idx6000)
    # every doc matches the filter: ids n, n-1, ... with scores dim * i^2; the heuristic picks BATCHES (:968)
    it = H.HybridReplay(index, q, k, H.IdListChild(range(1, n + 1)))
    assert it.mode == H.HYBRID_BATCHES
    assert it.results() == [(n - i, float(dim * i * i)) for i in range(k)]
    # 20 % of the docs (ids 5, 10, ...): n - 5 i with dim * (5 i)^2
    it = H.HybridReplay(index, q, k, H.IdListChild(range(5, n + 1, 5)))
    assert it.results() == [(n - 5 * i, float(dim * (5 * i) ** 2)) for i in range(k)]
    assert it.mode == H.HYBRID_BATCHES
    # '@t:other text': the intersection is empty although its estimate is 1200 -> the first batch finds nothing, the
    # re-estimate drops to 600 of 6000 and the policy flips to ad-hoc BF (the test asserts HYBRID_BATCHES_TO_ADHOC_BF)
    it = H.HybridReplay(index, q, k, H.IdListChild([], estimate=1200))
    assert it.results() == [] and it.mode == H.HYBRID_BATCHES_TO_ADHOC_BF and it.num_iterations == 1
    # everything but every 5th doc
    keep = [i for i in range(1, n + 1) if i % 5]
    exp = [(n - i, float(dim * i * i)) for i in range(13) if (n - i) % 5][:k]
    assert H.HybridReplay(index, q, k, H.IdListChild(keep)).results() == exp
    # no child at all: plain KNN
    it = H.HybridReplay(index, q, k, None)
    assert it.mode == H.STANDARD_KNN and it.results() == [(n - i, float(dim * i * i)) for i in range(k)]


def test_adhoc_bf_mode_kat():
    # test_vecsim.py:1362-1396: 100 docs of dim 128, every 10th passes the filter; forced ADHOC_BF
    n, dim, k = 100, 128, 10
    index = H.OracleIndex(ramp_index(n, dim, O.F32))
    q = np.full(dim, float(n), dtype=np.float32)
    it = H.HybridReplay(index, q, k, H.IdListChild(range(10, n + 1, 10)), policy=H.HYBRID_ADHOC_BF)
    assert it.results() == [(n - 10 * j, float(dim * (10 * j) ** 2)) for j in range(k)]
    assert it.mode == H.HYBRID_ADHOC_BF


@pytest.mark.parametrize("batch_size", [0, 7, 100])
def test_policies_agree_on_random_data(batch_size):
    rng = np.random.default_rng(1583 + batch_size)
    n, dim, k = 3000, 6, 10
    idx = O.FlatIndex(O.F32, dim, O.L2)
    idx.add_bulk(rng.standard_normal((n, dim)).astype(np.float32))
    index = H.OracleIndex(idx)
    q = rng.standard_normal(dim).astype(np.float32)
    child = sorted(rng.choice(np.arange(1, n + 1), 700, replace=False).tolist())
    adhoc = H.HybridReplay(index, q, k, H.IdListChild(child), policy=H.HYBRID_ADHOC_BF).results()
    batches = H.HybridReplay(index, q, k, H.IdListChild(child), policy=H.HYBRID_BATCHES, batch_size=batch_size)
    assert batches.results() == adhoc and batches.mode == H.HYBRID_BATCHES
    auto = H.HybridReplay(index, q, k, H.IdListChild(child)).results()
    assert auto == adhoc
    want = sorted((float(d), i) for i, d in zip(child, index.adhoc_distances(q, child)))[:k]
    assert adhoc == [(i, d) for d, i in want]
    if H.reference_heap_available():                       # the stand-in heap gives the same answer (no ties here)
        py = H.HybridReplay(index, q, k, H.IdListChild(child), policy=H.HYBRID_BATCHES, force_python_heap=True).results()
        assert py == adhoc


def test_change_policy_kat():
    # test_vecsim.py:1583-1643 on `VECTOR FLAT`, d = 2, COSINE, n = 6000 random vectors, the filter passes the first half
    rng = np.random.default_rng(10)
    n, dim, k = 6000, 2, 10
    data = rng.random((n, dim)).astype(np.float32)
    idx = O.FlatIndex(O.F32, dim, O.COSINE)
    idx.add_bulk(data)
    index = H.OracleIndex(idx)
    q = rng.random(dim).astype(np.float32)
    half = list(range(1, n // 2 + 1))
    # (1) 10 results in HYBRID_BATCHES; forcing ad-hoc BF returns the same scores
    b = H.HybridReplay(index, q, k, H.IdListChild(half))
    rb = b.results()
    assert b.mode == H.HYBRID_BATCHES and len(rb) == 10
    a = H.HybridReplay(index, q, k, H.IdListChild(half), policy=H.HYBRID_ADHOC_BF)
    assert [s for _, s in a.results()] == [s for _, s in rb] and a.mode == H.HYBRID_ADHOC_BF
    # (2) an empty child whose ESTIMATE is n/2: the policy changes to ad-hoc BF while running the batches, 0 results
    e = H.HybridReplay(index, q, k, H.IdListChild([], estimate=n // 2))
    assert e.results() == [] and e.mode == H.HYBRID_BATCHES_TO_ADHOC_BF and e.num_iterations == 2
    assert e.batch_sizes == [21, 41]                      # 10 * (6000 / 3000) + 1, then 10 * (6000 / 1500) + 1
    assert H.HybridReplay(index, q, k, H.IdListChild([], estimate=n // 2), policy=H.HYBRID_ADHOC_BF).results() == []
    # (3) one valid document (the query itself): found in the first batch, dropped with the heap when the policy
    # changes, found again by the ad-hoc pass
    idx.add(q, n + 1)
    one = H.HybridReplay(index, q, k, H.IdListChild([n + 1], estimate=n // 2))
    res = one.results()
    assert one.mode == H.HYBRID_BATCHES_TO_ADHOC_BF and [i for i, _ in res] == [n + 1]
    assert abs(res[0][1]) < 1e-6


@pytest.mark.parametrize("vtype", [O.F64, O.F32])
def test_hybrid_query_cosine_kat(vtype):
    # test_vecsim.py:1428-1487 on `VECTOR FLAT`: doc i = [i/N, 1, 1, 1], q = [1, 1, 1, 1], COSINE
    n, dim, k = 6000, 4, 10
    rows = np.ones((n, dim))
    rows[:, 0] = np.arange(1, n + 1) / n
    idx = O.FlatIndex(vtype, dim, O.COSINE)
    idx.add_bulk(rows)
    index = H.OracleIndex(idx)
    q = np.ones(dim)
    it = H.HybridReplay(index, q, k, H.IdListChild(range(1, n + 1)))
    ids = [i for i, _ in it.results()]
    assert it.mode == H.HYBRID_BATCHES                     # LAST_SEARCH_MODE asserted by the reference test
    if vtype == O.F64:
        assert ids == [n - i for i in range(10)]
    else:
        assert set(ids) <= {n - i for i in range(15)} and len(ids) == 10
    it = H.HybridReplay(index, q, k, H.IdListChild(range(10, n + 1, 10)))
    ids = [i for i, _ in it.results()]
    assert it.mode == H.HYBRID_ADHOC_BF
    if vtype == O.F64:
        assert ids == [n - 10 * i for i in range(10)]
    else:
        assert set(ids) == {n - 10 * i for i in range(10)}


def test_batches_mode_with_tags_middle_query_kat(idx6000):
    # test_vecsim.py:1040-1093: q = [N/2, N/2]; equal distances on both sides of the query, "closer results will come
    # before (secondary sorting by id)" -- and at the K boundary the LOWER id of the tied pair mid-5 / mid+5 is the one
    # that is kept (strict `<` admission, hybrid_reader.c:321, ids arriving in ascending order from the BY_ID batch)
    n, dim, k = 6000, 2, 10
    mid = n // 2
    index = H.OracleIndex(idx6000)
    q = np.full(dim, float(mid))
    by_score_then_id = lambda res: sorted(res, key=lambda t: (t[1], t[0]))   # the sorter behind the iterator (SORTBY)
    exp = [(mid, 0.0)] + [(mid + (-(i + 1) // 2 if i % 2 else i // 2), float(dim * ((i + 1) // 2) ** 2)) for i in range(1, 10)]
    got = by_score_then_id(H.HybridReplay(index, q, k, H.IdListChild(range(1, n + 1))).results())
    assert got == exp and got[-1][0] == mid - 5
    # ids that divide by 5
    exp5 = [(mid, 0.0)] + [(mid + (-((5 * i + 5) // 2) if i % 2 else (5 * i) // 2), float(dim * (5 * ((i + 1) // 2)) ** 2))
                           for i in range(1, 10)]
    got = by_score_then_id(H.HybridReplay(index, q, k, H.IdListChild(range(5, n + 1, 5))).results())
    assert got == exp5
    # ids that do not divide by 5
    expn, i = [], 0
    while len(expn) < 10:
        if (mid + (i + 1) // 2) % 5:
            expn.append((mid + (-((i + 1) // 2) if i % 2 else i // 2), float(dim * ((i + 1) // 2) ** 2)))
        i += 1
    got = by_score_then_id(H.HybridReplay(index, q, k, H.IdListChild([d for d in range(1, n + 1) if d % 5])).results())
    assert got == expn
