"""Pins oracle/flat_oracle.c against the reference's own known-answer tests for the VecSim seam.

Each test names the reference test it restates.  CPU only.
"""
import math

import numpy as np
import pytest
from scipy.spatial import distance as sdist

import oracle as O


def flat(n, dim, metric=O.L2, vtype=O.F32):
    """TestIndex::flat -- doc i (1..=n) is [i; dim]
    (reference src/redisearch_rs/vector_score_source/src/test_utils.rs:103-113)."""
    idx = O.FlatIndex(vtype, dim, metric)
    for i in range(1, n + 1):
        assert idx.add(np.full(dim, i, dtype=np.float32), i) == 1
    return idx


def flat_cosine(n, dim, vtype=O.F32):
    """TestIndex::flat_cosine -- doc i is [i/n, 1, 1, ...] (test_utils.rs:117-130)."""
    idx = O.FlatIndex(vtype, dim, O.COSINE)
    for i in range(1, n + 1):
        v = np.ones(dim, dtype=np.float64 if vtype == O.F64 else np.float32)
        v[0] = i / n
        idx.add(v, i)
    return idx


def test_flat_unfiltered_returns_top_k_nearest_by_score():
    # source_pytest_parity.rs:35-46
    idx = flat(100, 4)
    ids, sc = idx.topk(np.full(4, 100.0), 10)
    assert ids.tolist() == list(range(100, 90, -1))
    assert sc.tolist() == [4.0 * d * d for d in range(10)]


def test_middle_query_orders_by_distance_then_lower_id():
    # source_pytest_parity.rs:92-109 -- the tie-break KAT
    n, k = 100, 10
    mid = n // 2
    idx = flat(n, 4)
    ids, _ = idx.topk(np.full(4, float(mid)), k)
    expected = [mid]
    for d in range(1, 5):
        expected += [mid - d, mid + d]
    expected.append(mid - 5)
    assert ids.tolist() == expected


def test_dim1_unfiltered_knn_top3():
    # source_pytest_parity.rs:115-124 ; test_vecsim.py:1489-1529 (dist 1,4,9)
    idx = flat(10, 1)
    ids, sc = idx.topk([0.0], 3)
    assert ids.tolist() == [1, 2, 3] and sc.tolist() == [1.0, 4.0, 9.0]


def test_dim1_filtered_subset_adhoc():
    # source_pytest_parity.rs:132-143: children 6..10 -> [6,7,8]; test_vecsim.py: 36,49,64
    idx = flat(10, 1)
    nq = idx.normalized_query([0.0])
    d = {i: idx.distance_from(i, nq) for i in range(6, 11)}
    best = sorted(d, key=lambda i: (d[i], i))[:3]
    assert best == [6, 7, 8] and [d[i] for i in best] == [36.0, 49.0, 64.0]


def test_cosine_top_k_are_highest_ids():
    # source_pytest_parity.rs:68-85 and test_vecsim.py:1428-1487 (FLOAT64 exact, FLOAT32 window)
    n, k = 100, 10
    ids, _ = flat_cosine(n, 4).topk(np.ones(4), k)
    assert ids[0] == n and all(i > n - 15 for i in ids) and len(ids) == k
    ids64, _ = flat_cosine(6000, 4, O.F64).topk(np.ones(4), k)
    assert ids64.tolist() == list(range(6000, 5990, -1))


@pytest.mark.parametrize("vtype,eps", [(O.F32, 1e-6), (O.F64, 1e-9), (O.F16, 1e-2), (O.BF16, 1e-2)])
def test_sanity_cosine_and_l2_distances(vtype, eps):
    # test_vecsim.py:65-139 (cosine) and :141-212 (L2): scipy distances, per-type tolerance (:14)
    vecs = [[0.1, 0.1], [0.1, 0.2], [0.1, 0.3], [0.1, 0.4]]
    q = np.array([0.1, 0.1])
    for metric, fn in ((O.COSINE, sdist.cosine), (O.L2, sdist.sqeuclidean)):
        idx = O.FlatIndex(vtype, 2, metric)
        for i, v in enumerate(vecs):
            idx.add(np.array(v), i + 1)
        ids, sc = idx.topk(q, 4)
        assert ids.tolist() == [1, 2, 3, 4]
        for i, s in zip(ids, sc):
            assert abs(s - fn(np.array(vecs[i - 1]), q)) <= eps
        # delete-then-requery (test_vecsim.py:118-139)
        assert idx.delete(1) == 1
        ids, _ = idx.topk(q, 4)
        assert ids.tolist() == [2, 3, 4]


def test_ip_is_one_minus_dot():
    # tests/pytests/test_hybrid_vector_normalizer.py:57-58
    idx = O.FlatIndex(O.F32, 3, O.IP)
    idx.add(np.array([1.0, 2.0, 3.0]), 7)
    _, sc = idx.topk(np.array([0.5, 0.25, 2.0]), 1)
    assert sc[0] == pytest.approx(1.0 - (0.5 + 0.5 + 6.0))


def test_l2_scores_dim_times_i_squared():
    # test_vecsim.py:982-986, :1267-1277 (128*i^2) and :1362-1396 (128*(10j)^2)
    idx = flat(100, 128)
    ids, sc = idx.topk(np.full(128, 100.0), 10)
    assert sc.tolist() == [128.0 * i * i for i in range(10)]
    nq = idx.normalized_query(np.full(128, 100.0))
    assert [idx.distance_from(100 - 10 * j, nq) for j in range(10)] == [128.0 * (10 * j) ** 2 for j in range(10)]


def test_knn_zero_and_empty_index():
    # test_vecsim.py:214-243 (KNN 0 => empty); empty index
    idx = flat(5, 2)
    assert idx.topk([0.0, 0.0], 0)[0].size == 0
    assert O.FlatIndex(O.F32, 2, O.L2).topk([0.0, 0.0], 3)[0].size == 0


def test_missing_label_is_nan_and_overwrite():
    # hybrid_reader.c:316-320 ; SURVEY 8c(viii) overwrite-on-duplicate
    idx = flat(5, 2)
    nq = idx.normalized_query([0.0, 0.0])
    assert math.isnan(idx.distance_from(99, nq))
    assert idx.add(np.array([9.0, 9.0]), 3) == 0 and len(idx) == 5
    assert idx.distance_from(3, nq) == 162.0


@pytest.mark.parametrize("vtype", [O.F32, O.F64, O.F16, O.BF16])
def test_range_query_inclusive(vtype):
    # test_vecsim.py:2068-2111: n=99 docs [i]^4, q=[100]^4, radius 4*46^2 => exactly 46 results,
    # farthest exactly at the radius; BY_ID ascending by default; empty index => nothing
    idx = flat(99, 4, vtype=vtype)
    ids, sc = idx.range(np.full(4, 100.0), 4 * 46 ** 2, O.BY_ID)
    assert ids.tolist() == list(range(54, 100)) and sc[0] == 4 * 46 ** 2
    ids, sc = idx.range(np.full(4, 100.0), 4 * 46 ** 2, O.BY_SCORE)
    assert ids.tolist() == list(range(99, 53, -1))
    assert O.FlatIndex(vtype, 4, O.L2).range(np.zeros(4), 10.0)[0].size == 0


def test_batches_are_disjoint_next_best_by_id():
    # hybrid_reader.c:387-441 ; top_k/src/traits.rs:21-23 (ids strictly increasing inside a batch)
    idx = flat(100, 4)
    it = idx.batches(np.full(4, 100.0))
    seen = []
    while it.has_next():
        ids, sc = it.next(7, O.BY_ID)
        assert ids.tolist() == sorted(ids.tolist())
        if seen:
            assert max(sc) >= 0 and min(ids) < min(seen)  # next-best lie further away
        seen += ids.tolist()
    assert sorted(seen) == list(range(1, 101)) and seen[:7] == list(range(94, 101))


def test_batches_vs_adhoc_same_scores():
    # test_vecsim.py:1583-1643: both policies return identical score lists
    rng = np.random.default_rng(5)
    idx = O.FlatIndex(O.F32, 8, O.L2)
    idx.add_bulk(rng.standard_normal((500, 8)).astype(np.float32))
    q = rng.standard_normal(8).astype(np.float32)
    child = set(range(1, 501, 3))
    nq = idx.normalized_query(q)
    adhoc = sorted((idx.distance_from(i, nq), i) for i in child)[:10]
    got, it = [], idx.batches(q)
    while it.has_next() and len(got) < 10:
        ids, sc = it.next(25, O.BY_ID)
        got += [(s, int(i)) for i, s in zip(ids, sc) if int(i) in child]
    assert sorted(got)[:10] == adhoc


def test_prefer_adhoc_pins():
    # the four FLAT decision points + small-index rule
    assert O.prefer_adhoc(6000, 4, 6000, 10) == (False, 3)          # test_vecsim.py:1436-1478
    assert O.prefer_adhoc(6000, 4, 600, 10) == (True, 2)
    assert O.prefer_adhoc(6000, 2, 3000, 10)[0] is False            # :1615-1643
    est = 3000
    for _ in range(8):                                              # zero hits: estimate halves
        est //= 2
        r, mode = O.prefer_adhoc(6000, 2, est, 10, initial_check=False)
        if r:
            break
    assert r and mode == 4                                          # HYBRID_BATCHES_TO_ADHOC_BF
    assert O.prefer_adhoc(10, 1, 5, 3)[0] and O.prefer_adhoc(1000, 4, 31, 10)[0]


def test_refine_child_estimated_kats():
    # vector_score_source/src/source.rs:594-722 (formula hybrid_reader.c:355-365)
    def refine(est, new_results, n_left, index_size, upper):
        cur = int(np.float32(new_results) / np.float32(n_left) * index_size)
        return min((est + cur) // 2, upper)
    assert refine(10, 5, 10, 100, 10) == 10
    assert refine(80, 0, 10, 1000, 80) == 40
    assert refine(500, 10, 10, 1000, 500) == 500
    assert refine(0, 1, 10, 1000, 0) == 0


def test_multi_value_keeps_best_per_label():
    idx = O.FlatIndex(O.F32, 2, O.L2, multi=True)
    idx.add(np.array([0.0, 0.0]), 1)
    idx.add(np.array([5.0, 5.0]), 1)
    idx.add(np.array([1.0, 1.0]), 2)
    ids, sc = idx.topk(np.array([5.0, 5.0]), 5)
    assert ids.tolist() == [1, 2] and sc.tolist() == [0.0, 32.0]
    assert idx.delete(1) == 2 and len(idx) == 1


def test_half_conversions_roundtrip():
    xs = np.array([0.0, 1.0, -2.5, 65504.0, 1e-5, 6.1e-5, 0.1, 3.14159, 1e6], dtype=np.float32)
    with np.errstate(over="ignore"):                     # 1e6 -> inf is one of the cases
        for x in xs:
            assert O.lib.oracle_f32_to_f16(float(x)) == int(np.float16(x).view(np.uint16))
    for h in range(0, 0x7C00, 37):
        assert O.lib.oracle_f16_to_f32(h) == float(np.uint16(h).view(np.float16))


def test_heap_scan_matches_sort():
    rng = np.random.default_rng(1)
    idx = O.FlatIndex(O.F32, 32, O.COSINE)
    idx.add_bulk(rng.uniform(-1, 1, (3000, 32)).astype(np.float32))
    q = rng.uniform(-1, 1, 32).astype(np.float32)
    a, b = idx.topk(q, 10), idx.topk(q, 10, heap=True)
    assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()


def test_philox4x32_10_random123_known_answers():
    """The keyed corpus generator is Philox4x32-10; Random123's kat_vectors for it."""
    import ctypes as C

    def ph(ctr, key):
        c, k, o = (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), (C.c_uint32 * 4)()
        O.lib.oracle_philox4x32_10(c, k, o)
        return list(o)
    assert ph([0] * 4, [0] * 2) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert ph([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert ph([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_philox_rows_are_a_pure_function_of_seed_row_column():
    a = O.philox_rows(11, 0, 64, 37)
    assert a.dtype == np.float32 and a.min() >= -1.0 and a.max() < 1.0
    assert np.array_equal(O.philox_rows(11, 10, 5, 37), a[10:15])            # any row can be regenerated alone
    assert np.array_equal(O.philox_rows(11, 0, 64, 20), a[:, :20])           # columns do not depend on dim
    assert not np.array_equal(O.philox_rows(12, 0, 64, 37), a)
    assert np.array_equal(O.philox_rows(11, 2 ** 33, 2, 8), O.philox_rows(11, 2 ** 33, 2, 8, threads=1))
    assert not np.array_equal(O.philox_rows(11, 2 ** 33, 2, 8), O.philox_rows(11, 0, 2, 8))   # high counter word is used
    big = O.philox_rows(3, 0, 4096, 256)
    assert abs(float(big.mean())) < 5e-3 and abs(float(big.std()) - 1 / np.sqrt(3)) < 5e-3  # U(-1,1)
    assert np.array_equal(O.philox_rows(11, 0, 8, 9, O.F16), O.philox_rows(11, 0, 8, 9).astype(np.float16))
    assert np.array_equal(O.philox_rows(11, 0, 8, 9, O.F64), O.philox_rows(11, 0, 8, 9).astype(np.float64))
    u8 = O.philox_rows(11, 0, 8, 9, O.U8)
    assert u8.dtype == np.uint8 and np.array_equal(O.philox_rows(11, 0, 8, 9, O.I8).view(np.uint8), u8)


def test_fp16_hardware_conversion_path_equals_the_bit_level_one():
    """oracle/flat_oracle.c dot_or_l2_f16: the F16C loop (picked at run time) and the scalar loop over f16_to_f32 must give
    the same float for every length, incl. subnormals, infinities and ragged tails."""
    rng = np.random.default_rng(2)
    O.lib.oracle_f16_paths_agree.restype = int
    for d in (1, 7, 15, 16, 17, 33, 100, 768, 1000):
        for _ in range(20):
            a = rng.standard_normal(d).astype(np.float16)
            b = (rng.standard_normal(d) * rng.choice([1e-6, 1.0, 200.0])).astype(np.float16)
            a[rng.integers(0, d)] = np.float16(6e-8)        # subnormal
            assert O.lib.oracle_f16_paths_agree(O._p(a), O._p(b), d) == 1
