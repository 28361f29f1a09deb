"""GPU: batched / coalesced queries over PLAIN FLOAT32 IP / cosine indexes through the int8 matrix cores with the rows quantised IN
FLIGHT (round 6, knob gemm_qs_f8: gemm_qs_h8r_kernel<.., SRC_F8> -- fp32 chunks global -> registers, four v_fma_f32 + three
v_perm_b32 per chunk, int8 tile in LDS; nothing stored next to the index but four index-wide numbers).  The survivors of the
error-banded filter are re-scored from the fp32 rows with the single-query scan's arithmetic: replies BIT-IDENTICAL to
VecSimIndex_TopKQuery, and to the bf16-in-flight route (gemm_qs_f32_kernel) the same index takes with the knob off.
Reference: B x VecSimIndex_TopKQuery, src/iterators/hybrid_reader.c:374."""
import threading

import numpy as np
import pytest

from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu
F32, IP, COS, L2 = V.VecSimType_FLOAT32, V.VecSimMetric_IP, V.VecSimMetric_Cosine, V.VecSimMetric_L2


@pytest.fixture
def lib():
    lb = V.load()
    lb.RSGPU_SetTuning(b"gemm_qs_f8", 1)
    yield lb
    lb.RSGPU_SetTuning(b"gemm_qs_f8", 1)


def check(lib, g, queries, k, want, expect_launches=None):
    lib.RSGPU_ResetProfile()
    lib.RSGPU_SetProfiling(1)
    ids, sc, cnt = g.topk_batch(queries, k)
    lib.RSGPU_SetProfiling(0)
    launches, _, by = V.scan_profile()
    if expect_launches is not None:
        assert launches == expect_launches, "the batched path was not taken (%d profiled launches)" % launches
    for i, (wi, ws) in enumerate(want):
        assert cnt[i] == len(wi)
        assert ids[i][: cnt[i]].tolist() == wi.tolist(), i
        assert np.array_equal(sc[i][: cnt[i]], ws, equal_nan=True), i
    return ids, sc, cnt


@pytest.mark.parametrize("metric", [COS, IP])
@pytest.mark.parametrize("dim,n", [(768, 530_001), (512, 540_000), (384, 560_000), (256, 700_000), (128, 900_000)])
@pytest.mark.parametrize("k", [10, 100])
def test_fp32_rows_quantised_in_flight_are_bit_identical_to_single_queries(lib, metric, dim, n, k):
    g = V.VecSimIndex(F32, dim, metric)
    assert g.add_philox_rows(dim + k, 0, n, 1) == n
    queries = np.random.default_rng(dim * 3 + k).uniform(-1, 1, (300, dim)).astype(np.float32)
    want = [g.topk_query(q, k).results() for q in queries]
    got = check(lib, g, queries, k, want, expect_launches=2)
    assert lib.RSGPU_LastBatchRoute() == 6               # (fp32 rows quantised to int8 in flight: rsgpu_ext.h)
    lib.RSGPU_SetTuning(b"gemm_qs_f8", 0)               # the bf16-in-flight route over the same index: the same replies
    lib.RSGPU_ResetProfile()
    ids0, sc0, cnt0 = g.topk_batch(queries, k)
    assert lib.RSGPU_LastBatchRoute() == 2
    lib.RSGPU_SetTuning(b"gemm_qs_f8", 1)
    assert np.array_equal(ids0, got[0]) and np.array_equal(sc0, got[1]) and np.array_equal(cnt0, got[2])
    g.free()


def test_l2_and_indexes_created_without_the_knob_keep_their_routes(lib):
    n, dim, k = 540_000, 256, 10
    q = np.random.default_rng(1).uniform(-1, 1, (20, dim)).astype(np.float32)
    g = V.VecSimIndex(F32, dim, L2)
    g.add_philox_rows(5, 0, n, 1)
    check(lib, g, q, k, [g.topk_query(v, k).results() for v in q], expect_launches=1)
    g.free()
    lib.RSGPU_SetTuning(b"gemm_qs_f8", 0)
    g = V.VecSimIndex(F32, dim, COS)
    lib.RSGPU_SetTuning(b"gemm_qs_f8", 1)
    g.add_philox_rows(5, 0, n, 1)
    check(lib, g, q, k, [g.topk_query(v, k).results() for v in q], expect_launches=1)
    g.free()


def test_gaussian_rows_outliers_appends_and_deletes(lib):
    rng = np.random.default_rng(7)
    dim, n, k = 256, 560_000, 10
    g = V.VecSimIndex(F32, dim, IP)
    g.add_philox_rows(9, 0, n, 1)
    queries = rng.standard_normal((16, dim)).astype(np.float32)
    single = lambda: [g.topk_query(q, k).results() for q in queries]
    check(lib, g, queries, k, single(), expect_launches=1)
    extra = rng.standard_normal((40, dim)).astype(np.float32)
    extra[3] = queries[0] * 0.5
    for i in range(40):
        g.add_vector(extra[i], n + 1 + i)
    check(lib, g, queries, k, single(), expect_launches=1)
    for lbl in (5, 77, n + 4, 300_000):
        g.delete_vector(lbl)
    check(lib, g, queries, k, single(), expect_launches=1)
    g.add_vector(queries[3] * 9.0, 9_000_000)            # a row that outgrows the scale: the maxima are taken again
    want = single()
    assert want[3][0][0] == 9_000_000
    check(lib, g, queries, k, want)                      # (nine times coarser levels: lists may overflow -> the exact fall-back)
    g.free()


def test_concurrent_callers_take_the_wide_passes(lib):
    dim, n, k = 256, 600_000, 10
    g = V.VecSimIndex(F32, dim, COS)
    g.add_philox_rows(11, 0, n, 1)
    queries = np.random.default_rng(41).uniform(-1, 1, (64, dim)).astype(np.float32)
    want = [g.topk_query(q, k).results() for q in queries]
    got = [None] * len(queries)
    gate = threading.Barrier(32)

    def work(t):
        gate.wait()
        for i in range(t, len(queries), 32):
            got[i] = g.topk_query(queries[i], k).results()
    th = [threading.Thread(target=work, args=(t,)) for t in range(32)]
    [t.start() for t in th]
    [t.join() for t in th]
    for (wi, ws), (gi, gs) in zip(want, got):
        assert gi.tolist() == wi.tolist() and gs.tolist() == ws.tolist()
    g.free()
