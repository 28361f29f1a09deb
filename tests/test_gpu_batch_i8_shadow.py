"""GPU tests of the batched path over FLOAT16 IP / cosine indexes that carry the int8 shadow with one index-wide scale
(opt-in "shadow8"): the query-stationary filter passes run on the INT8 matrix cores over half the bytes, every threshold
is widened by the query's own Cauchy-Schwarz error band (actual quantisation-error norms), the survivors are re-scored
from the fp16 rows with the single-query scan's arithmetic.  The result must be BIT-IDENTICAL to one
VecSimIndex_TopKQuery per query (ids and distances) -- on benign data, on data that makes the band useless (the lists
overflow and the host falls back), after appends / deletes / a row that outgrows the scale -- and the batched path must
really have been taken.  (Shapes: 768-byte-multiple int8 rows the query-stationary kernel has a ring for -- dim 256, 512,
768, 1024, 1536.)"""
import numpy as np
import pytest
import torch

from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu
F16, IP, COS = V.VecSimType_FLOAT16, V.VecSimMetric_IP, V.VecSimMetric_Cosine


@pytest.fixture
def shadow8():
    lib = V.load()
    lib.RSGPU_SetTuning(b"shadow8", 1)
    yield lib
    lib.RSGPU_SetTuning(b"shadow8", 0)


def build(x, dim, metric):
    g = V.VecSimIndex(F16, dim, metric)
    torch.cuda.synchronize()
    g.add_device_rows(x.data_ptr(), x.shape[0], 1)
    return g


def plain_answers(x, dim, metric, queries, k, lib):
    """the same queries one by one on an index WITHOUT the shadow"""
    lib.RSGPU_SetTuning(b"shadow8", 0)
    g = build(x, dim, metric)
    out = [g.topk_query(q, k).results() for q in queries]
    g.free()
    lib.RSGPU_SetTuning(b"shadow8", 1)
    return out


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
        assert sc[i][: cnt[i]].tolist() == ws.tolist(), i
    return launches, by


@pytest.mark.parametrize("metric", [IP, COS])
@pytest.mark.parametrize("dim,n", [(768, 530_001), (512, 540_000), (256, 700_000)])
@pytest.mark.parametrize("k", [10, 100])
def test_int8_shadow_pass_is_bit_identical_to_single_queries(shadow8, metric, dim, n, k):
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(dim * 11 + k)
    x = (torch.rand((n, dim), device=dev, generator=gen) * 2 - 1).to(torch.float16)
    b = 300                                              # two passes, the second one padded
    queries = np.random.default_rng(dim + k).uniform(-1, 1, (b, dim)).astype(np.float16)
    want = plain_answers(x.clone(), dim, metric, queries, k, shadow8)
    g = build(x, dim, metric)
    launches, by = check(shadow8, g, queries, k, want, expect_launches=2)
    assert by == 2 * n * dim, "the passes did not read the int8 shadow (%d bytes accounted)" % by   # one byte per element


def test_gaussian_rows_and_outliers(shadow8):
    # heavy-tailed coordinates make the index-wide scale coarse for most rows: a wide band, still the exact answer
    dev = torch.device("cuda", 0)
    dim, n, k = 256, 600_000, 20
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    x = torch.randn((n, dim), device=dev, generator=gen)
    x[::1000] *= 6.0
    x = x.to(torch.float16)
    queries = np.random.default_rng(6).standard_normal((40, dim)).astype(np.float16)
    want = plain_answers(x.clone(), dim, IP, queries, k, shadow8)
    g = build(x, dim, IP)
    check(shadow8, g, queries, k, want)


def test_clustered_rows_overflow_the_band_and_fall_back_exactly(shadow8):
    dev = torch.device("cuda", 0)
    dim, n, k = 256, 600_000, 10
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    centre = torch.rand((1, dim), device=dev, generator=gen) * 2 - 1
    x = (centre + 1e-3 * (torch.rand((n, dim), device=dev, generator=gen) * 2 - 1)).to(torch.float16)
    queries = (centre.cpu().numpy() + 1e-3 * np.random.default_rng(4).uniform(-1, 1, (5, dim))).astype(np.float16)
    want = plain_answers(x.clone(), dim, COS, queries, k, shadow8)
    g = build(x, dim, COS)
    check(shadow8, g, queries, k, want)


def test_appends_deletes_and_a_row_that_outgrows_the_scale(shadow8):
    dev = torch.device("cuda", 0)
    dim, n, k = 256, 560_000, 10
    gen = torch.Generator(device=dev)
    gen.manual_seed(8)
    x = (torch.rand((n, dim), device=dev, generator=gen) * 2 - 1).to(torch.float16)
    queries = np.random.default_rng(9).uniform(-1, 1, (12, dim)).astype(np.float16)
    g = build(x, dim, IP)
    single = lambda: [g.topk_query(q, k).results() for q in queries]      # (the single-query path never reads the shadow)
    check(shadow8, g, queries, k, single(), expect_launches=1)
    # appended rows (one of them aligned with query 0: it must enter that query's answer) are quantised on demand
    extra = (np.random.default_rng(10).uniform(-1, 1, (50, dim))).astype(np.float16)
    extra[7] = queries[0]
    for i in range(50):
        g.add_vector(extra[i], n + 1 + i)
    want = single()
    assert want[0][0][0] == n + 8
    check(shadow8, g, queries, k, want, expect_launches=1)
    # deletes move rows around below the built prefix
    for lbl in (5, 77, n + 8, 300_000):
        g.delete_vector(lbl)
    check(shadow8, g, queries, k, single(), expect_launches=1)
    # a row four times larger than anything stored: the scale no longer fits, everything is quantised again
    big = (queries[3].astype(np.float32) * 4).astype(np.float16)
    g.add_vector(big, 9_000_000)
    want = single()
    assert want[3][0][0] == 9_000_000
    check(shadow8, g, queries, k, want, expect_launches=1)
    # an infinite element: no scale bounds the index, the plain fp16 passes answer (tolerance of the MFMA order)
    bad = extra[0].copy()
    bad[3] = np.float16(np.inf)
    g.add_vector(bad, 9_000_001)
    ids, sc, cnt = g.topk_batch(queries[:3], k)
    for i, (wi, ws) in enumerate([g.topk_query(q, k).results() for q in queries[:3]]):
        assert len(set(ids[i].tolist()) & set(wi.tolist())) >= k - 1


def test_zero_query_and_small_corpus(shadow8):
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(12)
    dim, n, k = 256, 540_000, 10
    x = (torch.rand((n, dim), device=dev, generator=gen) * 2 - 1).to(torch.float16)
    g = build(x, dim, IP)
    queries = np.random.default_rng(13).uniform(-1, 1, (4, dim)).astype(np.float16)
    queries[2] = 0
    check(shadow8, g, queries, k, [g.topk_query(q, k).results() for q in queries])
    small = build(x[:100_000].contiguous(), dim, IP)            # below the batched cut-over: the plain fp16 passes
    ids, sc, cnt = small.topk_batch(queries, k)                 # (MFMA summation order: tolerance, not bit-equality)
    for i, q in enumerate(queries):
        wi, ws = small.topk_query(q, k).results()
        assert len(set(ids[i].tolist()) & set(wi.tolist())) >= k - 1 or i == 2
        assert np.allclose(np.sort(sc[i]), np.sort(ws), atol=2e-3)


# ---- FLOAT32 indexes created with shadow8: the same int8 passes for THEIR batches (re-scored from the fp32 rows) -----------
F32 = V.VecSimType_FLOAT32


def build_f32(x, dim, metric):
    g = V.VecSimIndex(F32, dim, metric)
    torch.cuda.synchronize()
    g.add_device_rows(x.data_ptr(), x.shape[0], 1)
    return g


@pytest.mark.parametrize("metric", [IP, COS])
@pytest.mark.parametrize("dim,n,k", [(768, 530_001, 10), (256, 700_000, 100), (512, 540_000, 1000)])
def test_f32_index_batches_through_the_int8_rows(shadow8, metric, dim, n, k):
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(dim * 13 + k)
    x = torch.rand((n, dim), device=dev, generator=gen) * 2 - 1
    b = 300
    queries = np.random.default_rng(dim + k + 1).uniform(-1, 1, (b, dim)).astype(np.float32)
    shadow8.RSGPU_SetTuning(b"shadow8", 0)                 # plain fp32 index: the exact answers
    p = build_f32(x.clone(), dim, metric)
    want = [p.topk_query(q, k).results() for q in queries]
    p.free()
    shadow8.RSGPU_SetTuning(b"shadow8", 1)
    g = build_f32(x, dim, metric)
    launches, by = check(shadow8, g, queries, k, want, expect_launches=2)
    assert by == 2 * n * dim                               # both passes read one byte per element
    # its single queries (two-stage over the per-row int8 shadow) give the same answers
    for i in (0, 150, 299):
        si, ss = g.topk_query(queries[i], k).results()
        assert si.tolist() == want[i][0].tolist() and ss.tolist() == want[i][1].tolist()


def test_f32_index_appends_deletes_and_growth(shadow8):
    dev = torch.device("cuda", 0)
    dim, n, k = 256, 560_000, 10
    gen = torch.Generator(device=dev)
    gen.manual_seed(21)
    x = torch.rand((n, dim), device=dev, generator=gen) * 2 - 1
    queries = np.random.default_rng(22).uniform(-1, 1, (9, dim)).astype(np.float32)
    g = build_f32(x, dim, IP)
    shadow8.RSGPU_SetTuning(b"two_stage", 0)               # single queries as the plain fp32 scan: the reference answers
    single = lambda: [g.topk_query(q, k).results() for q in queries]
    want = single()
    shadow8.RSGPU_SetTuning(b"two_stage", 1)
    check(shadow8, g, queries, k, want, expect_launches=1)
    extra = np.random.default_rng(23).uniform(-1, 1, (40, dim)).astype(np.float32)
    extra[5] = queries[1]
    for i in range(40):
        g.add_vector(extra[i], n + 1 + i)
    for lbl in (9, 100_000, n + 3):
        g.delete_vector(lbl)
    g.add_vector(queries[2] * 3, 8_000_000)                # outgrows the scale
    shadow8.RSGPU_SetTuning(b"two_stage", 0)
    want = single()
    shadow8.RSGPU_SetTuning(b"two_stage", 1)
    assert want[1][0][0] == n + 6 and want[2][0][0] == 8_000_000
    check(shadow8, g, queries, k, want, expect_launches=1)


# ---- BFLOAT16 indexes (round 3): the same int8 passes, survivors re-scored with the bf16 scan's own arithmetic ----------
@pytest.mark.parametrize("metric", [IP, COS])
@pytest.mark.parametrize("dim,n", [(768, 530_001), (256, 700_000)])
@pytest.mark.parametrize("k", [10, 100])
def test_bfloat16_index_through_the_int8_passes(shadow8, metric, dim, n, k):
    BF16 = V.VecSimType_BFLOAT16
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(dim * 7 + k)
    x = (torch.rand((n, dim), device=dev, generator=gen) * 2 - 1).to(torch.bfloat16)
    queries = np.random.default_rng(dim + k).uniform(-1, 1, (300, dim)).astype(np.float32)

    def build_bf(rows):
        g = V.VecSimIndex(BF16, dim, metric)
        torch.cuda.synchronize()
        g.add_device_rows(rows.data_ptr(), rows.shape[0], 1)
        return g
    shadow8.RSGPU_SetTuning(b"shadow8", 0)
    p = build_bf(x.clone())
    want = [p.topk_query(q, k).results() for q in queries]
    p.free()
    shadow8.RSGPU_SetTuning(b"shadow8", 1)
    g = build_bf(x)
    try:
        launches, by = check(shadow8, g, queries, k, want, expect_launches=2)
        assert by == 2 * n * dim                     # one byte per element: the int8 rows were what the passes read
        # appends (one aligned with query 0) and deletes, then again
        extra = np.random.default_rng(3).uniform(-1, 1, (20, dim)).astype(np.float32)
        extra[4] = queries[0]
        for i in range(20):
            g.add_vector(extra[i], n + 1 + i)
        for lbl in (7, n + 3, 1234):
            g.delete_vector(lbl)
        single = [g.topk_query(q, k).results() for q in queries[:12]]
        check(shadow8, g, queries[:12], k, single)
    finally:
        g.free()
