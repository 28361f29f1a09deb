"""GPU tests of the batched path over PLAIN FLOAT16 IP / cosine indexes (BASELINE configs[2]'s shape) since round 6: the
query-stationary filter passes quantise the fp16 rows to int8 IN FLIGHT (redisearch_amd/csrc/h8_quant.hpp: one v_pk_fma_f16
per two elements, one v_perm_b32 per four) and run on the INT8 matrix cores -- nothing is stored next to the index but four
index-wide numbers (the largest element, the largest |x8|^2 and |ex|^2 under that very quantiser).  Every threshold is widened
by the query's own Cauchy-Schwarz band from those numbers and the survivors are re-scored from the fp16 rows with the
single-query scan's arithmetic: the replies must be BIT-IDENTICAL to one VecSimIndex_TopKQuery per query -- on benign data, on
data that makes the band useless, after appends / deletes / a row that outgrows the scale, with tiny and huge element ranges --
identical to what the fp16 MFMA pass (knob gemm_qs_h8 = 0) and the stored int8 shadow return, and the int8 pass must really
have run over the fp16 bytes.  Reference: B x VecSimIndex_TopKQuery, src/iterators/hybrid_reader.c:374."""
import numpy as np
import pytest
import torch

from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu
F16, BF16, IP, COS = V.VecSimType_FLOAT16, V.VecSimType_BFLOAT16, V.VecSimMetric_IP, V.VecSimMetric_Cosine


@pytest.fixture
def lib():
    lb = V.load()
    lb.RSGPU_SetTuning(b"gemm_qs_h8", 5)
    yield lb
    lb.RSGPU_SetTuning(b"gemm_qs_h8", 5)


def build(x, dim, metric, vtype=F16):
    g = V.VecSimIndex(vtype, dim, metric)
    torch.cuda.synchronize()
    g.add_device_rows(x.data_ptr(), x.shape[0], 1)
    return g


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
        assert np.array_equal(sc[i][: cnt[i]], ws, equal_nan=True), i      # (a zero query under cosine: NaN on both paths)
    return launches, by, (ids, sc, cnt)


def rand_rows(n, dim, seed, scale=1.0):
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    return ((torch.rand((n, dim), device=dev, generator=gen) * 2 - 1) * scale).to(torch.float16)


@pytest.mark.parametrize("metric", [IP, COS])
@pytest.mark.parametrize("dim,n", [(768, 530_001), (512, 540_000), (256, 700_000)])
@pytest.mark.parametrize("k,shape", [(10, 2), (100, 2), (100, 1), (10, 5), (100, 5)])
def test_in_flight_int8_pass_is_bit_identical_to_single_queries(lib, metric, dim, n, k, shape):
    # 2 / 1: every wave quantises its fragments (four waves x 64 queries / eight x 32); 5: quantised once per workgroup (gemm_qs_h8r_kernel)
    lib.RSGPU_SetTuning(b"gemm_qs_h8", shape)
    x = rand_rows(n, dim, dim * 13 + k)
    b = 300                                              # two passes, the second one padded
    queries = np.random.default_rng(dim + k).uniform(-1, 1, (b, dim)).astype(np.float16)
    g = build(x, dim, metric)
    want = [g.topk_query(q, k).results() for q in queries]
    launches, by, got = check(lib, g, queries, k, want, expect_launches=2)
    assert by == 2 * n * dim * 2, "the passes did not read the fp16 rows (%d bytes accounted)" % by
    # ... and the fp16 MFMA pass over the same index gives the same replies
    lib.RSGPU_SetTuning(b"gemm_qs_h8", 0)
    ids0, sc0, cnt0 = g.topk_batch(queries, k)
    assert np.array_equal(ids0, got[0]) and np.array_equal(sc0, got[1]) and np.array_equal(cnt0, got[2])
    g.free()


def test_the_route_is_taken_only_where_it_applies(lib):
    """BFLOAT16 rows, L2, multi-value indexes keep their own passes; the knob read at creation keeps a FLOAT16 index off it"""
    n, dim, k = 530_000, 256, 10
    x = rand_rows(n, dim, 1)
    q = np.random.default_rng(2).uniform(-1, 1, (20, dim)).astype(np.float16)
    lib.RSGPU_SetTuning(b"gemm_qs_h8", 0)
    g = build(x, dim, IP)                               # created with the knob off: no stats buffer, fp16 passes for good
    lib.RSGPU_SetTuning(b"gemm_qs_h8", 5)
    want = [g.topk_query(v, k).results() for v in q]
    check(lib, g, q, k, want, expect_launches=1)
    g.free()
    g = build(x, dim, V.VecSimMetric_L2)
    want = [g.topk_query(v, k).results() for v in q]
    check(lib, g, q, k, want, expect_launches=1)
    g.free()


def test_gaussian_rows_and_outliers(lib):
    # heavy-tailed coordinates make the index-wide scale coarse for most rows: a wide band, still the exact answer
    dev = torch.device("cuda", 0)
    dim, n, k = 256, 600_000, 20
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    x = torch.randn((n, dim), device=dev, generator=gen)
    x[::1000] *= 6.0
    x = x.to(torch.float16)
    queries = np.random.default_rng(6).standard_normal((40, dim)).astype(np.float16)
    g = build(x, dim, IP)
    want = [g.topk_query(q, k).results() for q in queries]
    check(lib, g, queries, k, want)


def test_clustered_rows_overflow_the_band_and_fall_back_exactly(lib):
    dev = torch.device("cuda", 0)
    dim, n, k = 256, 600_000, 10
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    centre = torch.rand((1, dim), device=dev, generator=gen) * 2 - 1
    x = (centre + 1e-3 * (torch.rand((n, dim), device=dev, generator=gen) * 2 - 1)).to(torch.float16)
    queries = (centre.cpu().numpy() + 1e-3 * np.random.default_rng(4).uniform(-1, 1, (5, dim))).astype(np.float16)
    g = build(x, dim, COS)
    want = [g.topk_query(q, k).results() for q in queries]
    check(lib, g, queries, k, want)


@pytest.mark.parametrize("scale", [3e-4, 1.0, 900.0])
def test_tiny_and_huge_elements(lib, scale):
    """max |x_i| far from 1: the fp16 inverse scale saturates at 65504 (tiny rows: fewer than 127 levels are used) or is a
    small fraction (huge rows); subnormal fp16 elements appear in the tiny case"""
    dim, n, k = 256, 560_000, 10
    x = rand_rows(n, dim, 21, scale)
    queries = (np.random.default_rng(22).uniform(-1, 1, (24, dim)) * scale).astype(np.float16)
    for metric in (IP, COS):
        g = build(x.clone(), dim, metric)
        want = [g.topk_query(q, k).results() for q in queries]
        # (tiny rows use ~20 of the 127 levels: the band may be wide enough for the lists to overflow -> exact fall-back)
        check(lib, g, queries, k, want, expect_launches=1 if scale >= 1.0 else None)
        g.free()


def test_appends_deletes_and_a_row_that_outgrows_the_scale(lib):
    dim, n, k = 256, 560_000, 10
    x = rand_rows(n, dim, 8)
    queries = np.random.default_rng(9).uniform(-1, 1, (12, dim)).astype(np.float16)
    g = build(x, dim, IP)
    single = lambda: [g.topk_query(q, k).results() for q in queries]
    check(lib, g, queries, k, single(), expect_launches=1)
    # appended rows (one of them aligned with query 0: it must enter that query's answer) enter the maxima on demand
    extra = (np.random.default_rng(10).uniform(-1, 1, (50, dim))).astype(np.float16)
    extra[7] = queries[0]
    for i in range(50):
        g.add_vector(extra[i], n + 1 + i)
    want = single()
    assert want[0][0][0] == n + 8
    check(lib, g, queries, k, want, expect_launches=1)
    # deletes move rows around: nothing is stored per row, the maxima stay bounds
    for lbl in (5, 77, n + 8, 300_000):
        g.delete_vector(lbl)
    check(lib, g, queries, k, single(), expect_launches=1)
    # a row four times larger than anything stored: the scale no longer fits, the maxima are taken again
    big = (queries[3].astype(np.float32) * 4).astype(np.float16)
    g.add_vector(big, 9_000_000)
    want = single()
    assert want[3][0][0] == 9_000_000
    check(lib, g, queries, k, want, expect_launches=1)
    # an infinite element: no scale bounds the index, the plain fp16 passes answer
    bad = extra[0].copy()
    bad[3] = np.float16(np.inf)
    g.add_vector(bad, 9_000_001)
    ids, sc, cnt = g.topk_batch(queries[:3], k)
    for i, (wi, ws) in enumerate([g.topk_query(q, k).results() for q in queries[:3]]):
        assert len(set(ids[i].tolist()) & set(wi.tolist())) >= k - 1


def test_zero_query_zero_rows_and_small_corpus(lib):
    dim, k = 256, 10
    x = rand_rows(600_000, dim, 30)
    x[1000:1100] = 0
    g = build(x, dim, IP)
    queries = np.random.default_rng(31).uniform(-1, 1, (6, dim)).astype(np.float16)
    queries[2] = 0
    want = [g.topk_query(q, k).results() for q in queries]
    check(lib, g, queries, k, want)
    g.free()
    small = build(rand_rows(20_000, dim, 32), dim, COS)   # below the batched path's size: the exact multi-query scan
    want = [small.topk_query(q, k).results() for q in queries]
    check(lib, small, queries, k, want)


def test_concurrent_callers_are_coalesced_into_the_same_passes(lib):
    """more than 16 queued VecSimIndex_TopKQuery callers take a wide pass (FlatIndex::topk_pass_wide -> topk_batch): same replies"""
    import threading
    dim, n, k = 256, 600_000, 10
    g = build(rand_rows(n, dim, 40), dim, COS)
    queries = np.random.default_rng(41).uniform(-1, 1, (64, dim)).astype(np.float16)
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
