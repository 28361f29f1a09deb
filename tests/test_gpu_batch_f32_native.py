"""GPU tests of the matrix-core passes over plain FLOAT32 indexes -- no stored shadow (round 4; gemm_qs_kernels.hip
`gemm_qs_f32_kernel`, batch_query.cpp `via_f32`): the fp32 tiles go global -> LDS by DMA as they are and are rounded to bf16
on their way to the MFMA; every bound is widened by the rounding band and the survivors are re-scored from the same fp32
rows with the single-query scan's arithmetic.  The replies must be BIT-IDENTICAL to one VecSimIndex_TopKQuery per query
(ids and distances) -- cosine, L2 and IP, every row width the kernel has, ragged tail tiles, rows whose norms differ by a factor
of 16, a row with a huge norm, clustered rows that overflow the band (fallback), deletes and appends, a non-finite query and
a non-finite row -- and wherever the route applies the passes must really have been taken (profiled batch launches, no
multi-query scan passes).  One case is held to the CPU oracle directly, one to a torch fp32 reference of the same op."""
import numpy as np
import pytest
import torch

from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu
F32, COS, L2, IP = V.VecSimType_FLOAT32, V.VecSimMetric_Cosine, V.VecSimMetric_L2, V.VecSimMetric_IP


@pytest.fixture(autouse=True)
def bf16_in_flight_route():
    """This file holds the bf16-in-flight form; since round 6 IP / cosine default to the int8-in-flight one (knob gemm_qs_f8,
    tests/test_gpu_batch_f8.py), so the knob is off here for the indexes these tests build and query."""
    lib = V.load()
    lib.RSGPU_SetTuning(b"gemm_qs_f8", 0)
    yield
    lib.RSGPU_SetTuning(b"gemm_qs_f8", 1)


def rows(n, dim, seed, spread=False):
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    x = torch.rand((n, dim), device=dev, generator=gen) * 2 - 1
    if spread:                                         # norms from 1/4 to 4 times the typical one
        x *= torch.exp2(torch.rand((n, 1), device=dev, generator=gen) * 4 - 2)
    return x


def build(x, dim, metric):
    g = V.VecSimIndex(F32, dim, metric)
    torch.cuda.synchronize()
    g.add_device_rows(x.data_ptr(), x.shape[0], 1)
    return g


def batched(g, queries, k, expect_launches):
    lib = V.load()
    before = V.coalesce_stats()["mq_passes"]
    lib.RSGPU_ResetProfile()
    lib.RSGPU_SetProfiling(1)
    out = g.topk_batch(queries, k)
    lib.RSGPU_SetProfiling(0)
    launches = V.scan_profile()[0]
    mq = V.coalesce_stats()["mq_passes"] - before
    if expect_launches is not None:
        assert (launches, mq) == (expect_launches, 0), "the matrix-core passes were not taken (%d launches, %d scan passes)" % (launches, mq)
    return out


def same_as_singles(g, queries, k, got, which=None):
    ids, sc, cnt = got
    for i in (range(len(queries)) if which is None else which):
        si, ss = g.topk_query(queries[i], k).results()
        assert cnt[i] == len(si), i
        assert ids[i][: cnt[i]].tolist() == si.tolist(), i
        assert np.array_equal(sc[i][: cnt[i]], ss, equal_nan=True), i      # (a NaN query: NaN distances on both sides)


@pytest.mark.parametrize("dim,n", [(768, 530_001), (512, 540_000), (384, 600_017), (256, 700_000), (128, 1_000_003)])
@pytest.mark.parametrize("k", [10, 100])
def test_cosine_pass_is_bit_identical_to_single_queries(dim, n, k):
    x = rows(n, dim, dim * 7 + k)
    g = build(x, dim, COS)
    try:
        b = 300                                        # two passes, the second one padded
        queries = np.random.default_rng(dim + k).uniform(-1, 1, (b, dim)).astype(np.float32)
        got = batched(g, queries, k, 2)
        assert (got[2] == k).all()
        same_as_singles(g, queries, k, got)
    finally:
        g.free()


@pytest.mark.parametrize("dim,n", [(768, 525_001), (512, 530_000), (384, 550_017), (256, 600_000), (128, 700_003)])
def test_l2_pass_is_bit_identical_to_single_queries(dim, n):
    x = rows(n, dim, dim + n, spread=True)
    g = build(x, dim, L2)
    try:
        b, k = 300, 50
        qt = rows(b, dim, dim + 1, spread=True)
        queries = qt.cpu().numpy()
        got = batched(g, queries, k, 2)
        assert (got[2] == k).all()
        same_as_singles(g, queries, k, got)
        # ... and against a torch fp32 reference of the same op
        for i in (0, 150, 299):
            ref = ((x - qt[i][None, :]) ** 2).sum(dim=1)
            rs, ri = torch.topk(ref, k, largest=False)
            assert np.allclose(got[1][i], rs.cpu().numpy().astype(np.float64), rtol=2e-4, atol=1e-3)   # (fp32 summation orders)
            assert len(set(got[0][i].tolist()) ^ set((ri.cpu().numpy() + 1).tolist())) <= 4
    finally:
        g.free()


@pytest.mark.parametrize("dim,n", [(768, 525_001), (256, 600_000), (128, 700_003)])
def test_ip_pass_is_bit_identical_to_single_queries(dim, n):
    """IP over rows that are not normalised: the band comes from the largest row norm of the index (a per-query widening)"""
    x = rows(n, dim, dim + n + 1, spread=True)
    g = build(x, dim, IP)
    try:
        b, k = 300, 50
        qt = rows(b, dim, dim + 2, spread=True)
        queries = qt.cpu().numpy()
        got = batched(g, queries, k, 2)
        assert (got[2] == k).all()
        same_as_singles(g, queries, k, got)
        for i in (0, 299):
            ref = 1.0 - (x @ qt[i])
            rs, ri = torch.topk(ref, k, largest=False)
            assert np.allclose(got[1][i], rs.cpu().numpy().astype(np.float64), rtol=2e-4, atol=2e-3)
            assert len(set(got[0][i].tolist()) ^ set((ri.cpu().numpy() + 1).tolist())) <= 4
        q2 = queries[:6].copy()
        q2[2, 1] = np.inf                                   # a non-finite query: the exact scan's
        same_as_singles(g, q2, k, g.topk_batch(q2, k))
    finally:
        g.free()


def test_against_the_cpu_oracle_directly():
    import oracle as O
    dim, n, k = 128, 530_000, 10
    x = rows(n, dim, 91)
    g = build(x, dim, COS)
    try:
        queries = np.random.default_rng(92).uniform(-1, 1, (6, dim)).astype(np.float32)
        got = batched(g, queries, k, 1)
        o = O.FlatIndex(O.F32, dim, O.COSINE)
        o.add_bulk(x.cpu().numpy())
        for i in range(6):
            oi, os_ = o.topk(queries[i], k)
            assert got[0][i].tolist() == oi.tolist(), i
            assert np.max(np.abs(got[1][i] - os_)) <= 1e-4
    finally:
        g.free()


def test_huge_row_widens_only_its_own_band_and_clusters_fall_back():
    dev = torch.device("cuda", 0)
    dim, n, k = 256, 600_000, 10
    x = rows(n, dim, 5, spread=True)
    x[12345] *= 1.0e15                                 # |x|^2 ~ 1e32: its own band is astronomic, nobody else's changes
    g = build(x, dim, L2)
    try:
        queries = rows(9, dim, 6, spread=True).cpu().numpy()
        got = batched(g, queries, k, 1)
        same_as_singles(g, queries, k, got)
    finally:
        g.free()
    # every row within 1e-3 of one direction: the whole corpus sits inside the bf16 band, the candidate lists overflow, the
    # host redoes the queries on the single-query path -- the answer must not change
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    dim, n = 128, 600_000
    centre = torch.rand((1, dim), device=dev, generator=gen) * 2 - 1
    x = centre + 1e-3 * (torch.rand((n, dim), device=dev, generator=gen) * 2 - 1)
    g = build(x, dim, COS)
    try:
        queries = (centre.cpu().numpy() + 1e-3 * np.random.default_rng(4).uniform(-1, 1, (5, dim))).astype(np.float32)
        got = g.topk_batch(queries, k)
        same_as_singles(g, queries, k, got)
    finally:
        g.free()


def test_deletes_appends_and_non_finite_inputs():
    dim, n, k = 256, 560_000, 20
    x = rows(n, dim, 17, spread=True)
    g = build(x, dim, L2)
    try:
        queries = rows(40, dim, 18, spread=True).cpu().numpy()
        same_as_singles(g, queries, k, batched(g, queries, k, 1))
        rng = np.random.default_rng(19)
        for lab in rng.choice(n, 300, replace=False):  # deletes move the last row into the hole: norms below the built prefix
            g.delete_vector(int(lab) + 1)
        extra = rows(5_000, dim, 20, spread=True)
        torch.cuda.synchronize()
        g.add_device_rows(extra.data_ptr(), extra.shape[0], n + 1)
        same_as_singles(g, queries, k, batched(g, queries, k, 1))
        q2 = queries.copy()
        q2[3, 7] = np.inf                               # a non-finite query is left to the exact scan
        q2[5, 0] = np.nan
        same_as_singles(g, q2, k, g.topk_batch(q2, k))
        bad = torch.zeros((1, dim), device="cuda")
        bad[0, 3] = float("inf")
        torch.cuda.synchronize()
        g.add_device_rows(bad.data_ptr(), 1, n + 10_000)   # a non-finite row: the route is refused, the answers stay exact
        same_as_singles(g, queries, k, g.topk_batch(queries, k), which=range(8))
    finally:
        g.free()


def test_knob_off_and_unsupported_shapes_take_the_exact_scans():
    lib = V.load()
    x = rows(530_000, 128, 23)
    g = build(x, 128, COS)
    try:
        queries = np.random.default_rng(24).uniform(-1, 1, (20, 128)).astype(np.float32)
        lib.RSGPU_SetTuning(b"gemm_qs_f32", 0)
        before = V.coalesce_stats()["mq_passes"]
        got = g.topk_batch(queries, 10)
        assert V.coalesce_stats()["mq_passes"] > before        # sixteen queries per exact pass
        same_as_singles(g, queries, 10, got)
        lib.RSGPU_SetTuning(b"gemm_qs_f32", 1)                 # eight waves x 32 queries: the A/B shape gives the same answers
        same_as_singles(g, queries, 10, batched(g, queries, 10, 1))
    finally:
        lib.RSGPU_SetTuning(b"gemm_qs_f32", 2)
        g.free()
    for dim, n, metric in ((96, 600_000, COS), (128, 100_000, COS), (1024, 530_000, IP)):   # no shape / below the cut-over / 1024 wide
        x = rows(n, dim, dim + 1)
        g = build(x, dim, metric)
        try:
            queries = np.random.default_rng(dim).uniform(-1, 1, (7, dim)).astype(np.float32)
            same_as_singles(g, queries, 10, g.topk_batch(queries, 10))
        finally:
            g.free()
