"""GPU tests of the opt-in two-stage exact scan (fp16 shadow + error-bounded filter + fp32 re-scoring):
ids AND distances must be bit-identical to the one-stage fp32 scan -- on random data, on tightly clustered data
(where many rows sit inside the error band), on data that overflows the candidate buffer (fallback), and after
deletes / appends (shadow upkeep)."""
import numpy as np
import pytest
import torch

from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu
F32, COS = V.VecSimType_FLOAT32, V.VecSimMetric_Cosine


MODES = [b"shadow16", b"shadow8"]


def pair(x, mode=b"shadow16", metric=COS):
    """the same rows in a plain index and in one with a low-precision shadow (fp16 or int8 + row scales)"""
    lib = V.load()
    out = []
    for shadow in (0, 1):
        lib.RSGPU_SetTuning(mode, shadow)
        idx = V.VecSimIndex(F32, x.shape[1], metric)
        torch.cuda.synchronize()
        idx.add_device_rows(x.data_ptr(), x.shape[0], 1)
        out.append(idx)
    lib.RSGPU_SetTuning(mode, 0)
    return out


def same(plain, shadow, q, k):
    pi, ps = plain.topk_query(q, k).results()
    si, ss = shadow.topk_query(q, k).results()
    assert si.tolist() == pi.tolist() and ss.tolist() == ps.tolist()
    return pi


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("dim,n", [(768, 300_000), (96, 1_000_000), (33, 400_003)])
def test_random_rows_identical(dim, n, mode):
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(dim + n)
    x = torch.rand((n, dim), device=dev, generator=gen) * 2 - 1
    plain, shadow = pair(x, mode)
    rng = np.random.default_rng(dim)
    for k in (1, 10, 16, 17, 100):
        for _ in range(4):
            same(plain, shadow, rng.uniform(-1, 1, dim).astype(np.float32), k)
    # a query that IS a stored row (distance ~0, others far) and its negation
    q = x[12345].cpu().numpy()
    assert same(plain, shadow, q, 10)[0] == 12346
    same(plain, shadow, -q, 10)


@pytest.mark.parametrize("mode", MODES)
def test_clustered_rows_inside_the_error_band(mode):
    # 300k rows = 300 tight clusters: thousands of rows within 1e-3 of the k-th distance; still exact
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    centers = torch.rand((300, 64), device=dev, generator=gen) * 2 - 1
    x = centers.repeat_interleave(1000, 0) + 1e-3 * (torch.rand((300_000, 64), device=dev, generator=gen) - 0.5)
    plain, shadow = pair(x, mode)
    for c in (0, 7, 299):
        same(plain, shadow, centers[c].cpu().numpy(), 10)
        same(plain, shadow, (centers[c] + 0.01).cpu().numpy(), 16)


@pytest.mark.parametrize("mode", MODES)
def test_candidate_overflow_falls_back_to_the_full_scan(mode):
    # every row identical: all of them pass the filter -> overflow -> one-stage path, ties by storage row
    x = torch.ones((300_000, 16), device="cuda")
    plain, shadow = pair(x, mode)
    ids, sc = shadow.topk_query(np.ones(16, np.float32), 10).results()
    assert ids.tolist() == list(range(1, 11))
    same(plain, shadow, np.ones(16, np.float32), 10)


@pytest.mark.parametrize("mode", MODES)
def test_shadow_follows_deletes_and_appends(mode):
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(9)
    x = torch.rand((300_000, 48), device=dev, generator=gen) * 2 - 1
    plain, shadow = pair(x, mode)
    q = x[777].cpu().numpy()
    assert same(plain, shadow, q, 5)[0] == 778
    for idx in (plain, shadow):
        assert idx.delete_vector(778) == 1                      # the last row moves into the hole
        assert idx.add_vector(x[777].cpu().numpy() * 3.0, 900_001) == 1   # same direction, staged host add
    ids = same(plain, shadow, q, 5)
    assert ids[0] == 900_001 and 778 not in ids.tolist()
    extra = torch.rand((50_000, 48), device=dev, generator=gen) * 2 - 1
    torch.cuda.synchronize()
    for idx in (plain, shadow):
        idx.add_device_rows(extra.data_ptr(), 50_000, 1_000_000)
    same(plain, shadow, extra[5].cpu().numpy(), 10)


def test_spiky_rows_make_the_int8_band_wide_but_results_stay_exact():
    # a few one-hot rows force the largest row scale to 1/127: the int8 error band swallows most rows, the
    # candidate buffer overflows and the query is answered by the full scan -- still identical
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(31)
    x = torch.rand((300_000, 128), device=dev, generator=gen) * 2 - 1
    x[1000:1010] = 0
    x[1000:1010, 5] = 1.0
    plain, shadow = pair(x, b"shadow8")
    rng = np.random.default_rng(3)
    for _ in range(3):
        same(plain, shadow, rng.uniform(-1, 1, 128).astype(np.float32), 10)
    q = np.zeros(128, np.float32)
    q[5] = 1.0
    assert set(same(plain, shadow, q, 10).tolist()) == set(range(1001, 1011))


# ---- the int8 shadow under IP and L2 (rows of any norm: the band carries |q| and the largest |x|), K up to 1024 -------
IP, L2 = V.VecSimMetric_IP, V.VecSimMetric_L2


def took_the_shadow_scan(idx, q, k, n, dim):
    """one query with the scan profile on: the scan that ran read dim + 8 bytes per row (int8 row + scale + norm)"""
    lib = V.load()
    lib.RSGPU_ResetProfile()
    lib.RSGPU_SetProfiling(1)
    idx.topk_query(q, k)
    lib.RSGPU_SetProfiling(0)
    launches, _, by = V.scan_profile()
    return launches >= 1 and by == launches * n * (dim + 8)


@pytest.mark.parametrize("metric", [IP, L2])
@pytest.mark.parametrize("dim,n", [(768, 300_000), (96, 1_000_000), (33, 400_003)])
def test_int8_shadow_ip_l2_identical(dim, n, metric):
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(dim + n + metric)
    # row norms spread over a factor of ~3 (0.7 .. 2 x a U(-1,1) row)
    x = (torch.rand((n, dim), device=dev, generator=gen) * 2 - 1) * (0.7 + 1.3 * torch.rand((n, 1), device=dev, generator=gen))
    plain, shadow = pair(x, b"shadow8", metric)
    rng = np.random.default_rng(dim + metric)
    for k in (1, 10, 16, 17, 100, 500, 1024):
        for _ in range(3):
            same(plain, shadow, (rng.uniform(-1, 1, dim) * rng.uniform(0.2, 4)).astype(np.float32), k)
    assert took_the_shadow_scan(shadow, rng.uniform(-1, 1, dim).astype(np.float32), 10, n, dim)
    assert not took_the_shadow_scan(plain, rng.uniform(-1, 1, dim).astype(np.float32), 10, n, dim)
    q = x[12345].cpu().numpy()
    ids = same(plain, shadow, q, 10)
    if metric == L2:
        assert ids[0] == 12346
    same(plain, shadow, -q, 10)
    same(plain, shadow, np.zeros(dim, np.float32), 10)      # zero query: no scale, answered by the fp32 scan


@pytest.mark.parametrize("metric", [COS, IP, L2])
def test_int8_shadow_large_k_and_cosine(metric):
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(77 + metric)
    x = torch.rand((500_000, 128), device=dev, generator=gen) * 2 - 1
    for mode in MODES if metric == COS else [b"shadow8"]:
        plain, shadow = pair(x, mode, metric)
        rng = np.random.default_rng(metric)
        for k in (129, 300, 1024, 1025, 2000):              # above 1024 the one-stage path answers
            same(plain, shadow, rng.uniform(-1, 1, 128).astype(np.float32), k)


@pytest.mark.parametrize("metric", [IP, L2])
def test_int8_shadow_l2_ip_clusters_deletes_and_nonfinite_rows(metric):
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5 + metric)
    centers = (torch.rand((300, 64), device=dev, generator=gen) * 2 - 1) * 3
    x = centers.repeat_interleave(1000, 0) + 1e-3 * (torch.rand((300_000, 64), device=dev, generator=gen) - 0.5)
    plain, shadow = pair(x, b"shadow8", metric)
    for c in (0, 7, 299):
        same(plain, shadow, centers[c].cpu().numpy(), 10)
        same(plain, shadow, (centers[c] + 0.01).cpu().numpy(), 16)
    for idx in (plain, shadow):
        assert idx.delete_vector(7001) == 1
        assert idx.add_vector(centers[7].cpu().numpy() * 10.0, 900_001) == 1    # the new largest norm widens the band
    ids = same(plain, shadow, centers[7].cpu().numpy(), 5)
    assert 7001 not in ids.tolist()
    if metric == IP:
        assert ids[0] == 900_001
    # a row with an infinite element: no finite band exists; the shadow index answers with the fp32 scan from then on
    bad = centers[3].cpu().numpy().copy()
    bad[0] = np.inf
    for idx in (plain, shadow):
        assert idx.add_vector(bad, 900_002) == 1
    pi, ps = plain.topk_query(centers[9].cpu().numpy(), 5).results()
    si, ss = shadow.topk_query(centers[9].cpu().numpy(), 5).results()
    assert si.tolist() == pi.tolist() and np.array_equal(ss, ps, equal_nan=True)
    assert not took_the_shadow_scan(shadow, centers[9].cpu().numpy(), 5, 300_001, 64)
