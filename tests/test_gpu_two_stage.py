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


def pair(x):
    """the same rows in a plain index and in one with the fp16 shadow"""
    lib = V.load()
    out = []
    for shadow in (0, 1):
        lib.RSGPU_SetTuning(b"shadow16", shadow)
        idx = V.VecSimIndex(F32, x.shape[1], COS)
        torch.cuda.synchronize()
        idx.add_device_rows(x.data_ptr(), x.shape[0], 1)
        out.append(idx)
    lib.RSGPU_SetTuning(b"shadow16", 0)
    return out


def same(plain, shadow, q, k):
    pi, ps = plain.topk_query(q, k).results()
    si, ss = shadow.topk_query(q, k).results()
    assert si.tolist() == pi.tolist() and ss.tolist() == ps.tolist()
    return pi


@pytest.mark.parametrize("dim,n", [(768, 300_000), (96, 1_000_000), (33, 400_003)])
def test_random_rows_identical(dim, n):
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(dim + n)
    x = torch.rand((n, dim), device=dev, generator=gen) * 2 - 1
    plain, shadow = pair(x)
    rng = np.random.default_rng(dim)
    for k in (1, 10, 16, 100):
        for _ in range(4):
            same(plain, shadow, rng.uniform(-1, 1, dim).astype(np.float32), k)
    # a query that IS a stored row (distance ~0, others far) and its negation
    q = x[12345].cpu().numpy()
    assert same(plain, shadow, q, 10)[0] == 12346
    same(plain, shadow, -q, 10)


def test_clustered_rows_inside_the_error_band():
    # 300k rows = 300 tight clusters: thousands of rows within 1e-3 of the k-th distance; still exact
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    centers = torch.rand((300, 64), device=dev, generator=gen) * 2 - 1
    x = centers.repeat_interleave(1000, 0) + 1e-3 * (torch.rand((300_000, 64), device=dev, generator=gen) - 0.5)
    plain, shadow = pair(x)
    for c in (0, 7, 299):
        same(plain, shadow, centers[c].cpu().numpy(), 10)
        same(plain, shadow, (centers[c] + 0.01).cpu().numpy(), 16)


def test_candidate_overflow_falls_back_to_the_full_scan():
    # every row identical: all of them pass the filter -> overflow -> one-stage path, ties by storage row
    x = torch.ones((300_000, 16), device="cuda")
    plain, shadow = pair(x)
    ids, sc = shadow.topk_query(np.ones(16, np.float32), 10).results()
    assert ids.tolist() == list(range(1, 11))
    same(plain, shadow, np.ones(16, np.float32), 10)


def test_shadow_follows_deletes_and_appends():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(9)
    x = torch.rand((300_000, 48), device=dev, generator=gen) * 2 - 1
    plain, shadow = pair(x)
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
