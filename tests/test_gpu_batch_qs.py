"""GPU tests of the query-stationary filter pass of the batched path (gemm_qs_kernels.hip): every supported
row width, both 16-bit types, ragged tail tiles, and the sub-list overflow fallback; each case is compared with
the tiled GEMM filter (gemm_qs=0, must be IDENTICAL: same MFMA arithmetic order per (query,row)) and with a
torch fp32 reference of the same op on the same 16-bit data."""
import numpy as np
import pytest
import torch

from redisearch_amd import vecsim as V
from tests.util import close

pytestmark = pytest.mark.gpu
F16, BF16 = V.VecSimType_FLOAT16, V.VecSimType_BFLOAT16
IP, COS = V.VecSimMetric_IP, V.VecSimMetric_Cosine


def both_paths(g, queries, k):
    lib = V.load()
    lib.RSGPU_SetTuning(b"gemm_qs", 1)
    a = g.topk_batch(queries, k)
    lib.RSGPU_SetTuning(b"gemm_qs", 0)
    b = g.topk_batch(queries, k)
    lib.RSGPU_SetTuning(b"gemm_qs", 1)
    return a, b


@pytest.mark.parametrize("dim,n", [(128, 700_003), (256, 600_000), (384, 550_017), (512, 530_000), (768, 525_001)])
@pytest.mark.parametrize("vtype", [F16, BF16])
def test_qs_filter_matches_tiled_filter_and_reference(dim, n, vtype):
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(dim + n)
    tdt = torch.float16 if vtype == F16 else torch.bfloat16
    x = (torch.rand((n, dim), device=dev, generator=gen) * 2 - 1).to(tdt)
    g = V.VecSimIndex(vtype, dim, IP)
    torch.cuda.synchronize()
    g.add_device_rows(x.data_ptr(), n, 1)
    b, k = 200, 50                                     # < 256 queries: padded query rows must stay silent
    qt = (torch.rand((b, dim), device=dev, generator=gen) * 2 - 1).to(tdt)
    queries = qt.float().cpu().numpy()                # exactly representable: the wrapper re-rounds to the type
    (ids, sc, cnt), (ids0, sc0, cnt0) = both_paths(g, queries, k)
    assert (cnt == k).all() and (cnt0 == k).all()
    assert np.array_equal(ids, ids0) and np.array_equal(sc, sc0)
    ref = 1.0 - qt.float() @ x.float().T
    rs, ri = torch.topk(ref, k, dim=1, largest=False)
    rs, ri = rs.cpu().numpy(), ri.cpu().numpy() + 1
    for i in range(b):
        assert close(sc[i], rs[i])
        assert len(set(ids[i].tolist()) ^ set(ri[i].tolist())) <= 4


def test_qs_sorted_corpus_overflows_and_falls_back():
    # best rows last and all in a few tiles: the per-workgroup sub-lists overflow, the query is redone exactly
    dim, n, k = 128, 600_000, 10
    base = np.linspace(-1, 1, n, dtype=np.float32)[:, None] * np.ones((1, dim), dtype=np.float32)
    t = torch.from_numpy(base.astype(np.float16)).cuda()
    g = V.VecSimIndex(F16, dim, IP)
    torch.cuda.synchronize()
    g.add_device_rows(t.data_ptr(), n, 1)
    q = np.ones((3, dim), dtype=np.float16)
    q[1] *= -1
    ids, sc, cnt = g.topk_batch(q, k)
    for i in range(3):
        si, ss = g.topk_query(q[i], k).results()
        assert ids[i].tolist() == si.tolist() and close(sc[i], ss)


def test_qs_cosine_and_large_k():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    n, dim, b, k = 540_000, 256, 256, 1000
    x = (torch.rand((n, dim), device=dev, generator=gen) * 2 - 1).to(torch.float16)
    g = V.VecSimIndex(F16, dim, COS)
    torch.cuda.synchronize()
    g.add_device_rows(x.data_ptr(), n, 1)
    queries = np.random.default_rng(6).uniform(-1, 1, (b, dim)).astype(np.float16)
    (ids, sc, cnt), (ids0, sc0, cnt0) = both_paths(g, queries, k)
    assert (cnt == k).all() and np.array_equal(ids, ids0) and np.array_equal(sc, sc0)
    for i in (0, 77, 255):
        si, ss = g.topk_query(queries[i], k).results()
        assert close(sc[i], ss) and len(set(si.tolist()) ^ set(ids[i].tolist())) <= 6


def test_many_batches_are_pipelined_and_identical_to_single_batches():
    # 700 queries = 3 passes over the corpus on two alternating slots; same answers as batch-by-batch calls
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    n, dim, k = 600_000, 128, 20
    x = (torch.rand((n, dim), device=dev, generator=gen) * 2 - 1).to(torch.float16)
    g = V.VecSimIndex(F16, dim, IP)
    torch.cuda.synchronize()
    g.add_device_rows(x.data_ptr(), n, 1)
    queries = np.random.default_rng(12).uniform(-1, 1, (700, dim)).astype(np.float16)
    ids, sc, cnt = g.topk_batch(queries, k)
    for lo in (0, 256, 512):
        i2, s2, c2 = g.topk_batch(queries[lo:lo + 256], k)
        assert np.array_equal(ids[lo:lo + 256], i2) and np.array_equal(sc[lo:lo + 256], s2) and np.array_equal(cnt[lo:lo + 256], c2)
