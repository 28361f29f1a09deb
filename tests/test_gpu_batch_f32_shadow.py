"""GPU tests of the batched path over FLOAT32 cosine indexes that carry an fp16 shadow (opt-in "shadow16"): the MFMA
filter pass reads the shadow with an error-band-wide threshold, the survivors are re-scored from the fp32 rows with the
single-query scan's arithmetic.  The result must be BIT-IDENTICAL to one VecSimIndex_TopKQuery per query (ids and
distances), and the batched path must really have been taken (one profiled device pipeline per 256 queries)."""
import numpy as np
import pytest
import torch

from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu
F32, COS = V.VecSimType_FLOAT32, V.VecSimMetric_Cosine


@pytest.fixture
def shadow16():
    lib = V.load()
    lib.RSGPU_SetTuning(b"shadow16", 1)
    yield lib
    lib.RSGPU_SetTuning(b"shadow16", 0)
    lib.RSGPU_SetTuning(b"two_stage", 1)


def build(x, dim):
    g = V.VecSimIndex(F32, dim, COS)
    torch.cuda.synchronize()
    g.add_device_rows(x.data_ptr(), x.shape[0], 1)
    return g


def batched_and_single(lib, g, queries, k):
    lib.RSGPU_ResetProfile()
    lib.RSGPU_SetProfiling(1)
    ids, sc, cnt = g.topk_batch(queries, k)
    lib.RSGPU_SetProfiling(0)
    launches, _, _ = V.scan_profile()
    single = [g.topk_query(q, k).results() for q in queries]
    return ids, sc, cnt, launches, single


@pytest.mark.parametrize("dim,n", [(768, 530_001), (512, 540_000), (384, 600_017), (256, 700_000), (128, 1_000_003)])
@pytest.mark.parametrize("k", [10, 100])
def test_batched_f32_over_shadow_is_bit_identical_to_single_queries(shadow16, dim, n, k):
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(dim * 7 + k)
    x = torch.rand((n, dim), device=dev, generator=gen) * 2 - 1
    g = build(x, dim)
    b = 300                                             # two passes, the second one padded
    queries = np.random.default_rng(dim + k).uniform(-1, 1, (b, dim)).astype(np.float32)
    ids, sc, cnt, launches, single = batched_and_single(shadow16, g, queries, k)
    assert launches == 2, "the batched path was not taken (%d profiled launches)" % launches
    assert (cnt == k).all()
    for i in range(b):
        si, ss = single[i]
        assert ids[i].tolist() == si.tolist(), i
        assert sc[i].tolist() == ss.tolist(), i
    # and the single-query answers are the exact fp32 ones (two_stage off = plain fp32 scan)
    shadow16.RSGPU_SetTuning(b"two_stage", 0)
    for i in (0, 150, 299):
        pi, ps = g.topk_query(queries[i], k).results()
        assert ids[i].tolist() == pi.tolist() and sc[i].tolist() == ps.tolist()
    shadow16.RSGPU_SetTuning(b"two_stage", 1)


def test_clustered_rows_overflow_the_band_and_fall_back_exactly(shadow16):
    # every row within 1e-3 of one direction: the whole corpus sits inside the fp16 error band, the candidate
    # lists overflow, the host redoes the queries on the single-query path -- the answer must not change
    dev = torch.device("cuda", 0)
    dim, n, k = 128, 600_000, 10
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    centre = torch.rand((1, dim), device=dev, generator=gen) * 2 - 1
    x = centre + 1e-3 * (torch.rand((n, dim), device=dev, generator=gen) * 2 - 1)
    g = build(x, dim)
    queries = (centre.cpu().numpy() + 1e-3 * np.random.default_rng(4).uniform(-1, 1, (5, dim))).astype(np.float32)
    ids, sc, cnt = g.topk_batch(queries, k)
    shadow16.RSGPU_SetTuning(b"two_stage", 0)
    for i in range(5):
        pi, ps = g.topk_query(queries[i], k).results()
        assert ids[i].tolist() == pi.tolist() and sc[i].tolist() == ps.tolist()


def test_small_or_unsupported_shapes_loop_over_single_queries(shadow16):
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(9)
    for dim, n in ((96, 300_000), (128, 100_000)):      # no QS shape for 96 halves; corpus below the batched cut-over
        x = torch.rand((n, dim), device=dev, generator=gen) * 2 - 1
        g = build(x, dim)
        queries = np.random.default_rng(dim).uniform(-1, 1, (7, dim)).astype(np.float32)
        ids, sc, cnt = g.topk_batch(queries, 10)
        for i in range(7):
            si, ss = g.topk_query(queries[i], 10).results()
            assert ids[i].tolist() == si.tolist() and sc[i].tolist() == ss.tolist()
