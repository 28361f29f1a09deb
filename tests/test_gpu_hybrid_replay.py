"""GPU twin of tests/test_hybrid_replay_cpu.py: the reference's HybridIterator (replayed, with the reference's own
min-max heap where oracle/_ref ships it) drives the MI355X FLAT index through the VecSim C ABI and must reproduce the
reference's end-to-end expectations (tests/pytests/test_vecsim.py:963-1038, :1362-1396, :1583-1643)."""
import numpy as np
import pytest
import torch

import oracle as O
from redisearch_amd import vecsim as V
from tests import hybrid_replay as H

pytestmark = pytest.mark.gpu


def ramp_index(n, dim, vtype, tdt):
    g = V.VecSimIndex(vtype, dim, V.VecSimMetric_L2)
    t = torch.arange(1, n + 1, dtype=tdt, device="cuda")[:, None].repeat(1, dim).contiguous()
    torch.cuda.synchronize()
    g.add_device_rows(t.data_ptr(), n, 1)
    return g


def test_batches_mode_with_text_kats():
    n, dim, k = 6000, 2, 10
    g = ramp_index(n, dim, V.VecSimType_FLOAT64, torch.float64)
    q = np.full(dim, float(n))
    index = H.GpuIndex(g)
    it = H.HybridReplay(index, q, k, H.IdListChild(range(1, n + 1)))
    assert it.mode == H.HYBRID_BATCHES
    assert it.results() == [(n - i, float(dim * i * i)) for i in range(k)]
    it = H.HybridReplay(index, q, k, H.IdListChild(range(5, n + 1, 5)))
    assert it.results() == [(n - 5 * i, float(dim * (5 * i) ** 2)) for i in range(k)] and it.mode == H.HYBRID_BATCHES
    it = H.HybridReplay(index, q, k, H.IdListChild([], estimate=1200))
    assert it.results() == [] and it.mode == H.HYBRID_BATCHES_TO_ADHOC_BF and it.num_iterations == 1
    keep = [i for i in range(1, n + 1) if i % 5]
    exp = [(n - i, float(dim * i * i)) for i in range(13) if (n - i) % 5][:k]
    assert H.HybridReplay(index, q, k, H.IdListChild(keep)).results() == exp
    it = H.HybridReplay(index, q, k, None)
    assert it.mode == H.STANDARD_KNN and it.results() == [(n - i, float(dim * i * i)) for i in range(k)]


def test_adhoc_bf_mode_kat_and_policy_agreement():
    n, dim, k = 100, 128, 10
    index = H.GpuIndex(ramp_index(n, dim, V.VecSimType_FLOAT32, torch.float32))
    q = np.full(dim, float(n), dtype=np.float32)
    it = H.HybridReplay(index, q, k, H.IdListChild(range(10, n + 1, 10)), policy=H.HYBRID_ADHOC_BF)
    assert it.results() == [(n - 10 * j, float(dim * (10 * j) ** 2)) for j in range(k)]
    # random data: the GPU index driven by the iterator == the oracle index driven by the same iterator
    rng = np.random.default_rng(1583)
    n, dim = 3000, 6
    data = rng.standard_normal((n, dim)).astype(np.float32)
    g = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2)
    t = torch.from_numpy(data).cuda()
    torch.cuda.synchronize()
    g.add_device_rows(t.data_ptr(), n, 1)
    o = O.FlatIndex(O.F32, dim, O.L2)
    o.add_bulk(data)
    qv = rng.standard_normal(dim).astype(np.float32)
    child = sorted(rng.choice(np.arange(1, n + 1), 700, replace=False).tolist())
    for policy, bs in ((0, 0), (H.HYBRID_BATCHES, 7), (H.HYBRID_ADHOC_BF, 0)):
        got = H.HybridReplay(H.GpuIndex(g), qv, k, H.IdListChild(child), policy=policy, batch_size=bs).results()
        want = H.HybridReplay(H.OracleIndex(o), qv, k, H.IdListChild(child), policy=policy, batch_size=bs).results()
        assert [i for i, _ in got] == [i for i, _ in want]
        assert np.allclose([d for _, d in got], [d for _, d in want], rtol=1e-5, atol=1e-6)
