"""GPU tests of the two top-K selection paths behind VecSimIndex_TopKQuery: the small-K threshold filter
(sample -> tau -> one filter pass -> exact select of the survivors) and the radix levels it falls back to.
Both must return the same exact answer as the oracle order (distance, then storage row)."""
import numpy as np
import pytest
import torch

from redisearch_amd import vecsim as V
from tests.util import close

pytestmark = pytest.mark.gpu
F32 = V.VecSimType_FLOAT32


def device_index(x, metric):
    idx = V.VecSimIndex(F32, x.shape[1], metric)
    torch.cuda.synchronize()
    idx.add_device_rows(x.data_ptr(), x.shape[0], 1)
    return idx


def reference_topk(x, q, k, metric):
    qt = torch.from_numpy(q).to(x.device)
    if metric == V.VecSimMetric_L2:
        d = ((x - qt) ** 2).sum(1)
    else:
        d = 1.0 - x @ qt
    rs, ri = torch.topk(d, min(k, d.numel()), largest=False)
    return ri.cpu().numpy() + 1, rs.cpu().numpy()


@pytest.mark.parametrize("n", [1_000, 20_000, 32_769, 70_001, 262_144, 300_001, 1_000_000])  # single-WG | radix | filter
@pytest.mark.parametrize("k", [1, 10, 32, 33, 128, 1024])  # <= 32: threshold filter; above: radix levels only
def test_filter_path_matches_radix_path_and_reference(n, k):
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(n + k)
    x = torch.rand((n, 32), device=dev, generator=gen) * 2 - 1
    idx = device_index(x, V.VecSimMetric_L2)
    lib = V.load()
    q = np.random.default_rng(k).uniform(-1, 1, 32).astype(np.float32)
    lib.RSGPU_SetTuning(b"filter_select", 1)
    fi, fs = idx.topk_query(q, k).results()
    lib.RSGPU_SetTuning(b"filter_select", 0)
    ri, rs = idx.topk_query(q, k).results()
    lib.RSGPU_SetTuning(b"filter_select", 1)
    assert fi.tolist() == ri.tolist() and fs.tolist() == rs.tolist()      # same kernels' keys: bit-identical
    ti, ts = reference_topk(x, q, k, V.VecSimMetric_L2)
    assert close(fs, ts)
    assert len(set(fi.tolist()) ^ set(ti.tolist())) <= 2                  # fp32 near-ties only


def test_adversarial_orders_and_overflow_fallback():
    dev = torch.device("cuda", 0)
    n, dim, k = 400_000, 8, 13
    ramp = torch.linspace(0, 1, n, device=dev)[:, None].repeat(1, dim)
    q = np.zeros(dim, dtype=np.float32)
    for x, expect in ((ramp, list(range(1, k + 1))),                       # best rows first
                      (ramp.flip(0).contiguous(), list(range(n, n - k, -1)))):  # best rows last
        ids, sc = device_index(x, V.VecSimMetric_L2).topk_query(q, k).results()
        assert ids.tolist() == expect and np.all(np.diff(sc) >= 0)
    # all rows identical: every key passes the filter -> candidate overflow -> radix levels resolve the
    # ties by storage row
    same = torch.ones((n, dim), device=dev)
    ids, sc = device_index(same, V.VecSimMetric_L2).topk_query(q, k).results()
    assert ids.tolist() == list(range(1, k + 1)) and (sc == float(dim)).all()
    # two distinct values, the boundary falls inside a run of equal keys
    two = torch.ones((n, dim), device=dev)
    two[::1000] = 0.5
    ids, sc = device_index(two, V.VecSimMetric_L2).topk_query(q, 450).results()
    assert ids[:400].tolist() == list(range(1, n + 1, 1000)) and ids[400:].tolist() == [2 + i + (i // 999) for i in range(50)]
