"""GPU: growth of the row matrix without copies (grow_buffer.cpp): virtual range + mapped physical chunks, the one
migration out of hipMalloc, re-reservation of a larger virtual range with the same chunks re-mapped, and the plain
realloc path (knob vmm=0) -- results identical throughout."""
import numpy as np
import pytest

import oracle as O
from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu
F32 = V.VecSimType_FLOAT32


def _check(g, seed, n, dim, probes):
    for r in probes:
        if r >= n:
            continue
        q = O.philox_rows(seed, r, 1, dim)[0]
        ids, sc = g.topk_query(q, 1).results()
        assert ids.tolist() == [r + 1] and sc[0] == 0.0, (n, r, ids, sc)
        assert np.array_equal(g.read_rows(r, 1)[0], q)


@pytest.mark.parametrize("vmm", [1, 0])
def test_rows_survive_growth_steps(vmm):
    lib = V.load()
    assert lib.RSGPU_SetTuning(b"vmm", vmm) == 0
    try:
        dim, seed = 1024, 77                       # 4 KiB rows: 65 536 rows = 256 MiB = the mapping threshold
        g = V.VecSimIndex(F32, dim, V.VecSimMetric_L2)
        n = 0
        probes = [0, 1, 40_000, 65_535, 65_536, 99_999, 250_000, 499_999, 700_000]
        for step in (40_000, 30_000, 60_000, 120_000, 250_000):    # 160 MiB -> 280 MiB (migration) -> ... -> 2 GiB
            assert g.add_philox_rows(seed, n, step, n + 1) == step
            n += step
            _check(g, seed, n, dim, probes)
        mem = g.stats_info().memory
        assert mem >= n * dim * 4 and mem <= n * dim * 4 * (1.6 if vmm == 0 else 1.25) + (64 << 20), mem
        v = O.philox_rows(seed, 10 ** 9, 1, dim)[0]
        assert g.add_vector(v, 10 ** 9) == 1       # staged host add on top of the mapped rows
        assert g.topk_query(v, 1).results()[0].tolist() == [10 ** 9]
        assert g.delete_vector(5) == 1             # swap-delete inside mapped memory
        assert g.topk_query(O.philox_rows(seed, 4, 1, dim)[0], 1).results()[0].tolist() != [5]
    finally:
        lib.RSGPU_SetTuning(b"vmm", 1)


def test_a_buffer_that_outgrows_its_virtual_range_is_rebuilt():
    """The virtual range is 64 x the size at mapping time (>= 64 GiB); with the test knob vmm_reserve_factor = 2 the
    range is 2 GiB and growing to 4 GiB rebuilds the buffer in a larger one (the one copying step): contents and
    results survive."""
    lib = V.load()
    assert lib.RSGPU_SetTuning(b"vmm_reserve_factor", 2) == 0
    try:
        dim, seed = 1024, 78
        g = V.VecSimIndex(F32, dim, V.VecSimMetric_L2)
        n = 0
        for step in (70_000, 1_000_000, 300_000):       # 273 MiB (range 2 GiB) -> 4.08 GiB (rebuilt, range 8 GiB) -> 5.2 GiB
            assert g.add_philox_rows(seed, n, step, n + 1) == step
            n += step
            _check(g, seed, n, dim, [0, 69_999, 70_000, 500_000, 1_069_999, 1_070_000, n - 1])
        mem = g.stats_info().memory
        assert n * dim * 4 <= mem <= n * dim * 4 + (1200 << 20)
    finally:
        lib.RSGPU_SetTuning(b"vmm_reserve_factor", 64)
