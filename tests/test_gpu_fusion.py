"""GPU parity of the FT.HYBRID fusion epilogue (fusion_kernels.hip via RSGPU_HybridFuse) against the oracle:
bit-identical fp64 scores and identical order, on the reference's KATs and on seeded random lists."""
import numpy as np
import pytest

import oracle as O
from redisearch_amd import search as S

pytestmark = pytest.mark.gpu


def same(args, **kw):
    gi, gs = S.hybrid_fuse(*args, **kw)
    oi, os_ = O.hybrid_fuse(*args, **kw)
    assert gi.tolist() == oi.tolist() and gs.tolist() == os_.tolist()
    return gi, gs


def test_reference_kats_through_the_device():
    ids, sc = same((S.RRF, [1, 2, 3], [0.9, 0.7, 0.5], [2, 3, 4, 5], [0.8, 0.6, 0.4, 0.2], 5))
    assert ids.tolist() == [2, 3, 1, 4, 5] and sc[0] == 1 / 62 + 1 / 61
    ids, sc = same((S.LINEAR, [7], [2.0], [7], [4.0], 10), weights=(0.3, 0.7))
    assert sc.tolist() == [0.3 * 2.0 + 0.7 * 4.0]
    same((S.RRF, [11, 12], [5.0, 4.0], [21, 22], [0.1, 0.2], 10))
    same((S.RRF, [1, 2, 3, 4], [4, 3, 2, 1], [4, 3, 2, 1], [.1, .2, .3, .4], 2))
    assert len(S.hybrid_fuse(S.RRF, [], [], [], [], 20)[0]) == 0
    same((S.RRF, [], [], [9, 8], [0.5, 0.7], 20))


@pytest.mark.parametrize("scoring", [S.RRF, S.LINEAR])
@pytest.mark.parametrize("na,nb,window", [(20, 20, 20), (100, 37, 50), (1000, 1000, 1000), (4096, 4096, 4096), (5000, 10, 4096)])
@pytest.mark.parametrize("metric", [-1, O.L2, O.COSINE])
def test_random_lists(scoring, na, nb, window, metric):
    rng = np.random.default_rng(na * 7 + nb + window + scoring)
    universe = rng.permutation(np.arange(1, 3 * max(na, nb) + 1))
    a_ids = universe[:na]
    b_ids = rng.permutation(universe[: 2 * max(na, nb)])[:nb]           # about half of b is shared with a
    a_sc = np.sort(rng.uniform(0, 10, na))[::-1].round(2)               # rounded: plenty of equal fused scores
    b_sc = np.sort(rng.uniform(0, 2, nb)).round(2)
    gi, gs = same((scoring, a_ids, a_sc, b_ids, b_sc, window), constant=60.0, weights=(0.35, 0.65), metric=metric)
    assert len(gi) == len(set(a_ids[:window].tolist()) | set(b_ids[:window].tolist()))
    top = S.hybrid_fuse(scoring, a_ids, a_sc, b_ids, b_sc, window, top_n=10, weights=(0.35, 0.65), metric=metric)
    assert top[0].tolist() == gi[:10].tolist()


def test_window_limit_is_an_error():
    with pytest.raises(RuntimeError):
        S.hybrid_fuse(S.RRF, [1], [1.0], [2], [1.0], 5000)
