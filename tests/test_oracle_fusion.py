"""CPU tests pinning the FT.HYBRID fusion oracle (oracle/scoring_oracle.c oracle_hybrid_fuse) to the reference's
own known answers: tests/cpptests/test_cpp_hybridmerger.cpp (RRF ranks / LINEAR weights), the closed forms of
src/vector_normalization.h:37-60, and the merger's window rule (src/result_processor.c:2549-2571)."""
import numpy as np

import oracle as O


def as_dict(ids, sc):
    return dict(zip(ids.tolist(), sc.tolist()))


def test_rrf_full_intersection_kat():
    # test_cpp_hybridmerger.cpp:610-662: constant 60, window 10
    d = as_dict(*O.hybrid_fuse(O.RRF, [1, 2, 3], [0.9, 0.5, 0.1], [3, 1, 2], [0.8, 0.4, 0.2], 10))
    assert abs(d[1] - (1 / 61 + 1 / 62)) < 1e-15 and abs(d[2] - (1 / 62 + 1 / 63)) < 1e-15 and abs(d[3] - (1 / 63 + 1 / 61)) < 1e-15


def test_rrf_partial_intersection_kat_and_order():
    # test_cpp_hybridmerger.cpp:1120-1170: {1,2,3} and {2,3,4,5}, window 5
    ids, sc = O.hybrid_fuse(O.RRF, [1, 2, 3], [0.9, 0.7, 0.5], [2, 3, 4, 5], [0.8, 0.6, 0.4, 0.2], 5)
    assert ids.tolist() == [2, 3, 1, 4, 5]
    assert np.allclose(sc, [1 / 62 + 1 / 61, 1 / 63 + 1 / 62, 1 / 61, 1 / 63, 1 / 64], rtol=0, atol=1e-15)


def test_rrf_disjoint_ranks_and_tie_break():
    # test_cpp_hybridmerger.cpp:560-585: disjoint lists -> rank r in either list scores 1/(60+r); equal scores:
    # lower doc id first (cmpByScore, result_processor.c:849)
    ids, sc = O.hybrid_fuse(O.RRF, [11, 12], [5.0, 4.0], [21, 22], [0.1, 0.2], 10)
    assert ids.tolist() == [11, 21, 12, 22] and sc.tolist() == [1 / 61, 1 / 61, 1 / 62, 1 / 62]


def test_linear_kats():
    # :265-291: 0.3*2.0 + 0.7*4.0 = 3.4 ; :330-350: disjoint 0.4*1.0 and 0.6*3.0
    assert O.hybrid_fuse(O.LINEAR, [7], [2.0], [7], [4.0], 10, weights=(0.3, 0.7))[1].tolist() == [0.3 * 2.0 + 0.7 * 4.0]
    d = as_dict(*O.hybrid_fuse(O.LINEAR, [1], [1.0], [2], [3.0], 10, weights=(0.4, 0.6)))
    assert abs(d[1] - 0.4) < 1e-15 and abs(d[2] - 1.8) < 1e-15


def test_window_cuts_each_upstream_and_ranks_follow_consumption():
    # only the first `window` of each list are consumed (result_processor.c:2556); a doc beyond the window of one
    # list only keeps the other list's contribution
    ids, sc = O.hybrid_fuse(O.RRF, [1, 2, 3, 4], [4, 3, 2, 1], [4, 3, 2, 1], [.1, .2, .3, .4], 2)
    d = as_dict(ids, sc)
    assert set(d) == {1, 2, 3, 4} and d[1] == 1 / 61 and d[4] == 1 / 61 and d[2] == 1 / 62 and d[3] == 1 / 62


def test_vector_normalisation_closed_forms():
    # vector_normalization.h:37-60
    assert O.vector_norm(O.L2, 0.0) == 1.0 and O.vector_norm(O.L2, 3.0) == 0.25
    assert O.vector_norm(O.IP, 1.0) == 1.0 and O.vector_norm(O.IP, -1.0) == 0.0 and O.vector_norm(O.IP, 0.0) == 0.5
    assert O.vector_norm(O.COSINE, 0.0) == 1.0 and O.vector_norm(O.COSINE, 1.0) == 0.5 and O.vector_norm(O.COSINE, 2.0) == 0.0
    ids, sc = O.hybrid_fuse(O.LINEAR, [5], [2.0], [5, 6], [3.0, 1.0], 10, weights=(1.0, 2.0), metric=O.L2)
    assert ids.tolist() == [5, 6] and sc.tolist() == [2.0 + 2.0 * 0.25, 2.0 * 0.5]
