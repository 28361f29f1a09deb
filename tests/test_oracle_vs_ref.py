"""The restatement against the reference's OWN code, for the pieces of the path that compile from their own sources
(oracle/ref_wrap.c, `make -C oracle ref` -> oracle/_ref/): src/vector_normalization.h and src/util/minmax_heap.c.
Skipped where neither the prebuilt libraries nor /root/reference exist."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from tests import hybrid_replay as H

needs_vecnorm = pytest.mark.skipif(O.ref_lib("libref_vecnorm") is None, reason="oracle/_ref not available")
needs_heap = pytest.mark.skipif(not H.reference_heap_available(), reason="oracle/_ref not available")


@needs_vecnorm
def test_vector_norm_equals_the_reference_functions():
    ref = O.ref_lib("libref_vecnorm")
    ref.ref_vector_norm.restype, ref.ref_vector_norm.argtypes = C.c_double, [C.c_int, C.c_double]
    rng = np.random.default_rng(37)
    values = np.concatenate([[0.0, 1.0, 2.0, -1.0, 0.5, 1e-300, 1e300, np.inf], rng.uniform(-3, 3, 2000),
                             rng.standard_exponential(2000) * 100])
    for metric in (0, 1, 2):                               # VecSimMetric_L2 / _IP / _Cosine
        for v in values.tolist():
            a, b = O.vector_norm(metric, v), ref.ref_vector_norm(metric, v)
            assert a == b or (np.isnan(a) and np.isnan(b)), (metric, v, a, b)


@needs_heap
def test_reference_heap_and_stand_in_agree_without_ties():
    rng = np.random.default_rng(88)
    for k in (1, 3, 10, 64):
        ref, py = H.make_heap(k), H.make_heap(k, force_python=True)
        assert isinstance(ref, H.RefHeap) and isinstance(py, H.PyHeap)
        scores = rng.permutation(5000)[:600].astype(np.float64) / 7.0      # distinct
        for doc_id, s in enumerate(scores.tolist(), start=1):
            for h in (ref, py):
                if h.count < k:
                    h.insert(doc_id, s)
                elif s < h.peek_max_score():
                    h.exchange_max(doc_id, s)
            assert ref.peek_max_score() == py.peek_max_score()
        out_ref = [ref.pop_min() for _ in range(ref.count)]
        out_py = [py.pop_min() for _ in range(py.count)]
        assert out_ref == out_py and [s for _, s in out_ref] == sorted(s for _, s in out_ref)
        assert sorted(s for _, s in out_ref) == sorted(scores.tolist())[:k]


@needs_heap
def test_reference_heap_tie_rule():
    # cmpVecSimResByScore (hybrid_reader.c:34-44): equal scores compare by doc id with "smaller id = greater"; what the
    # heap then does with ties is the reference implementation's behaviour -- pinned here so a change would be noticed
    h = H.make_heap(3)
    for doc_id in (5, 1, 9, 3, 7):
        if h.count < 3:
            h.insert(doc_id, 1.0)
        elif 1.0 < h.peek_max_score():                     # strict admission: equal scores never replace (:321)
            h.exchange_max(doc_id, 1.0)
    got = [h.pop_min() for _ in range(3)]
    assert sorted(d for d, _ in got) == [1, 5, 9] and all(s == 1.0 for _, s in got)
