"""Generates tests/golden/ref_*.json by RUNNING the reference's own code where it compiles from its own sources
(oracle/_ref, built in place from /root/reference by `make -C oracle ref`; see oracle/ref_wrap.c):

  ref_vector_norm.json   src/vector_normalization.h  VectorNorm_L2 / _IP / _Cosine over a grid of values
  ref_minmax_heap.json   src/util/minmax_heap.c driven with the hybrid iterator's comparator
                         (hybrid_reader.c:34-44) and admission rule (:321): K-bounded top-K traces, with ties

Run in the container that has /root/reference:   python tests/golden/make_golden.py
The fixtures let tests/test_oracle_vs_ref.py run where neither /root/reference nor oracle/_ref exists."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from tests import hybrid_replay as H  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def vector_norm_cases():
    rng = np.random.default_rng(37)
    vals = np.concatenate([[0.0, 1.0, 2.0, -1.0, 0.5, 1e-300, 1e300], rng.uniform(-3, 3, 300), rng.standard_exponential(300) * 100])
    return [float(v) for v in vals]


def heap_traces():
    rng = np.random.default_rng(88)
    traces = []
    for k, n, levels in ((1, 40, 5), (3, 60, 4), (10, 200, 12), (10, 200, 100000), (64, 500, 30)):
        scores = (rng.integers(0, levels, n) / 4.0).tolist()     # few levels => many ties
        traces.append({"k": k, "scores": scores})
    return traces


def run_heap_trace(k, scores, force_python=False):
    h = H.make_heap(k, force_python)
    for doc_id, s in enumerate(scores, start=1):
        if h.count < k:
            h.insert(doc_id, s)
        elif s < h.peek_max_score():
            h.exchange_max(doc_id, s)
    return [list(h.pop_min()) for _ in range(h.count)]


def main():
    ref = O.ref_lib("libref_vecnorm")
    assert ref is not None and H.reference_heap_available(), "needs oracle/_ref (make -C oracle ref, /root/reference)"
    ref.ref_vector_norm.restype, ref.ref_vector_norm.argtypes = C.c_double, [C.c_int, C.c_double]
    vals = vector_norm_cases()
    out = {"source": "src/vector_normalization.h via oracle/ref_wrap.c", "values": vals,
           "L2": [ref.ref_vector_norm(0, v) for v in vals], "IP": [ref.ref_vector_norm(1, v) for v in vals],
           "COSINE": [ref.ref_vector_norm(2, v) for v in vals]}
    json.dump(out, open(os.path.join(HERE, "ref_vector_norm.json"), "w"))
    traces = heap_traces()
    for t in traces:
        t["yielded"] = run_heap_trace(t["k"], t["scores"])
    json.dump({"source": "src/util/minmax_heap.c + cmpVecSimResByScore (hybrid_reader.c:34-44), admission :321",
               "traces": traces}, open(os.path.join(HERE, "ref_minmax_heap.json"), "w"))
    print("wrote", len(vals), "values x 3 metrics and", len(traces), "heap traces")


if __name__ == "__main__":
    main()
