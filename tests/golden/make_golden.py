"""Generates tests/golden/ref_*.json by RUNNING the reference's own code where it compiles from its own sources
(oracle/_ref, built in place from /root/reference by `make -C oracle ref`; see oracle/ref_wrap.c):

  ref_vector_norm.json   src/vector_normalization.h  VectorNorm_L2 / _IP / _Cosine over a grid of values
  ref_minmax_heap.json   src/util/minmax_heap.c driven with the hybrid iterator's comparator
                         (hybrid_reader.c:34-44) and admission rule (:321): K-bounded top-K traces, with ties
  ref_scorer_trees.json  src/ext/default.c + src/index_result/index_result.c (compiled in place, driven through oracle/ext_harness.c):
                         seven scorers and IndexResult_MinOffsetDelta over 240 random result trees -- flat and nested
                         intersections / unions of terms with positions, virtual / numeric / metric leaves, and the shapes
                         round 4's general tile kernel builds (Intersection{Union{..}, term, ...}; a NOT child = a virtual
                         leaf of frequency 0)

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


SCORERS = ("TFIDF", "TFIDF.DOCNORM", "BM25", "BM25STD", "BM25STD.TANH", "DISMAX", "DOCSCORE")


def scorer_tree_cases():
    """[(spec, kwargs)]: result-tree specs as oracle/ext.py Tree takes them (nested lists after the JSON round trip)"""
    rng = np.random.default_rng(2024)

    def term(name):
        pos = sorted(set(int(x) for x in rng.integers(1, 60, int(rng.integers(0, 5)))))
        return ["term", float(rng.choice([1.0, 0.5, 2.0])), int(rng.choice([1, 1, 2, 3, 7, 40])), float(rng.uniform(0.1, 9)),
                float(rng.uniform(0.05, 6)), name, pos]

    def leaf():
        k = str(rng.choice(["term", "term", "term", "virtual", "numeric", "metric"]))
        return term("t") if k == "term" else [k, float(rng.choice([1.0, 0.5, 0.0, 3.0])), int(rng.choice([0, 1, 2]))]

    def tree(depth):
        if depth == 0 or rng.random() < 0.3:
            return leaf()
        return [str(rng.choice(["union", "intersection", "intersection"])), float(rng.choice([1.0, 0.3, 3.0])),
                [tree(depth - 1) for _ in range(int(rng.integers(1, 5)))]]

    cases = []
    for i in range(240):
        if i % 4 == 0:      # the general tile kernel's shapes: a root intersection of terms / unions of terms, at times with a NOT child
            kids = []
            for _ in range(int(rng.integers(1, 4))):
                if rng.random() < 0.4:
                    kids.append(["union", float(rng.choice([1.0, 0.5, 2.0])), [term("u") for _ in range(int(rng.integers(1, 4)))]])
                else:
                    kids.append(term("a"))
            if rng.random() < 0.5:
                kids.append(["virtual", float(rng.choice([1.0, 0.3, 4.0])), 0])
            spec = ["intersection", float(rng.choice([1.0, 1.5])), kids]
        else:
            spec = tree(int(rng.integers(1, 4)))
            if spec[0] not in ("union", "intersection"):
                spec = ["intersection", 1.0, [spec]]
        kw = dict(doc_score=float(rng.choice([1.0, 0.25, 0.5])), max_freq=int(rng.choice([1, 3, 50])), doc_len=int(rng.choice([1, 9, 1234])),
                  num_docs=int(rng.integers(1, 10 ** 5)), avg_doc_len=float(rng.uniform(1, 300)), min_score=float(rng.choice([0.0, 0.0, 2.0])),
                  tanh_factor=int(rng.choice([1, 4, 20])))
        cases.append([spec, kw])
    return cases


def spec_tuple(s):
    """JSON lists -> the tuples oracle/ext.py Tree and the tests' _to_oracle take"""
    if s[0] == "term":
        return ("term", s[1], s[2], s[3], s[4], s[5], list(s[6]) if s[6] is not None else None)
    if s[0] in ("virtual", "numeric", "metric"):
        return (s[0], s[1], s[2])
    return (s[0], s[1], [spec_tuple(c) for c in s[2]])


def scorer_trees():
    from oracle import ext as X
    assert X.have_ref(), "needs oracle/_ref/libref_default_ext.so (make -C oracle ref, /root/reference)"
    h = X.Host()
    assert h.load_ref() == X.OK
    out = []
    for spec, kw in scorer_tree_cases():
        t = X.Tree(spec_tuple(spec))
        scores = {sc: h.score(sc, t, **kw) for sc in SCORERS}
        assert all(v == v for v in scores.values()), spec      # no NaN in a fixture
        out.append({"spec": spec, "args": kw, "slop": int(h.ref_slop(t)), "scores": scores})
    return out


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
    trees = scorer_trees()
    json.dump({"source": "src/ext/default.c + src/index_result/index_result.c compiled in place (oracle/_ref/libref_default_ext.so), "
                         "driven through oracle/ext_harness.c", "scorers": list(SCORERS), "cases": trees},
              open(os.path.join(HERE, "ref_scorer_trees.json"), "w"))
    print("wrote", len(vals), "values x 3 metrics,", len(traces), "heap traces and", len(trees), "scored result trees")


if __name__ == "__main__":
    main()
