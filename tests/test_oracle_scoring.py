"""Pins oracle/scoring_oracle.c against the reference's scorer / IDF known-answer tests. CPU only."""
import math

import numpy as np
import pytest

import oracle as O


def test_idf_kats():
    # reference src/redisearch_rs/idf/tests/tests.rs:23-80
    idf = O.lib.oracle_idf
    assert idf(100, 10) == 3.0 and idf(100, 0) == idf(100, 1)
    assert idf(0, 1) == 1.0 and idf(0, 0) == 1.0 and idf(1, 1) == 1.0
    assert idf(1000, 1) == 9.0 and idf(1000, 500) == 1.0 and idf(1000, 1000) == 1.0
    prev = idf(10000, 1)
    for t in (2, 10, 100, 1000, 5000, 10000):
        assert idf(10000, t) <= prev
        prev = idf(10000, t)


def test_idf_bm25_kats():
    f = O.lib.oracle_idf_bm25
    assert f(1000, 1) > f(1000, 500)
    assert abs(f(5, 10) - f(10, 10)) < 1e-15          # total clamped to term (idf/src/lib.rs:104)
    assert f(100, 10) == pytest.approx(math.log(1 + 90.5 / 10.5), rel=1e-15)


def hybrid_tree(term_node):
    metric = O.Node(O.R_METRIC, 1.0, 0)  # NewMetricResult: freq 0, weight 1 (index_result/src/core/mod.rs:127-136)
    return O.Node(O.R_HYBRID, 1.0, 0, children=[metric, O.union([term_node])])


def test_hybrid_non_vector_score_kats():
    # reference tests/pytests/test_vecsim.py:1248-1341: N=100 docs, "other" in 10, "text" in 100
    # (overwritten docs are not GC'd from the posting count), docLen 1 / 2, avgDocLen 1.9
    N, avg = 100, 1.9
    other = O.term(1, idf=O.lib.oracle_idf(N, 10), bm25_idf=O.lib.oracle_idf_bm25(N, 10))
    text = O.term(1, idf=O.lib.oracle_idf(N, 100), bm25_idf=O.lib.oracle_idf_bm25(N, 100))
    kw = dict(num_docs=N, avg_doc_len=avg)
    # doc 100 ("other", docLen 1) and docs 91.. ("text value", docLen 2)
    assert O.score("TFIDF", hybrid_tree(other), max_freq=1, doc_len=1, **kw) == 3.0
    assert O.score("TFIDF.DOCNORM", hybrid_tree(text), max_freq=1, doc_len=2, **kw) == 0.5
    assert O.score("BM25", hybrid_tree(other), doc_len=1, **kw) == pytest.approx(1.0948904833203477, rel=1e-7)
    assert O.score("BM25", hybrid_tree(text), doc_len=2, **kw) == pytest.approx(0.36496349444011583, rel=1e-7)
    # The pytest literals were produced by an older float32 IDF (log(1.0F + ...)); the snapshot's
    # idf/src/lib.rs:102-108 computes in f64, so they agree to ~1e-5 (the pytest itself allows 0.01).
    assert O.score("BM25STD", hybrid_tree(other), doc_len=1, **kw) == pytest.approx(2.8078501570291188, rel=1e-7)
    assert O.score("BM25STD", hybrid_tree(text), doc_len=2, **kw) == pytest.approx(0.004858144472727694, rel=2e-5)
    assert O.score("DISMAX", hybrid_tree(other)) == 1.0 and O.score("DISMAX", hybrid_tree(text)) == 1.0
    assert O.score("DOCSCORE", hybrid_tree(text), doc_score=1.0) == 1.0


def score_index_docs():
    """testScoreIndex corpus (reference tests/pytests/test_scorers.py:31-69): doc n has
    title 'hello world '*n (weight 10), body 'lorem ipsum '*n, score sqrt((N-n+10)/(N+10))."""
    N = 25
    docs = []
    for n in range(1, N):
        sc = np.float32(math.sqrt(float(N - n + 10) / float(N + 10)))
        hello = list(range(1, 2 * n, 2))   # token positions 1,3,5..
        world = list(range(2, 2 * n + 1, 2))
        docs.append(dict(n=n, score=float(sc), f=10 * n, doc_len=22 * n, max_freq=10 * n, hello=hello, world=world))
    return docs


@pytest.mark.parametrize("scorer,expected", [
    ("TFIDF", [("doc1", 1.97), ("doc2", 1.94), ("doc3", 1.91), ("doc4", 1.88), ("doc5", 1.85)]),
    ("TFIDF.DOCNORM", [("doc1", 0.9), ("doc2", 0.88), ("doc3", 0.87), ("doc4", 0.86), ("doc5", 0.84)]),
    ("BM25", [("doc17", 0.73), ("doc18", 0.73), ("doc16", 0.72), ("doc19", 0.72), ("doc15", 0.72)]),
    ("BM25STD", [("doc1", 0.08), ("doc2", 0.08), ("doc3", 0.08), ("doc4", 0.08), ("doc5", 0.08)]),
    ("BM25STD.TANH", [("doc1", 0.02), ("doc2", 0.02), ("doc3", 0.02), ("doc4", 0.02), ("doc5", 0.02)]),
    ("DISMAX", [("doc24", 480.0), ("doc23", 460.0), ("doc22", 440.0), ("doc21", 420.0), ("doc20", 400.0)]),
    ("DOCSCORE", [("doc1", 0.99), ("doc2", 0.97), ("doc3", 0.96), ("doc4", 0.94), ("doc5", 0.93)]),
])
def test_score_index_top5(scorer, expected):
    docs = score_index_docs()
    nd = len(docs)
    avg = sum(d["doc_len"] for d in docs) / nd
    idf, bidf = O.lib.oracle_idf(nd, nd), O.lib.oracle_idf_bm25(nd, nd)
    scored = []
    for d in docs:
        tree = O.intersection([O.term(d["f"], idf, bidf, offsets=d["hello"]),
                               O.term(d["f"], idf, bidf, offsets=d["world"])])
        s = O.score(scorer, tree, doc_score=d["score"], max_freq=d["max_freq"], doc_len=d["doc_len"],
                    num_docs=nd, avg_doc_len=avg)
        scored.append((s, d["n"]))
    # sorter tie-break: equal score => lower doc id first (reference src/result_processor.c:849);
    # doc ids follow insertion order here
    scored.sort(key=lambda t: (-t[0], t[1]))
    got = [("doc%d" % n, round(s, 2)) for s, n in scored[:5]]
    assert got == expected


def test_bm25std_explain_kats():
    # reference tests/pytests/test_scorers.py:198-242: IDF 0.13 (3 docs, term in 3), F 10,
    # docLen 23/35/45, avg 34.33 => 0.54 / 0.52 / 0.51 ; weighted variant => 0.12
    bidf = O.lib.oracle_idf_bm25(3, 3)
    assert round(bidf, 2) == 0.13
    avg = (23 + 35 + 45) / 3
    for doc_len, exp in ((23, 0.54), (35, 0.52), (45, 0.51)):
        tree = O.intersection([O.term(10, 0, bidf), O.term(10, 0, bidf)])
        assert round(O.score("BM25STD", tree, doc_len=doc_len, num_docs=3, avg_doc_len=avg), 2) == exp
    inner = O.union([O.term(10, 0, bidf, weight=0.5), O.term(10, 0, bidf)], weight=0.3)
    assert round(O.score("BM25STD", inner, doc_len=23, num_docs=3, avg_doc_len=avg), 2) == 0.12


def test_slop_kats():
    # reference tests/pytests/test_scorers.py:98-109,170-181 (slop 1/2/3) and
    # src/index_result/index_result.c:51-103
    mk = lambda a, b: O.intersection([O.term(1, 1, offsets=a), O.term(1, 1, offsets=b)])
    assert O.lib.oracle_slop(mk([1], [2]).ptr) == 1
    assert O.lib.oracle_slop(mk([1], [3]).ptr) == 2
    assert O.lib.oracle_slop(mk([1], [4]).ptr) == 3
    assert O.lib.oracle_slop(mk([2, 4, 8], [0, 5, 12]).ptr) == 1      # doc comment example
    assert O.lib.oracle_slop(O.term(1, 1).ptr) == 1                   # non-aggregate
    assert O.lib.oracle_slop(O.intersection([O.term(1, 1), O.term(1, 1), O.term(1, 1)]).ptr) == 2  # no offsets: num-1


def test_legacy_bm25_explain_kat():
    # reference tests/pytests/test_scorers.py:159-196: 0.35 = 1 * IDF 1.00 * F 10 / (F 10 + 1.2*(1-.5+.5*30.00))
    t = O.term(10, idf=1.0)
    assert round(O.score("BM25", t, avg_doc_len=30.0), 2) == 0.35


def test_filterout_and_minscore_rules():
    # default.c:112-128: score 0 doc -> 0 ; tfidf below minScore -> 0
    t = O.intersection([O.term(2, 1.0)])
    assert O.score("TFIDF", t, doc_score=0.0) == 0.0
    assert O.score("TFIDF", t, doc_score=1.0, max_freq=2, min_score=5.0) == 0.0
    assert O.score("TFIDF", t, doc_score=1.0, max_freq=0) == 0.0


def test_hamming():
    a = np.frombuffer(b"\xff\x00\x0f", dtype=np.uint8)
    b = np.frombuffer(b"\x0f\x00\x0f", dtype=np.uint8)
    assert O.lib.oracle_hamming(O._p(a), 3, O._p(b), 3) == 1.0 / 5
    assert O.lib.oracle_hamming(O._p(a), 3, O._p(b), 2) == 0.0


def test_flat_form_matches_tree_form():
    rng = np.random.default_rng(3)
    T, M, N = 2, 200, 5000
    freq = rng.integers(1, 9, (T, M)).astype(np.uint32)
    doc_len = rng.integers(20, 400, M).astype(np.uint32)
    max_freq = np.maximum(freq.max(0), 1).astype(np.uint32)
    doc_score = rng.uniform(0.1, 1, M).astype(np.float32)
    n = [700, 90]
    idf = [O.lib.oracle_idf(N, x) for x in n]
    bidf = [O.lib.oracle_idf_bm25(N, x) for x in n]
    w = [1.0, 0.5]
    for scorer in O.SCORER_IDS:
        out = O.score_flat(scorer, freq, doc_len, max_freq, doc_score, idf, bidf, w, 1.0, N, 180.5)
        for m in range(0, M, 17):
            tree = O.intersection([O.term(int(freq[t, m]), idf[t], bidf[t], weight=w[t]) for t in range(T)])
            ref = O.score(scorer, tree, doc_score=float(doc_score[m]), max_freq=int(max_freq[m]),
                          doc_len=int(doc_len[m]), num_docs=N, avg_doc_len=180.5)
            assert out[m] == ref, scorer


def test_bm25std_norm_explain_kats():
    # reference tests/pytests/test_scorers.py:244-291: 'Final BM25STD.NORM: 1.00 = Original Score: 0.54 / Max Score: 0.54',
    # 0.97 = 0.52 / 0.54, 0.95 = 0.51 / 0.54 ; the weighted query: 1.00 / 0.97 / 0.95 of 0.12 / 0.12 / 0.12
    bidf = O.lib.oracle_idf_bm25(3, 3)
    avg = (23 + 35 + 45) / 3
    raw = [O.score("BM25STD", O.intersection([O.term(10, 0, bidf), O.term(10, 0, bidf)]), doc_len=dl, num_docs=3,
                   avg_doc_len=avg) for dl in (23, 35, 45)]
    assert [round(x, 2) for x in raw] == [0.54, 0.52, 0.51]
    assert [round(x, 2) for x in O.max_normalize(raw)] == [1.00, 0.97, 0.95]
    raw = [O.score("BM25STD", O.union([O.term(10, 0, bidf, weight=0.5), O.term(10, 0, bidf)], weight=0.3), doc_len=dl,
                   num_docs=3, avg_doc_len=avg) for dl in (23, 35, 45)]
    assert [round(x, 2) for x in raw] == [0.12, 0.12, 0.12]
    assert [round(x, 2) for x in O.max_normalize(raw)] == [1.00, 0.97, 0.95]
    # maxValue starts at 0: all-zero and all-negative score lists pass through unchanged (result_processor.c:1784)
    assert O.max_normalize([0.0, 0.0]).tolist() == [0.0, 0.0]
    assert O.max_normalize([-1.0, -2.0]).tolist() == [-1.0, -2.0]
    assert O.max_normalize([]).tolist() == []
