"""Queries beyond the device path's limits (32 terms / 64 nodes / 16 levels per query tree, DESIGN "Known limits") must
FAIL LOUDLY at the ABI -- NULL + RSGPU_LastError, never a truncated answer -- and are then the host's: the reference's own
iterators (restated by the oracle) walk the postings and the scorer PLUGIN (librsgpu_scorers.so, any number of children:
it IS the reference's per-result interface, reference src/ext/default.c:253-302) scores every result.  Here: a 33-term
intersection and a 33-term union refused by RSGPU_Intersect / RSGPU_Union / RSGPU_EvalTreeNodes, a 65-node tree refused,
a 20-level tree evaluated but refused by the device scorer; the 33-term intersection answered on the host path and, with one term fewer, identically by the device."""
import numpy as np
import pytest

import oracle as O
from oracle import ext as X
from redisearch_amd import search as S
from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu


def _lists(n_lists, n_docs=3000, seed=5):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n_lists):
        docs = np.unique(np.concatenate([rng.integers(1, n_docs, n_docs // 2), np.arange(7, n_docs, 97)])).astype(np.uint64)
        ii = O.InvertedIndex(O.C_FREQS_ONLY)
        ii.add_many(docs, rng.integers(1, 9, docs.size).astype(np.uint32))
        out.append(ii)
    return out


def test_over_limit_trees_are_refused_with_a_reason():
    o = _lists(33)
    g = [S.Postings.from_flat(l.flatten()) for l in o]
    try:
        for build, what in ((lambda: S.intersect(g), "32"), (lambda: S.union(g), "32"),
                            (lambda: S.NodeHits(("and", 1.0, [("t", i) for i in range(33)]), g), "32")):
            with pytest.raises(RuntimeError) as e:
                build()
            assert what in str(e.value), str(e.value)
        # 65 nodes over 32 terms: a chain of single-child aggregates on top of a 32-term intersection
        tree = ("and", 1.0, [("t", i) for i in range(32)])
        for _ in range(32):
            tree = ("or", 1.0, [tree])
        with pytest.raises(RuntimeError, match="64 nodes"):
            S.NodeHits(tree, g[:32])
        # 20 nested aggregates (the limit: 16 levels below the root)
        tree = ("t", 0)
        for lvl in range(20):
            tree = ("and" if lvl % 2 else "or", 1.0, [tree, ("t", lvl + 1)])
        deep = S.NodeHits(tree, g[:21])            # the boolean evaluation has no depth limit; the device SCORER has
        assert len(deep.read()[0]) > 0
        doc_len = np.full(3001, 100, np.uint32)
        with pytest.raises(RuntimeError, match="16 levels"):
            deep.score(S.DocTable(doc_len, np.ones(3001, np.float32)), "BM25STD", [1.0] * 21, [1.0] * 21, [1.0] * 21, 3000, 100.0)
        deep.free()
    finally:
        for x in g:
            x.free()


def test_a_33_term_intersection_is_answered_by_the_host_plugin_path():
    from redisearch_amd import build as B
    B.build_c()
    o = _lists(33)
    n_docs, avg = 3000, 120.0
    rng = np.random.default_rng(9)
    doc_len = rng.integers(20, 300, n_docs + 1).astype(np.uint32)
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in o]
    bidf = [S.calculate_idf_bm25(n_docs, l.unique_docs) for l in o]
    # the host path: the reference's intersection (oracle restatement) -> one result tree per hit -> the plugin's scorer
    oi, of, _ = O.intersect(o)
    assert len(oi) > 5
    host = X.Host()
    assert host.load_plugin() == X.OK
    host_scores = []
    for h in range(len(oi)):   # (of: [list][hit], the caller's list order)
        kids = [("term", 1.0, int(of[s][h]), idf[s], bidf[s], "t%d" % s, None) for s in range(33)]
        tree = X.Tree(("intersection", 1.0, kids))
        host_scores.append(host.score("BM25STD", tree, doc_score=1.0, doc_len=int(doc_len[int(oi[h])]), num_docs=n_docs, avg_doc_len=avg, slop=1))
    want = O.score_flat("BM25STD", of, doc_len[oi.astype(np.int64)], np.ones(len(oi)), np.ones(len(oi), np.float32),
                        idf, bidf, [1.0] * 33, 1.0, n_docs, avg)
    assert np.allclose(host_scores, want, rtol=1e-12, atol=0)
    # ... and the same query with one term fewer runs on the device and agrees with the same host composition
    g = [S.Postings.from_flat(l.flatten()) for l in o[:32]]
    try:
        hits = S.intersect(g)
        gi, gf = hits.read()
        oi32, of32, _ = O.intersect(o[:32])
        assert gi.tolist() == oi32.tolist() and gf.tolist() == of32.tolist()
        gs = hits.score(S.DocTable(doc_len, np.ones(n_docs + 1, np.float32)), "BM25STD", idf[:32], bidf[:32], [1.0] * 32, n_docs, avg)
        host32 = []
        for h in range(len(oi32)):
            kids = [("term", 1.0, int(of32[s][h]), idf[s], bidf[s], "t", None) for s in range(32)]
            host32.append(host.score("BM25STD", X.Tree(("intersection", 1.0, kids)), doc_score=1.0, doc_len=int(doc_len[int(oi32[h])]),
                                     num_docs=n_docs, avg_doc_len=avg, slop=1))
        assert np.allclose(gs, host32, rtol=1e-12, atol=0)
    finally:
        for x in g:
            x.free()
