"""GPU: the general form of the hybrid tile kernel (hybrid_kernels.hip hybrid_tree_tile_kernel; RSGPU_HybridQueryPath == 2) --
RSGPU_HybridQuery with the hit list wanted / five to eight lists / slop-dependent scorers over lists with offsets, and
RSGPU_HybridTreeQuery over a root intersection of terms, unions of terms and intersections of terms with max_slop / in_order --
against (a) the staged form behind the same entry points (knob hybrid_tree_tiles = 0: RSGPU_EvalTree / the ten-kernel pipeline,
which tests/test_gpu_tree.py and test_gpu_hybrid_query.py pin to the CPU oracle) BIT FOR BIT: hit count, top-N ids and scores,
KNN ids and distances, the hit list's doc ids, per-term frequencies and term records; and (b) the CPU oracle directly: set algebra
over the decoded lists, the oracle's proximity restatement, its result-tree scorers in the reference's order
(result_processor.c:849: score descending, doc id ascending), O.FlatIndex distances."""
import zlib

import numpy as np
import pytest

import oracle as O
from redisearch_amd import search as S
from redisearch_amd import vecsim as V
from tests.test_gpu_tree import OracleTree, rand_list

pytestmark = pytest.mark.gpu
T, U, I = S.OP_TERM, S.OP_UNION, S.OP_INTERSECT
SCORERS = ["BM25STD", "BM25STD.TANH", "BM25", "TFIDF", "TFIDF.DOCNORM", "DOCSCORE", "DISMAX"]


def knob(name, value):
    V.load().RSGPU_SetTuning(name.encode(), int(value))


def general_and_staged(make, want_path=2):
    """make() -> a HybridQuery / HybridTreeQuery; -> (general results, staged results, general hits, staged hits)"""
    hq = make()
    try:
        knob("hybrid_tree_tiles", 1)
        hq.run()
        path = S.hybrid_path()
        a = hq.results()
        ha = hq.take_hits() if hq._hits_ptr is not None else None
        hq.run()                                  # again: flags re-armed, scratch reused
        a2 = hq.results()
        knob("hybrid_tree_tiles", 0)
        hq.run()
        assert S.hybrid_path() == 0
        b = hq.results()
        hb = hq.take_hits() if hq._hits_ptr is not None else None
    finally:
        knob("hybrid_tree_tiles", 1)
    assert path == want_path, "the general tile kernel did not take the query (path %d)" % path
    for x in (a, a2):
        assert x["n_hits"] == b["n_hits"], ("hit count", x["n_hits"], b["n_hits"])
        assert x["top"][0].tolist() == b["top"][0].tolist(), ("top-N ids", x["top"][0][:8], b["top"][0][:8])
        assert x["top"][1].tolist() == b["top"][1].tolist(), ("top-N scores", x["top"][1][:8], b["top"][1][:8])
        assert x["knn"][0].tolist() == b["knn"][0].tolist(), ("KNN ids", x["knn"][0][:8], b["knn"][0][:8])
        assert x["knn"][1].tolist() == b["knn"][1].tolist(), ("KNN distances", x["knn"][1][:8], b["knn"][1][:8])
    return a, b, ha, hb


def same_hit_lists(ha, hb, n_lists, with_records=False):
    ia, fa = ha.read()
    ib, fb = hb.read()
    assert len(ia) == len(ib)
    bad = np.flatnonzero(ia != ib)
    assert bad.size == 0, ("hit ids differ first at", bad[:4], ia[bad[:4]], ib[bad[:4]])
    assert np.array_equal(fa, fb), ("frequencies", np.argwhere(fa != fb)[:4])
    assert ha.leaf_order() == hb.leaf_order()
    if with_records and len(ia):
        for l in range(n_lists):
            ra, rb = ha.read_records(l), hb.read_records(l)
            for key in ("entry", "freq", "off_pos", "off_len"):
                assert np.array_equal(ra[key], rb[key]), (l, key)
            assert ra["mask"] == rb["mask"], (l, "mask")
    return ia, fa


def flat_corpus(n_docs, dfs, seed, codec=O.C_FREQS_ONLY, first=1):
    rng = np.random.default_rng(seed)
    out = []
    for df in dfs:
        docs = np.flatnonzero(rng.random(n_docs) < df).astype(np.uint64) + first
        ii = O.InvertedIndex(codec)
        ii.add_many(docs, np.minimum(1 + rng.geometric(0.5, docs.size), 255).astype(np.uint32))
        out.append(ii)
    return out, rng


def table_for(rng, n_docs):
    arrays = ((50 + rng.poisson(150, n_docs + 1)).astype(np.uint32), rng.choice([1.0, 0.5, 0.25], n_docs + 1).astype(np.float32),
              rng.integers(1, 50, n_docs + 1).astype(np.uint32))
    t = S.DocTable(*arrays)
    t._arrays = arrays                          # (doc_len, doc_score, max_freq) for the oracle side
    return t


# ---- RSGPU_HybridQuery (flat AND): what the two-launch form left to the staged pipeline ---------------------------------------
@pytest.mark.parametrize("n_lists", [1, 2, 4])
@pytest.mark.parametrize("scorer", ["BM25STD", "TFIDF", "DISMAX"])
def test_flat_query_with_the_hit_list_wanted(n_lists, scorer):
    n_docs = 600_000
    lists_o, rng = flat_corpus(n_docs, (0.5, 0.45, 0.6, 0.4)[:n_lists], 200 + n_lists)
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    table = table_for(rng, n_docs)
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 48, V.VecSimMetric_L2)
    idx.add_philox_rows(7, 0, 200_000, 1)
    q = O.philox_rows(7, 1 << 40, 1, 48)[0]
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists_o]
    bidf = [S.calculate_idf_bm25(n_docs, l.unique_docs) for l in lists_o]
    w = [1.0, 0.5, 2.0, 1.5][:n_lists]
    a, b, ha, hb = general_and_staged(lambda: S.HybridQuery(g, table, scorer, idf, bidf, w, n_docs, 200.0, top_n=10, index=idx, q=q,
                                                            k=10, root_weight=1.25, want_hits=True))
    want_ids, want_freqs = O.intersect(lists_o)[:2]
    assert a["n_hits"] == len(want_ids) and len(ha) == len(want_ids)
    ids, fr = same_hit_lists(ha, hb, n_lists)
    assert ids.tolist() == np.asarray(want_ids).tolist()
    for li in range(n_lists):                    # (RSGPU_Hits_Read gives the frequencies back in the caller's list order)
        assert fr[li].tolist() == np.asarray(want_freqs[li]).tolist(), ("frequencies of list", li)
    # the hit list the tile kernel wrote is a hit list like any other: scoring it stage by stage gives the staged list's scores
    sa = ha.score(table, scorer, idf, bidf, w, n_docs, 200.0, root_weight=1.25)
    sb = hb.score(table, scorer, idf, bidf, w, n_docs, 200.0, root_weight=1.25)
    assert np.array_equal(sa, sb)
    assert ha.topn(10)[0].tolist() == a["top"][0].tolist() and ha.topn(10)[1].tolist() == a["top"][1].tolist()
    assert ha.knn_rerank(idx, q, 10)[0].tolist() == a["knn"][0].tolist()
    idx.free()


@pytest.mark.parametrize("n_lists", [5, 6, 8])
def test_flat_query_of_five_to_eight_lists(n_lists):
    n_docs = 500_000
    lists_o, rng = flat_corpus(n_docs, (0.7, 0.75, 0.8, 0.72, 0.78, 0.9, 0.85, 0.95)[:n_lists], 300 + n_lists)
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    table = table_for(rng, n_docs)
    idx = V.VecSimIndex(V.VecSimType_FLOAT16, 64, V.VecSimMetric_Cosine)
    idx.add_philox_rows(9, 0, 300_000, 1)
    q = O.philox_rows(9, 1 << 40, 1, 64, O.F16)[0]
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists_o]
    bidf = [S.calculate_idf_bm25(n_docs, l.unique_docs) for l in lists_o]
    w = [1.0, 0.5, 2.0, 1.5, 0.25, 3.0, 1.0, 0.75][:n_lists]
    for scorer in ("BM25STD", "BM25", "DISMAX"):
        a, b, _, _ = general_and_staged(lambda: S.HybridQuery(g, table, scorer, idf, bidf, w, n_docs, 200.0, top_n=12, index=idx, q=q, k=7))
        assert a["n_hits"] == len(O.intersect(lists_o)[0])
        assert len(a["top"][0]) == 12 and len(a["knn"][0]) == 7
    idx.free()


@pytest.mark.parametrize("scorer", ["BM25", "TFIDF", "TFIDF.DOCNORM"])
def test_slop_dependent_scorers_over_lists_with_offsets(scorer):
    """Full-codec lists (FT.CREATE's default): TFIDF / legacy BM25 divide by IndexResult_MinOffsetDelta of the hit's term offsets
    (index_result.c:51-103) -- the tile kernel computes it per hit where round 3's two-launch form handed the query back."""
    rng = np.random.default_rng(zlib.crc32(scorer.encode()) % 1000)
    built = [rand_list(rng, O.C_FULL, n, 9000, True) for n in (3000, 4200, 5000)]
    lists_o, recs = [x[0] for x in built], [x[1] for x in built]
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    n_docs = 9000
    table = table_for(rng, n_docs)
    sizes = [l.unique_docs for l in lists_o]
    idf = [S.calculate_idf(n_docs, s) for s in sizes]
    bidf = [S.calculate_idf_bm25(n_docs, s) for s in sizes]
    w = [1.0, 0.5, 2.0]
    a, b, ha, hb = general_and_staged(lambda: S.HybridQuery(g, table, scorer, idf, bidf, w, n_docs, 180.0, top_n=15, root_weight=1.5,
                                                            want_hits=True))
    same_hit_lists(ha, hb, 3, with_records=True)
    a2, _, _, _ = general_and_staged(lambda: S.HybridQuery(g, table, scorer, idf, bidf, w, n_docs, 180.0, top_n=15, root_weight=1.5))
    assert a2["top"][0].tolist() == a["top"][0].tolist() and a2["top"][1].tolist() == a["top"][1].tolist()
    # the oracle directly: every hit's result tree scored (term offsets -> the slop the scorer divides by), the reference's order
    ot = OracleTree(I, [(T, 1.0, [0]), (T, 1.0, [1]), (T, 1.0, [2])], recs, sizes)
    assert a["n_hits"] == len(ot.docs) and ha.read()[0].tolist() == ot.docs
    dl, ds, mf = table._arrays
    scored = []
    for d in ot.docs:
        node = ot.node(d, idf, bidf, w)
        node.c.weight = 1.5
        scored.append((O.score(scorer, node, float(ds[d]), int(mf[d]), int(dl[d]), n_docs, 180.0), d))
    scored.sort(key=lambda t: (-t[0], t[1]))
    assert a["top"][0].tolist() == [d for _, d in scored[:15]]
    assert a["top"][1].tolist() == [x for x, _ in scored[:15]]


def test_a_window_that_overflows_lds_and_sixty_four_bit_doc_ids():
    """one very long list against short ones (the probed window does not fit the pool: the in-memory search), doc ids above 2^32"""
    first = (1 << 33) + 5
    n_docs = 3_000_000
    rng = np.random.default_rng(5)
    docs_small = np.sort(rng.choice(n_docs, 3000, replace=False)).astype(np.uint64) + first
    docs_big = np.flatnonzero(rng.random(n_docs) < 0.9).astype(np.uint64) + first
    docs_mid = np.flatnonzero(rng.random(n_docs) < 0.3).astype(np.uint64) + first
    lists_o = []
    for d in (docs_small, docs_big, docs_mid):
        ii = O.InvertedIndex(O.C_FREQS_ONLY)
        ii.add_many(d, np.minimum(1 + rng.geometric(0.5, d.size), 255).astype(np.uint32))
        lists_o.append(ii)
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    table = S.DocTable((50 + rng.poisson(150, n_docs)).astype(np.uint32), np.ones(n_docs, np.float32), first_doc_id=first)
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists_o]
    bidf = [S.calculate_idf_bm25(n_docs, l.unique_docs) for l in lists_o]
    a, b, ha, hb = general_and_staged(lambda: S.HybridQuery(g, table, "BM25STD", idf, bidf, [1.0, 1.0, 1.0], n_docs, 200.0, top_n=10,
                                                            want_hits=True))
    ids, _ = same_hit_lists(ha, hb, 3)
    assert ids.tolist() == np.asarray(O.intersect(lists_o)[0]).tolist() and ids.min() > (1 << 33)


def test_no_hits_and_fewer_hits_than_asked_for():
    n_docs = 200_000
    rng = np.random.default_rng(9)
    even = (np.arange(2, n_docs, 2)).astype(np.uint64)
    odd = (np.arange(1, n_docs, 2)).astype(np.uint64)
    few = np.asarray([10, 11, 500, 501, 70_000], np.uint64)
    mk = lambda d: (lambda ii: (ii.add_many(d, np.ones(d.size, np.uint32)), ii)[1])(O.InvertedIndex(O.C_FREQS_ONLY))
    table = table_for(rng, n_docs)
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 16, V.VecSimMetric_L2)
    idx.add_philox_rows(3, 0, 1000, 1)
    q = O.philox_rows(3, 1 << 40, 1, 16)[0]
    for lists_o, want in (([mk(even), mk(odd)], 0), ([mk(few), mk(even)], 3)):
        g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
        a, b, ha, hb = general_and_staged(lambda: S.HybridQuery(g, table, "BM25STD", [1.0, 1.0], [1.0, 1.0], [1.0, 1.0], n_docs, 200.0,
                                                                top_n=10, index=idx, q=q, k=10, want_hits=True))
        assert a["n_hits"] == want and len(ha) == want and len(a["top"][0]) == want
        same_hit_lists(ha, hb, 2)
    idx.free()


# ---- RSGPU_HybridTreeQuery -----------------------------------------------------------------------------------------------------
SHAPES = [
    ("term_and_or", [(T, 1.0, [0]), (U, 0.5, [1, 2, 3])]),                       # a (b|c|d)
    ("or_term_or", [(U, 2.0, [0, 1]), (T, 1.0, [2]), (U, 1.0, [3, 4])]),         # (a|b) c (d|e)
    ("and_in_and", [(I, 3.0, [0, 1]), (T, 1.0, [2])]),                           # (a b) c
    ("or_and_and_term", [(U, 1.0, [0, 1, 2]), (I, 0.7, [3, 4]), (T, 1.0, [5])]), # (a|b|c) (d e) f
    ("single_child_aggregates", [(U, 2.0, [0]), (I, 0.5, [1]), (T, 1.0, [2])]),
    ("eight_lists", [(U, 1.0, [0, 1, 2]), (T, 1.0, [3]), (U, 1.5, [4, 5]), (I, 1.0, [6, 7])]),
]


def tree_case(rng, shape, with_offsets, max_slop=None, in_order=False, n_range=(900, 2200), max_doc=2500, scorers=SCORERS,
              want_hits=True, with_knn=True):
    codec = O.C_FULL if with_offsets else O.C_FREQS_ONLY
    n_lists = sum(len(gp[2]) for gp in shape)
    built = [rand_list(rng, codec, int(rng.integers(*n_range)), max_doc, with_offsets) for _ in range(n_lists)]
    lists_o, recs = [x[0] for x in built], [x[1] for x in built]
    sizes = [l.unique_docs for l in lists_o]
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    groups = [(op, w, [g[i] for i in idx]) for op, w, idx in shape]
    n_docs = max_doc
    doc_len = rng.integers(5, 200, n_docs + 1).astype(np.uint32)
    doc_score = rng.choice([1.0, 0.5], n_docs + 1).astype(np.float32)
    max_freq = rng.integers(1, 40, n_docs + 1).astype(np.uint32)
    table = S.DocTable(doc_len, doc_score, max_freq)
    idf = [S.calculate_idf(n_docs, s) for s in sizes]
    bidf = [S.calculate_idf_bm25(n_docs, s) for s in sizes]
    w = [float(x) for x in rng.choice([1.0, 0.5, 2.0], n_lists)]
    avg = float(doc_len[1:].mean())
    idx = q = None
    if with_knn:
        idx = V.VecSimIndex(V.VecSimType_FLOAT32, 24, V.VecSimMetric_L2)
        idx.add_philox_rows(11, 0, max_doc // 2, 100)       # documents 100 .. 100 + max_doc / 2 have a vector
        q = O.philox_rows(11, 1 << 40, 1, 24)[0]
    ot = OracleTree(I, shape, recs, sizes, max_slop, in_order)
    n_hits = None
    for scorer in scorers:
        a, b, ha, hb = general_and_staged(lambda: S.HybridTreeQuery(I, groups, max_slop=max_slop, in_order=in_order, table=table,
                                                                    scorer=scorer, idf=idf, bm25_idf=bidf, weight=w, num_docs=n_docs,
                                                                    avg_doc_len=avg, top_n=10, index=idx, q=q, k=10 if with_knn else 0,
                                                                    root_weight=1.5, want_hits=want_hits))
        n_hits = a["n_hits"]
        assert n_hits == len(ot.docs), (scorer, n_hits, len(ot.docs))
        if want_hits:
            ids, fr = same_hit_lists(ha, hb, n_lists, with_records=with_offsets)
            assert ids.tolist() == ot.docs
            assert ha.leaf_order() == ot.leaf_order
        # the oracle directly: every document's result tree, the reference's order
        scored = []
        for d in ot.docs:
            node = ot.node(d, idf, bidf, w)
            node.c.weight = 1.5
            scored.append((O.score(scorer, node, float(doc_score[d]), int(max_freq[d]), int(doc_len[d]), n_docs, avg), d))
        scored.sort(key=lambda t: (-t[0], t[1]))
        want = scored[:10]
        assert a["top"][0].tolist() == [d for _, d in want], (scorer, a["top"][0], want)
        if scorer == "BM25STD.TANH":
            assert a["top"][1] == pytest.approx([s for s, _ in want], rel=1e-12)
        else:
            assert a["top"][1].tolist() == [s for s, _ in want], scorer
        if with_knn and scorer == scorers[0]:
            n_vec = max_doc // 2
            cand = np.asarray([d for d in ot.docs if 100 <= d < 100 + n_vec], np.int64)
            o = O.FlatIndex(O.F32, 24, O.L2)
            if len(cand):
                o.add_bulk(O.philox_rows(11, 0, n_vec, 24)[cand - 100], 1)      # oracle labels 1 .. m  <->  cand[0 .. m)
                li, ls = o.topk(q, 10)
                assert a["knn"][0].tolist() == cand[li.astype(np.int64) - 1].tolist()
                assert np.all(np.abs(a["knn"][1] - ls) <= 1e-4 + 1e-5 * np.abs(ls))
            else:
                assert len(a["knn"][0]) == 0
    if idx is not None:
        idx.free()
    return n_hits


@pytest.mark.parametrize("with_offsets", [False, True])
@pytest.mark.parametrize("name,shape", SHAPES)
def test_tree_query_matches_the_staged_form_and_the_oracle(name, shape, with_offsets):
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 10000 + int(with_offsets))
    assert tree_case(rng, shape, with_offsets) > 0


@pytest.mark.parametrize("max_slop,in_order", [(0, False), (2, False), (None, True), (1, True), (10, True), (40, False)])
@pytest.mark.parametrize("name,shape", [SHAPES[0], SHAPES[1], SHAPES[3]])
def test_tree_query_with_slop_and_order(name, shape, max_slop, in_order):
    """max_slop / in_order on the root intersection (Intersection::current_is_relevant, intersection.rs:205-215): a union child's
    positions are the merge of its matched terms' (proximity.rs OffsetIter::Merge); in_order pins the caller's child order"""
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 10000 + 17 * (max_slop or 0) + int(in_order))
    tree_case(rng, shape, True, max_slop, in_order, scorers=["BM25STD", "TFIDF", "BM25"])


def test_tree_query_over_many_tiles():
    """lists long enough for hundreds of tiles (the small cases above fit one or two): a (b|c) d with offsets and a window"""
    rng = np.random.default_rng(123)
    n = tree_case(rng, [(T, 1.0, [0]), (U, 0.5, [1, 2]), (T, 2.0, [3])], True, 6, False, n_range=(60_000, 90_000), max_doc=150_000,
                  scorers=["TFIDF", "BM25STD"])
    assert n > 1000


def test_tree_query_shapes_the_general_kernel_declines():
    """a root whose children are all unions has no list every hit must hold (nothing drives the probe), a root union is not an
    intersection: staged, same entry point, same answers as stage by stage.  BM25STD.NORM (the maximum over all hits = the first
    entry's score: divided on the host) takes the general kernel."""
    rng = np.random.default_rng(31)
    built = [rand_list(rng, O.C_FREQS_ONLY, int(rng.integers(900, 2000)), 2500, False) for _ in range(4)]
    g = [S.Postings.from_flat(x[0].flatten()) for x in built]
    sizes = [x[0].unique_docs for x in built]
    table = table_for(rng, 2500)
    idf = [S.calculate_idf(2500, s) for s in sizes]
    bidf = [S.calculate_idf_bm25(2500, s) for s in sizes]
    w = [1.0, 2.0, 0.5, 1.0]
    # (round 5: a root union of terms / intersections takes the tile kernel -- unless it has a union child or the scorer divides
    # by the result's slop, which differs from hit to hit in a union; round 6: also when the hit list is wanted)
    # (... and a root intersection of unions only: the smallest union drives, term by term -- round 6: with the hit list too)
    for root, groups, scorer, kw, path in ((I, [(U, 1.0, g[:2]), (U, 1.0, g[2:])], "BM25STD", {}, 2),
                                           (I, [(U, 1.0, g[:2]), (U, 1.0, g[2:])], "BM25STD", dict(want_hits=True), 2),
                                           (U, [(T, 1.0, g[:1]), (I, 1.0, g[1:3])], "BM25STD", {}, 2),
                                           (U, [(T, 1.0, g[:1]), (U, 1.0, g[1:3])], "BM25STD", {}, 0),
                                           (U, [(T, 1.0, g[:1]), (T, 1.0, g[1:2])], "TFIDF", {}, 0),
                                           (U, [(T, 1.0, g[:1]), (T, 1.0, g[1:2])], "BM25STD", dict(want_hits=True), 2),
                                           (I, [(T, 1.0, g[:1]), (U, 1.0, g[1:3])], "BM25STD.NORM", {}, 2)):
        nl = sum(len(x[2]) for x in groups)
        hq = S.HybridTreeQuery(root, groups, table=table, scorer=scorer, idf=idf[:nl], bm25_idf=bidf[:nl], weight=w[:nl], num_docs=2500,
                               avg_doc_len=150.0, top_n=10, **kw)
        hq.run()
        assert S.hybrid_path() == path, (root, scorer, kw)
        r = hq.results()
        h = S.TreeHits(root, groups)
        h.score(table, scorer, idf[:nl], bidf[:nl], w[:nl], 2500, 150.0, want_scores=False)
        ti, ts = h.topn(10)
        assert r["n_hits"] == len(h) and r["top"][0].tolist() == ti.tolist() and r["top"][1].tolist() == ts.tolist()
        if kw.get("want_hits"):
            hh = hq.take_hits()
            assert hh.read()[0].tolist() == h.read()[0].tolist() and np.array_equal(hh.read()[1], h.read()[1])


def test_tree_query_with_top_n_and_k_up_to_sixty_four():
    """top_n / k between 33 and 64 (round 5; staged before): the tiles' and the reduce kernel's bounds are the k-th of 64 bests"""
    rng = np.random.default_rng(64)
    n_docs = 150_000
    lists_o, rng = flat_corpus(n_docs, (0.3, 0.25, 0.2), 640)
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    table = table_for(rng, n_docs)
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists_o]
    bidf = [S.calculate_idf_bm25(n_docs, l.unique_docs) for l in lists_o]
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 24, V.VecSimMetric_L2)
    idx.add_philox_rows(11, 0, 60_000, 100)
    q = O.philox_rows(11, 1 << 40, 1, 24)[0]
    for root, groups in ((I, [(T, 1.0, g[:1]), (U, 0.5, g[1:])]), (U, [(T, 1.0, g[:1]), (I, 2.0, g[1:])]), (I, [(U, 1.0, g[:2]), (U, 1.0, g[2:])])):
        for scorer, top_n, k in (("BM25STD", 64, 40), ("DISMAX", 33, 64), ("BM25STD.NORM", 63, 64)):
            a, b, _, _ = general_and_staged(lambda: S.HybridTreeQuery(root, groups, table=table, scorer=scorer, idf=idf, bm25_idf=bidf,
                                                                      weight=[1.0, 2.0, 0.5], num_docs=n_docs, avg_doc_len=200.0, top_n=top_n,
                                                                      index=idx, q=q, k=k, root_weight=1.5))
            assert len(a["top"][0]) == top_n and len(a["knn"][0]) == k
    idx.free()


# ---- a root intersection of unions only: `(run|running|ran) (shoe|shoes)` (round 5) ----------------------------------------------
ALL_UNION_SHAPES = [
    ("(a|b) (c|d)", [(U, 1.0, [0, 1]), (U, 0.5, [2, 3])]),
    ("(a|b|c) (d|e) (f|g)", [(U, 2.0, [0, 1, 2]), (U, 1.0, [3, 4]), (U, 1.5, [5, 6])]),
    ("(a) (b|c)", [(U, 2.0, [0]), (U, 1.0, [1, 2])]),
]


@pytest.mark.parametrize("with_offsets,max_slop,in_order", [(False, None, False), (True, None, False), (True, 6, False), (True, None, True)])
@pytest.mark.parametrize("name,shape", ALL_UNION_SHAPES)
def test_root_of_unions_only_is_driven_by_its_smallest_union(name, shape, with_offsets, max_slop, in_order):
    """The stemmer's expansions on every term: no list every hit holds.  The child union with the fewest postings drives -- one pass
    of the tile kernel per term of it, a document an earlier term of that union holds belongs to that term's pass (veto), one reduce
    kernel over all passes' tiles: against the staged form bit for bit and the CPU oracle (tree_case)."""
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 10000 + int(with_offsets) + (max_slop or 0) + 3 * int(in_order))
    # (round 6: WITH the hit list -- every pass's hits in doc-id order, the runs merged by doc id behind the reduce kernel; ids,
    # per-leaf frequencies, term records and leaf order against the staged list and the oracle's)
    assert tree_case(rng, shape, with_offsets, max_slop=max_slop, in_order=in_order, want_hits=True) > 0


def test_root_of_unions_over_many_tiles():
    rng = np.random.default_rng(41)
    n = tree_case(rng, [(U, 1.0, [0, 1]), (U, 1.0, [2, 3, 4])], False, n_range=(10_000, 30_000), max_doc=200_000, scorers=["BM25STD", "DISMAX"],
                  want_hits=True)
    assert n > 2000
    # a mass tie (DOCSCORE: three distinct scores) handed back by the reduce kernel (forced: a cap of 4 survivors): settled across
    # the passes by doc id, against the oracle's (score descending, doc id ascending)
    try:
        knob("hybrid_surv_cap", 4)
        rng = np.random.default_rng(42)
        assert tree_case(rng, [(U, 1.0, [0, 1]), (U, 1.0, [2, 3])], False, n_range=(10_000, 30_000), max_doc=200_000, scorers=["DOCSCORE", "DISMAX"],
                         want_hits=False) > 1000
    finally:
        knob("hybrid_surv_cap", 4096)


# ---- a root UNION on the tile path (round 5) ----------------------------------------------------------------------------------------
UNION_SHAPES = [
    ("a|b", [(T, 1.0, [0]), (T, 1.0, [1])]),
    ("a|b|c|d", [(T, 1.0, [0]), (T, 1.0, [1]), (T, 1.0, [2]), (T, 1.0, [3])]),
    ("(a b)|(c d)", [(I, 2.0, [0, 1]), (I, 0.5, [2, 3])]),
    ("a|(b c)|d", [(T, 1.0, [0]), (I, 1.5, [1, 2]), (T, 1.0, [3])]),
    ("(a b c)|(d e)|f|(g h)", [(I, 1.0, [0, 1, 2]), (I, 2.0, [3, 4]), (T, 1.0, [5]), (I, 0.25, [6, 7])]),
]


@pytest.mark.parametrize("with_offsets", [False, True])
@pytest.mark.parametrize("name,shape", UNION_SHAPES)
def test_root_union_takes_the_tile_kernel(name, shape, with_offsets):
    """`hello | world =>[KNN ...]`: the commonest shape the staged pipeline still held (0.69 ms against 0.13 for an AND).  One
    pass of the tile kernel per child (its shortest list drives; a document an earlier child matches belongs to that child's
    pass; a later child intersection counts only when it matches as a whole), ONE reduce: against the staged form bit for bit
    and the CPU oracle directly -- set algebra, the result tree Union{matched children} scored by the oracle's scorers in the
    reference's order, O.FlatIndex distances"""
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 10000 + int(with_offsets))
    codec = O.C_FULL if with_offsets else O.C_FREQS_ONLY
    n_lists = sum(len(gp[2]) for gp in shape)
    max_doc = 30_000
    built = [rand_list(rng, codec, int(rng.integers(3000, 9000)), max_doc, with_offsets) for _ in range(n_lists)]
    recs, sizes = [x[1] for x in built], [x[0].unique_docs for x in built]
    g = [S.Postings.from_flat(x[0].flatten()) for x in built]
    groups = [(op, wt, [g[i] for i in ix]) for op, wt, ix in shape]
    doc_len = rng.integers(5, 200, max_doc + 1).astype(np.uint32)
    doc_score = rng.choice([1.0, 0.5], max_doc + 1).astype(np.float32)
    max_freq = rng.integers(1, 40, max_doc + 1).astype(np.uint32)
    table = S.DocTable(doc_len, doc_score, max_freq)
    idf = [S.calculate_idf(max_doc, s_) for s_ in sizes]
    bidf = [S.calculate_idf_bm25(max_doc, s_) for s_ in sizes]
    w = [float(x) for x in rng.choice([1.0, 0.5, 2.0], n_lists)]
    avg = float(doc_len[1:].mean())
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 24, V.VecSimMetric_L2)
    n_vec = 12_000
    idx.add_philox_rows(11, 0, n_vec, 5000)                     # documents 5 000 .. 16 999 have a vector
    idx.delete_vector(6000)                                     # (the device label table, not identity arithmetic)
    q = O.philox_rows(11, 1 << 40, 1, 24)[0]
    ot = OracleTree(U, shape, recs, sizes)
    assert len(ot.docs) > 1500
    for scorer in ("BM25STD", "BM25STD.TANH", "DOCSCORE", "DISMAX", "BM25STD.NORM"):
        want_hits = scorer in ("BM25STD", "DISMAX")             # (round 6: the hit list of a multi-pass query -- runs merged by doc id)
        a, b, ha, hb = general_and_staged(lambda: S.HybridTreeQuery(U, groups, table=table, scorer=scorer, idf=idf, bm25_idf=bidf, weight=w,
                                                                    num_docs=max_doc, avg_doc_len=avg, top_n=10, index=idx, q=q, k=10,
                                                                    root_weight=1.5, want_hits=want_hits))
        assert a["n_hits"] == len(ot.docs), (scorer, a["n_hits"], len(ot.docs))
        if want_hits:
            ids, _ = same_hit_lists(ha, hb, n_lists, with_records=with_offsets)
            assert ids.tolist() == ot.docs
        if scorer == "BM25STD.NORM":
            continue                                             # (held to the staged form above; the oracle has no such scorer)
        scored = []
        for d in ot.docs:
            node = ot.node(d, idf, bidf, w)
            node.c.weight = 1.5
            scored.append((O.score(scorer, node, float(doc_score[d]), int(max_freq[d]), int(doc_len[d]), max_doc, avg), d))
        scored.sort(key=lambda t: (-t[0], t[1]))
        assert a["top"][0].tolist() == [d for _, d in scored[:10]], (scorer, a["top"][0], scored[:10])
        if scorer == "BM25STD.TANH":
            assert a["top"][1] == pytest.approx([x for x, _ in scored[:10]], rel=1e-12)
        else:
            assert a["top"][1].tolist() == [x for x, _ in scored[:10]], scorer
    cand = np.asarray([d for d in ot.docs if 5000 <= d < 5000 + n_vec and d != 6000], np.int64)
    o = O.FlatIndex(O.F32, 24, O.L2)
    o.add_bulk(O.philox_rows(11, 0, n_vec, 24)[cand - 5000], 1)
    li, ls = o.topk(q, 10)
    assert a["knn"][0].tolist() == cand[li.astype(np.int64) - 1].tolist()
    assert np.all(np.abs(a["knn"][1] - ls) <= 1e-4 + 1e-5 * np.abs(ls))
    idx.free()


def test_root_union_over_many_tiles_and_one_sided_children():
    """long lists (hundreds of tiles per pass), a child that matches almost everything next to one that matches almost nothing,
    and a mass tie at the reduce kernel's bound settled by the exact select"""
    rng = np.random.default_rng(5)
    n_docs = 900_000
    lists_o, rng = flat_corpus(n_docs, (0.6, 0.002, 0.3, 0.45), 909)
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    table = table_for(rng, n_docs)
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists_o]
    bidf = [S.calculate_idf_bm25(n_docs, l.unique_docs) for l in lists_o]
    w = [1.0, 0.5, 2.0, 1.5]
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 48, V.VecSimMetric_L2)
    idx.add_philox_rows(7, 0, 200_000, 1)
    q = O.philox_rows(7, 1 << 40, 1, 48)[0]
    for groups in ([(T, 1.0, g[:1]), (T, 1.0, g[1:2])], [(T, 1.0, g[1:2]), (I, 2.0, g[2:4]), (T, 1.0, g[:1])]):
        order = [x for _, _, gl in groups for x in gl]
        sel = [g.index(x) for x in order]
        kw = dict(table=table, idf=[idf[i] for i in sel], bm25_idf=[bidf[i] for i in sel], weight=[w[i] for i in sel], num_docs=n_docs,
                  avg_doc_len=200.0, index=idx, q=q, root_weight=1.25)
        for scorer, top_n, k, cap in (("BM25STD", 10, 10, 2048), ("DOCSCORE", 32, 5, 2048), ("DOCSCORE", 10, 10, 4)):
            try:
                knob("hybrid_surv_cap", cap)
                # (cap 4: the reduce kernel hands a mass tie back; the exact select's (key, position) order is doc-id order inside
                # one pass only: the entries at the cut's key are settled by a second select over their doc ids -- path 2 still)
                a, b, ha, hb = general_and_staged(lambda: S.HybridTreeQuery(U, groups, scorer=scorer, top_n=top_n, k=k, want_hits=cap == 2048, **kw))
            finally:
                knob("hybrid_surv_cap", 4096)
            assert a["n_hits"] > 500_000 and len(a["top"][0]) == top_n and len(a["knn"][0]) == k
            if ha is not None:                                   # hundreds of tiles per run, runs of very different lengths
                ids, _ = same_hit_lists(ha, hb, len(order))
                assert len(ids) == a["n_hits"] and np.all(np.diff(ids.astype(np.int64)) > 0)
    idx.free()


# ---- RSGPU_EvalTree through the tile kernel -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("with_offsets,max_slop,in_order", [(False, None, False), (True, None, False), (True, 3, False), (True, None, True)])
@pytest.mark.parametrize("name,shape", SHAPES)
def test_eval_tree_builds_the_same_hit_list_with_the_tile_kernel(name, shape, with_offsets, max_slop, in_order):
    """RSGPU_EvalTree (and with it the iterator seam's tree iterators) over a root intersection with an aggregate child: the
    tile kernel probes every list in place; the staged evaluation builds each child's hit list first.  Same doc ids, per-term
    frequencies, term records, leaf order and -- scored stage by stage -- the same doubles from every scorer."""
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 10000 + 3 * int(with_offsets) + (max_slop or 0) + 5 * int(in_order))
    codec = O.C_FULL if with_offsets else O.C_FREQS_ONLY
    n_lists = sum(len(gp[2]) for gp in shape)
    built = [rand_list(rng, codec, int(rng.integers(900, 2200)), 2500, with_offsets) for _ in range(n_lists)]
    g = [S.Postings.from_flat(x[0].flatten()) for x in built]
    sizes = [x[0].unique_docs for x in built]
    groups = [(op, w, [g[i] for i in idx]) for op, w, idx in shape]
    try:
        knob("hybrid_tree_tiles", 1)
        ha = S.TreeHits(I, groups, max_slop=max_slop, in_order=in_order)
        assert S.hybrid_path() == 2
        knob("hybrid_tree_tiles", 0)
        hb = S.TreeHits(I, groups, max_slop=max_slop, in_order=in_order)
        assert S.hybrid_path() == 0
    finally:
        knob("hybrid_tree_tiles", 1)
    ids, _ = same_hit_lists(ha, hb, n_lists, with_records=with_offsets)
    ot = OracleTree(I, shape, [x[1] for x in built], sizes, max_slop, in_order)
    assert ids.tolist() == ot.docs
    table = table_for(rng, 2500)
    idf = [S.calculate_idf(2500, s) for s in sizes]
    bidf = [S.calculate_idf_bm25(2500, s) for s in sizes]
    w = [float(x) for x in rng.choice([1.0, 0.5, 2.0], n_lists)]
    for scorer in SCORERS + ["BM25STD.NORM"]:
        sa = ha.score(table, scorer, idf, bidf, w, 2500, 150.0, root_weight=1.5)
        sb = hb.score(table, scorer, idf, bidf, w, 2500, 150.0, root_weight=1.5)
        assert np.array_equal(sa, sb), scorer


# ---- BM25STD.NORM on the tile paths ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [None, SHAPES[0][1], SHAPES[3][1]])
def test_bm25std_norm_is_ranked_as_bm25std_and_divided_by_the_first_score(shape):
    """SCORER BM25STD.NORM = BM25STD, then every score over the largest one (RPMaxScoreNormalizer, result_processor.c:1770-1812):
    the tile kernels rank BM25STD, the host divides by the first entry's score -- against the staged pipeline (score_max_kernel +
    score_normalize_kernel over ALL hits, then the selection), bit for bit, and against BM25STD's own top list."""
    rng = np.random.default_rng(77 if shape is None else len(shape))
    if shape is None:
        lists_o, rng = flat_corpus(400_000, (0.4, 0.5, 0.3), 78)
        g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
        n_docs, sizes = 400_000, [l.unique_docs for l in lists_o]
        make = lambda scorer, n: S.HybridQuery(g, table, scorer, idf, bidf, w, n_docs, 200.0, top_n=n, root_weight=0.7)
        want_path = 1
    else:
        n_lists = sum(len(gp[2]) for gp in shape)
        built = [rand_list(rng, O.C_FREQS_ONLY, int(rng.integers(900, 2200)), 2500, False) for _ in range(n_lists)]
        g = [S.Postings.from_flat(x[0].flatten()) for x in built]
        n_docs, sizes = 2500, [x[0].unique_docs for x in built]
        groups = [(op, wt, [g[i] for i in idx]) for op, wt, idx in shape]
        make = lambda scorer, n: S.HybridTreeQuery(I, groups, table=table, scorer=scorer, idf=idf, bm25_idf=bidf, weight=w, num_docs=n_docs,
                                                   avg_doc_len=200.0, top_n=n, root_weight=0.7)
        want_path = 2
    table = table_for(rng, n_docs)
    idf = [S.calculate_idf(n_docs, s_) for s_ in sizes]
    bidf = [S.calculate_idf_bm25(n_docs, s_) for s_ in sizes]
    w = [1.0, 0.5, 2.0, 1.5, 1.0, 0.25][:len(sizes)]
    for n in (1, 10, 31):
        hq = make("BM25STD.NORM", n)
        hq.run()
        assert S.hybrid_path() == want_path
        a = hq.results()
        try:
            knob("hybrid_tiles", 0)
            hq.run()
            assert S.hybrid_path() == 0
            b = hq.results()
        finally:
            knob("hybrid_tiles", 1)
        assert a["n_hits"] == b["n_hits"] and a["top"][0].tolist() == b["top"][0].tolist() and a["top"][1].tolist() == b["top"][1].tolist()
        raw = make("BM25STD", n)
        raw.run()
        r = raw.results()
        assert a["top"][0].tolist() == r["top"][0].tolist() and a["top"][1].tolist() == (r["top"][1] / r["top"][1][0]).tolist()
        assert a["top"][1][0] == 1.0


# ---- NOT children ------------------------------------------------------------------------------------------------------------------
NOT_SHAPES = [
    ("a -b", [(T, 1.0, [0])], [(1.0, [1])]),
    ("a b -c", [(T, 1.0, [0]), (T, 1.0, [1])], [(2.0, [2])]),
    ("a (b|c) -d -e", [(T, 1.0, [0]), (U, 0.5, [1, 2])], [(1.0, [3]), (1.0, [4])]),
    ("(a b) c -(d|e)", [(I, 3.0, [0, 1]), (T, 1.0, [2])], [(1.0, [3, 4])]),
]


@pytest.mark.parametrize("with_offsets,max_slop,in_order", [(False, None, False), (True, None, False), (True, 4, False), (True, None, True)])
@pytest.mark.parametrize("name,pos,neg", NOT_SHAPES)
def test_not_children_against_the_oracle(name, pos, neg, with_offsets, max_slop, in_order):
    """`a -b` under a root intersection: the excluded lists veto a candidate in the tile kernel; the result is the positive
    children's PLUS a virtual child of frequency 0 per NOT (not.rs:106-118): it adds nothing to any sum and has no offsets, but
    IndexResult_MinOffsetDelta counts it (offset-less slop = children - 1).  No staged twin exists (RSGPU_EvalTree has no NOT
    node): the CPU oracle directly -- set algebra, the oracle's proximity test over the positive children, its scorers over
    Intersection{..., Virtual} in the reference's order."""
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 10000 + 3 * int(with_offsets) + (max_slop or 0) + 5 * int(in_order))
    codec = O.C_FULL if with_offsets else O.C_FREQS_ONLY
    n_lists = sum(len(gp[2]) for gp in pos) + sum(len(x[1]) for x in neg)
    built = [rand_list(rng, codec, int(rng.integers(700, 1800)), 2500, with_offsets) for _ in range(n_lists)]
    recs, sizes = [x[1] for x in built], [x[0].unique_docs for x in built]
    g = [S.Postings.from_flat(x[0].flatten()) for x in built]
    groups = [(op, wt, [g[i] for i in idx]) for op, wt, idx in pos] + [(S.OP_NOT, wt, [g[i] for i in idx]) for wt, idx in neg]
    n_docs = 2500
    doc_len = rng.integers(5, 200, n_docs + 1).astype(np.uint32)
    doc_score = rng.choice([1.0, 0.5], n_docs + 1).astype(np.float32)
    max_freq = rng.integers(1, 40, n_docs + 1).astype(np.uint32)
    table = S.DocTable(doc_len, doc_score, max_freq)
    idf = [S.calculate_idf(n_docs, s_) for s_ in sizes]
    bidf = [S.calculate_idf_bm25(n_docs, s_) for s_ in sizes]
    w = [float(x) for x in rng.choice([1.0, 0.5, 2.0], n_lists)]
    avg = float(doc_len[1:].mean())
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 24, V.VecSimMetric_L2)
    idx.add_philox_rows(11, 0, 1200, 100)
    q = O.philox_rows(11, 1 << 40, 1, 24)[0]
    ot = OracleTree(I, pos, recs, sizes, max_slop, in_order)
    gone = set().union(*[set(recs[i]) for _, ix in neg for i in ix])
    docs = [d for d in ot.docs if d not in gone]
    assert len(docs) < len(ot.docs) or not ot.docs
    # round 5: the hit list of such a query (hits_out) -- the positive children's columns, the virtual children in its result tree
    hq = S.HybridTreeQuery(I, groups, max_slop=max_slop, in_order=in_order, table=table, scorer="BM25STD", idf=idf, bm25_idf=bidf, weight=w,
                           num_docs=n_docs, avg_doc_len=avg, top_n=10, index=idx, q=q, k=10, root_weight=1.5, want_hits=True)
    hq.run()
    assert S.hybrid_path() == 2
    hh = hq.take_hits()
    hi, hf = hh.read()
    assert hi.tolist() == docs
    positive = [i for _, _, ix in pos for i in ix]
    for li in range(n_lists):
        assert hf[li].tolist() == [(recs[li][d][0] if (li in positive and d in recs[li]) else 0) for d in docs], li
    assert sorted(hh.leaf_order()) == sorted(positive)
    for scorer in SCORERS:
        hq = S.HybridTreeQuery(I, groups, max_slop=max_slop, in_order=in_order, table=table, scorer=scorer, idf=idf, bm25_idf=bidf, weight=w,
                               num_docs=n_docs, avg_doc_len=avg, top_n=10, index=idx, q=q, k=10, root_weight=1.5)
        hq.run()
        assert S.hybrid_path() == 2
        a = hq.results()
        assert a["n_hits"] == len(docs), (scorer, a["n_hits"], len(docs))
        scored = []
        for d in docs:
            kids = ot.node(d, idf, bidf, w).kids + [O.Node(O.R_VIRTUAL, wt, 0) for wt, _ in neg]
            node = O.intersection(kids)
            node.c.weight = 1.5
            scored.append((O.score(scorer, node, float(doc_score[d]), int(max_freq[d]), int(doc_len[d]), n_docs, avg), d))
        # (RSGPU_Hits_Score over the returned list: every hit's score, not just the ranked ten)
        hs = hh.score(table, scorer, idf, bidf, w, n_docs, avg, root_weight=1.5)
        if scorer == "BM25STD.TANH":
            assert hs == pytest.approx([x for x, _ in scored], rel=1e-12)
        else:
            assert hs.tolist() == [x for x, _ in scored], scorer
        scored.sort(key=lambda t: (-t[0], t[1]))
        assert a["top"][0].tolist() == [d for _, d in scored[:10]], (scorer, a["top"][0], scored[:10])
        if scorer == "BM25STD.TANH":
            assert a["top"][1] == pytest.approx([x for x, _ in scored[:10]], rel=1e-12)
        else:
            assert a["top"][1].tolist() == [x for x, _ in scored[:10]], scorer
    cand = np.asarray([d for d in docs if 100 <= d < 1300], np.int64)
    if len(cand):
        o = O.FlatIndex(O.F32, 24, O.L2)
        o.add_bulk(O.philox_rows(11, 0, 1200, 24)[cand - 100], 1)
        li, ls = o.topk(q, 10)
        assert a["knn"][0].tolist() == cand[li.astype(np.int64) - 1].tolist()
    else:
        assert len(a["knn"][0]) == 0
    idx.free()


def test_not_children_with_an_empty_excluded_list_and_under_a_mass_tie():
    """round-4 advisor: `a -b` with an EMPTY b is `a` (plus the virtual child): the empty list is simply not probed; and a mass tie
    at the reduce kernel's bound (forced with hybrid_surv_cap = 4) is settled by the exact select over the tiles' lists -- a NOT
    query has no staged form to fall back to, so whether it succeeded used to depend on its data"""
    rng = np.random.default_rng(17)
    built = [rand_list(rng, O.C_FREQS_ONLY, int(rng.integers(1500, 2000)), 2500, False) for _ in range(2)]
    recs, sizes = [x[1] for x in built], [x[0].unique_docs for x in built]
    g = [S.Postings.from_flat(x[0].flatten()) for x in built]
    empty = S.Postings.from_flat(O.InvertedIndex(O.C_FREQS_ONLY).flatten())
    assert empty.num_entries == 0
    n_docs = 2500
    table = table_for(rng, n_docs)
    dl, ds, mf = table._arrays
    idf = [S.calculate_idf(n_docs, s_) for s_ in sizes] + [0.0]
    bidf = [S.calculate_idf_bm25(n_docs, s_) for s_ in sizes] + [0.0]
    w = [1.0, 2.0, 0.0]
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 24, V.VecSimMetric_L2)
    idx.add_philox_rows(11, 0, 1200, 100)
    q = O.philox_rows(11, 1 << 40, 1, 24)[0]
    ot = OracleTree(I, [(T, 1.0, [0]), (T, 1.0, [1])], recs, sizes)
    for scorer in ("BM25STD", "DOCSCORE", "TFIDF"):
        scored = []
        for d in ot.docs:
            node = O.intersection(ot.node(d, idf, bidf, w).kids + [O.Node(O.R_VIRTUAL, 1.0, 0)])
            node.c.weight = 1.5
            scored.append((O.score(scorer, node, float(ds[d]), int(mf[d]), int(dl[d]), n_docs, 150.0), d))
        scored.sort(key=lambda t: (-t[0], t[1]))
        hq = S.HybridTreeQuery(I, [(T, 1.0, g[:1]), (T, 1.0, g[1:2]), (S.OP_NOT, 1.0, [empty])], table=table, scorer=scorer, idf=idf,
                               bm25_idf=bidf, weight=w, num_docs=n_docs, avg_doc_len=150.0, top_n=10, index=idx, q=q, k=10, root_weight=1.5)
        for cap in (2048, 4):
            try:
                knob("hybrid_surv_cap", cap)
                hq.run()
            finally:
                knob("hybrid_surv_cap", 4096)
            assert S.hybrid_path() == 2
            a = hq.results()
            assert a["n_hits"] == len(ot.docs)
            assert a["top"][0].tolist() == [d for _, d in scored[:10]], (scorer, cap)
            assert a["top"][1].tolist() == [x for x, _ in scored[:10]], (scorer, cap)
            assert len(a["knn"][0]) == 10
    idx.free()


def test_not_children_where_the_tile_kernel_cannot_run():
    rng = np.random.default_rng(3)
    built = [rand_list(rng, O.C_FREQS_ONLY, 800, 2500, False) for _ in range(3)]
    g = [S.Postings.from_flat(x[0].flatten()) for x in built]
    table = table_for(rng, 2500)
    ones = [1.0] * 3
    # a NOT child under a root of unions only WITH hits_out (refused until round 6): the union drives term by term, the runs are
    # merged by doc id -- (a | b) - c in doc-id order, a's / b's frequency columns (0 where the term is absent)
    hq = S.HybridTreeQuery(I, [(U, 1.0, g[:2]), (S.OP_NOT, 1.0, g[2:])], table=table, scorer="BM25STD", idf=ones, bm25_idf=ones, weight=ones,
                           num_docs=2500, avg_doc_len=150.0, top_n=10, want_hits=True)
    hq.run()
    assert S.hybrid_path() == 2
    docs_ = [set(x[1]) for x in built]
    want = sorted((docs_[0] | docs_[1]) - docs_[2])
    hh = hq.take_hits()
    hi, hf = hh.read()
    assert hi.tolist() == want
    for col, l in enumerate(hh.leaf_order()):
        assert hf[col].tolist() == [built[l][1][d][0] if d in built[l][1] else 0 for d in want], (col, l)
    # (without hits_out: round 5)
    hq = S.HybridTreeQuery(I, [(U, 1.0, g[:2]), (S.OP_NOT, 1.0, g[2:])], table=table, scorer="BM25STD", idf=ones, bm25_idf=ones, weight=ones,
                           num_docs=2500, avg_doc_len=150.0, top_n=10)
    hq.run()
    assert S.hybrid_path() == 2
    docs = [set(x[1]) for x in built]
    assert hq.results()["n_hits"] == len((docs[0] | docs[1]) - docs[2])
    with pytest.raises(RuntimeError):
        S.TreeHits(I, [(T, 1.0, g[:1]), (S.OP_NOT, 1.0, g[1:2])])


# ---- RSGPU_IntersectEx with a window through the tile kernel ---------------------------------------------------------------------
@pytest.mark.parametrize("n_lists", [2, 3, 5])
@pytest.mark.parametrize("max_slop,in_order", [(0, False), (3, False), (None, True), (0, True), (12, True)])
def test_phrase_intersection_builds_the_same_hit_list_with_the_tile_kernel(n_lists, max_slop, in_order):
    """`"hello world"` (slop 0, in order) and looser windows over Full-codec lists: the tile kernel's list against the staged
    probe -> prox_filter -> scan -> write, and against the oracle's intersection with the same window"""
    rng = np.random.default_rng(1000 * n_lists + 10 * (max_slop or 0) + int(in_order))
    built = [rand_list(rng, O.C_FULL, int(rng.integers(2500, 6000)), 8000, True) for _ in range(n_lists)]
    lists_o = [x[0] for x in built]
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    try:
        knob("hybrid_tree_tiles", 1)
        ha = S.intersect(g, max_slop=max_slop, in_order=in_order)
        assert S.hybrid_path() == 2
        knob("hybrid_tree_tiles", 0)
        hb = S.intersect(g, max_slop=max_slop, in_order=in_order)
        assert S.hybrid_path() == 0
    finally:
        knob("hybrid_tree_tiles", 1)
    ids, fr = same_hit_lists(ha, hb, n_lists, with_records=True)
    oi, of, _ = O.intersect_ex(lists_o, max_slop, in_order)
    assert ids.tolist() == oi.tolist() and np.array_equal(fr, of)
    table = table_for(rng, 8000)
    sizes = [l.unique_docs for l in lists_o]
    idf = [S.calculate_idf(8000, s_) for s_ in sizes]
    bidf = [S.calculate_idf_bm25(8000, s_) for s_ in sizes]
    for scorer in ("TFIDF", "BM25", "BM25STD"):
        assert np.array_equal(ha.score(table, scorer, idf, bidf, [1.0] * n_lists, 8000, 150.0), hb.score(table, scorer, idf, bidf, [1.0] * n_lists, 8000, 150.0))
