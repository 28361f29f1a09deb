"""GPU: RSGPU_HybridTreeNodesQuery -- the hybrid query over a query tree of ANY depth (the filter the reference hands its hybrid
iterator nests freely: src/iterators/hybrid_reader.c:625; `a ((b c)|d)`, `a (b|(c d)) (e|f)`).  The general tile kernel takes a
root intersection over at most eight lists with a term to drive it when its predicate -- sets of lists of which ONE must match,
nested intersections under a union that count only when they match as a WHOLE -- expresses the tree (csrc/search_abi.cpp
tree_groups), and folds the score over the whole result tree in registers (score_one<true, 0, 4>).  Held to
(a) the staged form behind the same entry point (knob hybrid_tree_tiles = 0: RSGPU_EvalTreeNodes + RSGPU_Hits_Score / _TopN /
_KnnRerank, which tests/test_gpu_tree.py pins to the oracle) BIT FOR BIT: hit count, top-N ids and scores, KNN ids and distances;
(b) the CPU oracle directly: DeepOracle's set algebra (children in the order an intersection iterates them: ascending estimate x
sort weight, stable -- intersection.rs:94-119; a union's result holds the matched children only -- union_flat.rs:297-320), the
oracle's result-tree scorers in the reference's order (score descending, doc id ascending), O.FlatIndex distances."""
import zlib

import numpy as np
import pytest

import oracle as O
from redisearch_amd import search as S
from redisearch_amd import vecsim as V
from tests.test_gpu_hybrid_general import general_and_staged, knob, same_hit_lists, table_for
from tests.test_gpu_tree import DeepOracle, rand_list

pytestmark = pytest.mark.gpu
SCORERS = ["BM25STD", "BM25STD.TANH", "BM25", "TFIDF", "TFIDF.DOCNORM", "DOCSCORE", "DISMAX"]
SLOP_DEPENDENT = {"BM25", "TFIDF", "TFIDF.DOCNORM"}


def t(i):
    return ("t", i)


NESTED = [
    # name, tree, number of lists
    ("a ((b c)|d)", ("and", 1.0, [t(0), ("or", 0.5, [("and", 2.0, [t(1), t(2)]), t(3)])]), 4),
    ("a (b|(c d)) (e|f)", ("and", 1.0, [t(0), ("or", 1.5, [t(1), ("and", 0.7, [t(2), t(3)])]), ("or", 2.0, [t(4), t(5)])]), 6),
    ("a (b (c|d))", ("and", 1.0, [t(0), ("and", 0.7, [t(1), ("or", 1.5, [t(2), t(3)])])]), 4),
    ("(a (b|c)) d", ("and", 1.0, [("and", 3.0, [t(0), ("or", 1.0, [t(1), t(2)])]), t(3)]), 4),
    ("a ((b|c)|(d e))", ("and", 1.0, [t(0), ("or", 1.0, [("or", 0.5, [t(1), t(2)]), ("and", 2.0, [t(3), t(4)])])]), 5),
    ("a (b ((c d)|e))", ("and", 1.0, [t(0), ("and", 1.25, [t(1), ("or", 0.5, [("and", 2.0, [t(2), t(3)]), t(4)])])]), 5),
    ("a (b|(c d)) ((e f)|g|h)", ("and", 1.0, [t(0), ("or", 1.0, [t(1), ("and", 0.5, [t(2), t(3)])]),
                                              ("or", 2.0, [("and", 1.5, [t(4), t(5)]), t(6), t(7)])]), 8),
    ("single-child aggregates", ("and", 1.0, [t(0), ("or", 2.0, [("and", 0.5, [t(1)])]), ("and", 1.5, [("or", 0.5, [t(2)])])]), 3),
    # (no term every hit holds: the plain union drives, term by term)
    ("(a|b) ((c d)|e)", ("and", 1.0, [("or", 1.0, [t(0), t(1)]), ("or", 1.0, [("and", 1.0, [t(2), t(3)]), t(4)])]), 5),
    ("a ((b (c d))|e)", ("and", 1.0, [t(0), ("or", 1.0, [("and", 2.0, [t(1), ("and", 0.5, [t(2), t(3)])]), t(4)])]), 5),
    # round 6 (staged until then): more than four levels; a union below an intersection below a union -- the match is folded over
    # the result tree inside the kernel (HybridTreeArgs::tree_pred)
    ("five levels", ("and", 1.0, [t(0), ("and", 1.0, [t(1), ("and", 1.0, [t(2), ("and", 1.0, [t(3), ("or", 1.0, [t(4), t(5)])])])])]), 6),
    ("a (b|(c (d|e)))... as ((b (c|d))|e)", ("and", 1.0, [t(0), ("or", 1.0, [("and", 1.0, [t(1), ("or", 1.0, [t(2), t(3)])]), t(4)])]), 5),
    ("(a|((b|c) d)) e", ("and", 1.0, [("or", 1.5, [t(0), ("and", 2.0, [("or", 0.5, [t(1), t(2)]), t(3)])]), t(4)]), 5),
    ("seven levels, unions and intersections in turn",
     ("and", 1.0, [t(0), ("or", 0.5, [t(1), ("and", 2.0, [t(2), ("or", 1.5, [t(3), ("and", 0.7, [t(4), ("or", 1.0, [t(5), ("and", 3.0, [t(6), t(7)])])])])])])]), 8),
    ("a ((b|c) (d|(e f)))|g)", ("and", 1.0, [t(0), ("or", 1.0, [("and", 1.0, [("or", 2.0, [t(1), t(2)]), ("or", 0.5, [t(3), ("and", 1.0, [t(4), t(5)])])]), t(6)])]), 7),
]


def nested_case(rng, tree, n_lists, with_offsets, want_path=2, scorers=SCORERS, n_range=(900, 2200), max_doc=2500, want_hits=False, deep=True):
    codec = O.C_FULL if with_offsets else O.C_FREQS_ONLY
    built = [rand_list(rng, codec, int(rng.integers(*n_range)), max_doc, with_offsets) for _ in range(n_lists)]
    lists_o, recs = [x[0] for x in built], [x[1] for x in built]
    sizes = [l.unique_docs for l in lists_o]
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    n_docs = max_doc
    doc_len = rng.integers(5, 200, n_docs + 1).astype(np.uint32)
    doc_score = rng.choice([1.0, 0.5], n_docs + 1).astype(np.float32)
    max_freq = rng.integers(1, 40, n_docs + 1).astype(np.uint32)
    table = S.DocTable(doc_len, doc_score, max_freq)
    idf = [S.calculate_idf(n_docs, s) for s in sizes]
    bidf = [S.calculate_idf_bm25(n_docs, s) for s in sizes]
    w = [float(x) for x in rng.choice([1.0, 0.5, 2.0], n_lists)]
    avg = float(doc_len[1:].mean())
    n_vec = max_doc // 2
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 24, V.VecSimMetric_L2)
    idx.add_philox_rows(11, 0, n_vec, 100)                  # documents 100 .. 100 + max_doc / 2 have a vector
    q = O.philox_rows(11, 1 << 40, 1, 24)[0]
    ot = DeepOracle(tree, recs, sizes)
    for scorer in scorers:
        # (a scorer that divides by the slop reads the term offsets through the nested children: staged until round 6 -- a child's
        # offsets are its leaves' in the result, merged: the two-level proximity code over the children's leaf ranges)
        path = want_path
        a, b, ha, hb = general_and_staged(lambda: S.HybridNodesQuery(tree, g, table=table, scorer=scorer, idf=idf, bm25_idf=bidf, weight=w,
                                                                     num_docs=n_docs, avg_doc_len=avg, top_n=10, index=idx, q=q, k=10,
                                                                     root_weight=1.5, want_hits=want_hits), want_path=path)
        assert a["n_hits"] == len(ot.docs), (scorer, a["n_hits"], len(ot.docs))
        if want_hits:
            ids, _ = same_hit_lists(ha, hb, n_lists, with_records=with_offsets)
            assert ids.tolist() == list(ot.docs)
        scored = []
        for d in ot.docs:
            node = ot.node(ot.tree, d, idf, bidf, w)
            node.c.weight = 1.5
            scored.append((O.score(scorer, node, float(doc_score[d]), int(max_freq[d]), int(doc_len[d]), n_docs, avg), d))
        scored.sort(key=lambda x: (-x[0], x[1]))
        want = scored[:10]
        assert a["top"][0].tolist() == [d for _, d in want], (scorer, a["top"][0], want)
        if scorer == "BM25STD.TANH":
            assert a["top"][1] == pytest.approx([s for s, _ in want], rel=1e-12)
        else:
            assert a["top"][1].tolist() == [s for s, _ in want], scorer
        if scorer == scorers[0]:
            cand = np.asarray([d for d in ot.docs if 100 <= d < 100 + n_vec], np.int64)
            if len(cand):
                o = O.FlatIndex(O.F32, 24, O.L2)
                o.add_bulk(O.philox_rows(11, 0, n_vec, 24)[cand - 100], 1)
                li, ls = o.topk(q, 10)
                assert a["knn"][0].tolist() == cand[li.astype(np.int64) - 1].tolist()
                assert np.all(np.abs(a["knn"][1] - ls) <= 1e-4 + 1e-5 * np.abs(ls))
            else:
                assert len(a["knn"][0]) == 0
    idx.free()
    return len(ot.docs)


@pytest.mark.parametrize("with_offsets", [False, True])
@pytest.mark.parametrize("name,tree,n_lists", NESTED)
def test_nested_trees_take_the_tile_kernel(name, tree, n_lists, with_offsets):
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 10000 + int(with_offsets))
    assert nested_case(rng, tree, n_lists, with_offsets) > 0


@pytest.mark.parametrize("max_slop,in_order", [(4, False), (None, True), (12, True)])
def test_the_roots_window_over_nested_children(max_slop, in_order):
    """`a ((b c)|d)` SLOP n / INORDER on the ROOT (round 6; staged until then): IndexResult_IsWithinRange reads a nested child's offsets
    through the aggregate -- its leaves' positions, merged -- so the window is the two-level one over the children's leaf ranges"""
    rng = np.random.default_rng(600 + (max_slop or 0) + int(in_order))
    for tree, n_lists in ((("and", 1.0, [t(0), ("or", 0.5, [("and", 2.0, [t(1), t(2)]), t(3)])], max_slop, in_order), 4),
                          (("and", 1.0, [t(0), ("and", 0.7, [t(1), ("or", 1.5, [t(2), t(3)])]), t(4)], max_slop, in_order), 5)):
        assert nested_case(rng, tree, n_lists, True, scorers=["BM25STD", "TFIDF", "DISMAX"]) >= 0


def test_two_level_trees_through_the_nodes_entry_point():
    """root -> children -> terms through RSGPU_HybridTreeNodesQuery: the two-level instantiation (ScoreParams::n_nodes = 0)"""
    rng = np.random.default_rng(77)
    tree = ("and", 1.0, [("or", 0.5, [t(0), t(1)]), t(2), ("and", 2.0, [t(3), t(4)])])
    assert nested_case(rng, tree, 5, False, deep=False) > 0
    assert nested_case(rng, tree, 5, True, deep=False) > 0          # (the per-hit slop from the term offsets: the two-level form has it)
    assert nested_case(rng, ("and", 1.0, [t(1), t(0)]), 2, True, deep=False) > 0


DECLINED = [
    ("ten levels", ("and", 1.0, [t(0), ("and", 1.0, [t(1), ("and", 1.0, [("and", 1.0, [("and", 1.0, [("and", 1.0, [("and", 1.0, [("and", 1.0, [("and", 1.0, [
        ("or", 1.0, [t(2), t(3)])])])])])])]), t(4)])])]), 5),
    ("a root union", ("or", 1.0, [t(0), ("and", 1.0, [t(1), ("or", 1.0, [t(2), t(3)])])]), 4),
    ("no term and no union of terms to drive", ("and", 1.0, [("or", 1.0, [("and", 1.0, [t(0), t(1)]), t(2)]), ("or", 1.0, [("and", 1.0, [t(3), t(4)]), t(5)])]), 6),
    ("a nested window", ("and", 1.0, [t(0), ("or", 1.0, [("and", 1.0, [t(1), t(2)], 3, False), t(3)])]), 4),
]


@pytest.mark.parametrize("name,tree,n_lists", DECLINED)
def test_shapes_the_tile_kernel_declines_are_answered_stage_by_stage(name, tree, n_lists):
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 10000)
    with_offsets = name == "a nested window"
    scorers = ["BM25STD", "DISMAX"] if with_offsets else ["BM25STD", "TFIDF", "DISMAX"]
    if with_offsets:
        # DeepOracle applies the nested node's window itself
        assert nested_case(rng, tree, n_lists, True, want_path=0, scorers=scorers) >= 0
    else:
        assert nested_case(rng, tree, n_lists, False, want_path=0, scorers=scorers) > 0


@pytest.mark.parametrize("with_offsets", [False, True])
@pytest.mark.parametrize("which", [0, 1, 5, 8, 11, 12, 14])
def test_hits_out_of_a_nested_tree_is_the_staged_hit_list(which, with_offsets):
    """round 6: on the tile path (staged until then) -- doc ids, per-leaf frequencies (0 for a leaf outside the result), term
    records and leaf order of RSGPU_EvalTreeNodes' list"""
    rng = np.random.default_rng(5 + which)
    name, tree, n_lists = NESTED[which]
    assert nested_case(rng, tree, n_lists, with_offsets, scorers=["BM25STD", "TFIDF"], want_hits=True) > 0


def test_nested_tree_over_many_tiles_and_a_mutated_index():
    """lists of 10^5 entries (the driver spans ~60 tiles), 64-bit doc ids, BM25STD.NORM, and a vector index whose labels sit in the
    device table (deletes + re-adds): the tile kernel against the staged form bit for bit, the hit count against numpy"""
    rng = np.random.default_rng(23)
    first = (1 << 40) + 7
    n_docs = 400_000
    dfs = [0.15, 0.3, 0.4, 0.35, 0.25]
    docs, g = [], []
    for df in dfs:
        d = np.flatnonzero(rng.random(n_docs) < df).astype(np.uint64) + first
        ii = O.InvertedIndex(O.C_FREQS_ONLY)
        ii.add_many(d, np.minimum(1 + rng.geometric(0.5, d.size), 255).astype(np.uint32))
        docs.append(d)
        g.append(S.Postings.from_flat(ii.flatten()))
    tree = ("and", 1.0, [t(0), ("or", 0.5, [("and", 2.0, [t(1), t(2)]), ("and", 1.5, [t(3), t(4)])])])
    want = np.intersect1d(docs[0], np.union1d(np.intersect1d(docs[1], docs[2]), np.intersect1d(docs[3], docs[4])))
    arrays = ((50 + rng.poisson(150, n_docs + 1)).astype(np.uint32), rng.choice([1.0, 0.5, 0.25], n_docs + 1).astype(np.float32),
              rng.integers(1, 50, n_docs + 1).astype(np.uint32))
    table = S.DocTable(*arrays, first_doc_id=first - 1)
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 32, V.VecSimMetric_L2)
    idx.add_philox_rows(13, 0, 100_000, first)
    for lab in rng.choice(100_000, 500, replace=False).tolist():
        assert idx.delete_vector(first + lab) == 1
    assert idx.label_table() == 1
    q = O.philox_rows(13, 1 << 40, 1, 32)[0]
    idf = [S.calculate_idf(n_docs, d.size) for d in docs]
    bidf = [S.calculate_idf_bm25(n_docs, d.size) for d in docs]
    for scorer in ("BM25STD", "BM25STD.NORM", "DISMAX"):
        a, b, _, _ = general_and_staged(lambda: S.HybridNodesQuery(tree, g, table=table, scorer=scorer, idf=idf, bm25_idf=bidf, weight=[1.0] * 5,
                                                                   num_docs=n_docs, avg_doc_len=200.0, top_n=10, index=idx, q=q, k=10))
        assert a["n_hits"] == want.size and len(a["top"][0]) == 10 and len(a["knn"][0]) == 10
    idx.free()


def test_malformed_node_arrays_are_refused_with_a_message():
    rng = np.random.default_rng(1)
    built = [rand_list(rng, O.C_FREQS_ONLY, 500, 2500, False) for _ in range(3)]
    g = [S.Postings.from_flat(x[0].flatten()) for x in built]
    table = table_for(rng, 2500)
    ones = [1.0] * 3
    for tree in (("and", 1.0, [t(0), t(0)]),                      # a list twice
                 ("and", 1.0, [t(0), t(7)])):                     # a list that is not there
        hq = S.HybridNodesQuery(tree, g, table=table, scorer="BM25STD", idf=ones, bm25_idf=ones, weight=ones, num_docs=2500, avg_doc_len=150.0,
                                top_n=10)
        with pytest.raises(RuntimeError):
            hq.run()
    # a list no node names: RSGPU_EvalTreeNodes ignores it, so does the query (staged)
    hq = S.HybridNodesQuery(("and", 1.0, [t(0), ("or", 1.0, [("and", 1.0, [t(1)])])]), g, table=table, scorer="BM25STD", idf=ones, bm25_idf=ones,
                            weight=ones, num_docs=2500, avg_doc_len=150.0, top_n=10)
    hq.run()
    assert S.hybrid_path() == 0 and hq.results()["n_hits"] > 0
    knob("hybrid_tree_tiles", 1)


# ---- NOT nodes under the root, the root's own window ---------------------------------------------------------------------------------
def test_not_nodes_next_to_nested_children_against_the_oracle():
    """`a ((b c)|d) -e -(f|g)`: the excluded terms veto a candidate in the tile kernel; the result is the positive children's plus
    one virtual child of frequency 0 per NOT (not.rs:106-118), which adds nothing to any sum but counts as a child of the
    intersection (the offset-less slop is children - 1).  No staged twin (RSGPU_EvalTreeNodes has no NOT node): the CPU oracle
    directly -- DeepOracle over the positive subtree, set difference, its scorers over Intersection{..., Virtual, Virtual}."""
    rng = np.random.default_rng(31)
    n_lists = 7
    built = [rand_list(rng, O.C_FREQS_ONLY, int(rng.integers(900, 2200)), 2500, False) for _ in range(n_lists)]
    recs, sizes = [x[1] for x in built], [x[0].unique_docs for x in built]
    g = [S.Postings.from_flat(x[0].flatten()) for x in built]
    positive = ("and", 1.0, [t(0), ("or", 0.5, [("and", 2.0, [t(1), t(2)]), t(3)])])
    nots = [(1.0, [4]), (2.0, [5, 6])]
    tree = ("and", 1.0, positive[2] + [("not", wt, [t(i) for i in ix]) for wt, ix in nots])
    n_docs = 2500
    doc_len = rng.integers(5, 200, n_docs + 1).astype(np.uint32)
    doc_score = rng.choice([1.0, 0.5], n_docs + 1).astype(np.float32)
    max_freq = rng.integers(1, 40, n_docs + 1).astype(np.uint32)
    table = S.DocTable(doc_len, doc_score, max_freq)
    idf = [S.calculate_idf(n_docs, s) for s in sizes]
    bidf = [S.calculate_idf_bm25(n_docs, s) for s in sizes]
    w = [float(x) for x in rng.choice([1.0, 0.5, 2.0], n_lists)]
    avg = float(doc_len[1:].mean())
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 24, V.VecSimMetric_L2)
    idx.add_philox_rows(11, 0, 1200, 100)
    q = O.philox_rows(11, 1 << 40, 1, 24)[0]
    ot = DeepOracle(positive, recs, sizes)
    gone = set().union(*[set(recs[i]) for _, ix in nots for i in ix])
    docs = [d for d in ot.docs if d not in gone]
    assert 0 < len(docs) < len(ot.docs)
    for scorer in SCORERS:
        hq = S.HybridNodesQuery(tree, g, table=table, scorer=scorer, idf=idf, bm25_idf=bidf, weight=w, num_docs=n_docs, avg_doc_len=avg,
                                top_n=10, index=idx, q=q, k=10, root_weight=1.5)
        hq.run()
        assert S.hybrid_path() == 2
        a = hq.results()
        assert a["n_hits"] == len(docs), (scorer, a["n_hits"], len(docs))
        scored = []
        for d in docs:
            node = O.intersection(ot.node(ot.tree, d, idf, bidf, w).kids + [O.Node(O.R_VIRTUAL, wt, 0) for wt, _ in nots])
            node.c.weight = 1.5
            scored.append((O.score(scorer, node, float(doc_score[d]), int(max_freq[d]), int(doc_len[d]), n_docs, avg), d))
        scored.sort(key=lambda x: (-x[0], x[1]))
        assert a["top"][0].tolist() == [d for _, d in scored[:10]], (scorer, a["top"][0], scored[:10])
        if scorer == "BM25STD.TANH":
            assert a["top"][1] == pytest.approx([x for x, _ in scored[:10]], rel=1e-12)
        else:
            assert a["top"][1].tolist() == [x for x, _ in scored[:10]], scorer
    cand = np.asarray([d for d in docs if 100 <= d < 1300], np.int64)
    o = O.FlatIndex(O.F32, 24, O.L2)
    o.add_bulk(O.philox_rows(11, 0, 1200, 24)[cand - 100], 1)
    li, _ = o.topk(q, 10)
    assert a["knn"][0].tolist() == cand[li.astype(np.int64) - 1].tolist()
    # with the hit list (round 6; refused through this entry point until then): the positive children's documents, in order
    hq = S.HybridNodesQuery(tree, g, table=table, scorer="BM25STD", idf=idf, bm25_idf=bidf, weight=w, num_docs=n_docs, avg_doc_len=avg,
                            top_n=10, want_hits=True)
    hq.run()
    assert S.hybrid_path() == 2 and hq.take_hits().read()[0].tolist() == docs
    # where the tile kernel cannot run, such a query is refused with a message (a NOT below a union: malformed)
    hq = S.HybridNodesQuery(("and", 1.0, [t(0), ("or", 1.0, [t(1), ("not", 1.0, [t(2)])])]), g[:3], table=table, scorer="BM25STD", idf=idf,
                            bm25_idf=bidf, weight=w, num_docs=n_docs, avg_doc_len=avg, top_n=10)
    with pytest.raises(RuntimeError):
        hq.run()
    idx.free()


@pytest.mark.parametrize("max_slop,in_order", [(4, False), (None, True), (2, True)])
def test_two_level_trees_with_a_window_and_a_not_node_match_the_tree_query(max_slop, in_order):
    """root -> children -> terms through the nodes entry point is the two-level tree query: the root's max_slop / in_order, a
    slop-dependent scorer over the term offsets, a NOT node -- the same answers as RSGPU_HybridTreeQuery, bit for bit (which
    tests/test_gpu_hybrid_general.py holds to the oracle)"""
    rng = np.random.default_rng(7 + (max_slop or 0) + int(in_order))
    built = [rand_list(rng, O.C_FULL, int(rng.integers(1200, 2200)), 2500, True) for _ in range(5)]
    sizes = [x[0].unique_docs for x in built]
    g = [S.Postings.from_flat(x[0].flatten()) for x in built]
    table = table_for(rng, 2500)
    idf = [S.calculate_idf(2500, s) for s in sizes]
    bidf = [S.calculate_idf_bm25(2500, s) for s in sizes]
    w = [1.0, 0.5, 2.0, 1.5, 0.0]
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 24, V.VecSimMetric_L2)
    idx.add_philox_rows(11, 0, 1200, 100)
    q = O.philox_rows(11, 1 << 40, 1, 24)[0]
    tree = ("and", 1.0, [t(0), ("or", 0.5, [t(1), t(2)]), t(3), ("not", 1.0, [t(4)])], max_slop, in_order)
    groups = [(S.OP_TERM, 1.0, g[:1]), (S.OP_UNION, 0.5, g[1:3]), (S.OP_TERM, 1.0, g[3:4]), (S.OP_NOT, 1.0, g[4:])]
    total = 0
    for scorer in ("BM25STD", "TFIDF.DOCNORM", "BM25"):
        kw = dict(table=table, scorer=scorer, idf=idf, bm25_idf=bidf, weight=w, num_docs=2500, avg_doc_len=150.0, top_n=10, index=idx, q=q, k=10,
                  root_weight=1.5)
        hn = S.HybridNodesQuery(tree, g, **kw)
        hn.run()
        assert S.hybrid_path() == 2
        a = hn.results()
        ht = S.HybridTreeQuery(S.OP_INTERSECT, groups, max_slop=max_slop, in_order=in_order, **kw)
        ht.run()
        assert S.hybrid_path() == 2
        b = ht.results()
        assert a["n_hits"] == b["n_hits"]
        total += a["n_hits"]
        for key in ("top", "knn"):
            assert a[key][0].tolist() == b[key][0].tolist() and a[key][1].tolist() == b[key][1].tolist(), (scorer, key)
    assert total > 0 or max_slop == 2
    idx.free()
