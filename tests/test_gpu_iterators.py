"""GPU: Boundary 3 -- the reference's QueryIterator vtable (src/iterators/iterator_api.h:46-151) served from device hit
lists by redisearch_amd/lib/librsgpu_iterators.so.

The iterators are driven the way the module drives them (oracle/ext_harness.c plays the module: it supplies the
RSIndexResult constructors an iterator written in C builds `current` with, reads / skips / rewinds through the vtable and
scores the results with a scorer extension) and held to
  * the CPU oracle's intersection / union / NOT over the same posting lists -- doc ids, the aggregate's frequency and
    field mask, every child's record (frequency, field mask, term positions) in the reference's child order;
  * the reference's own known answers: tests/cpptests/test_cpp_index.cpp:542-601 (read / skip-to sequence of a two-list
    intersection), rqe_iterators/tests/integration/intersection.rs fixtures (tests/intersection_cases.py);
  * the reference's compiled scorers (oracle/_ref/libref_default_ext.so, src/ext/default.c) applied to the iterator's
    results one by one == RSGPU_Hits_Score over the same hit list in one batch.
"""
import zlib

import numpy as np
import pytest

import oracle as O
import oracle.ext as X
from redisearch_amd import search as S
from tests import intersection_cases as IC
from tests.test_gpu_proximity import gpu, rand_list

pytestmark = pytest.mark.gpu

ALL = (1 << 128) - 1
HAS_FREQ = {O.C_FULL, O.C_FREQS_FIELDS, O.C_FREQS_ONLY, O.C_FREQS_OFFSETS, O.C_FULL_WIDE, O.C_FREQS_FIELDS_WIDE}
HAS_MASK = {O.C_FULL, O.C_FREQS_FIELDS, O.C_FIELDS_ONLY, O.C_FIELDS_OFFSETS, O.C_FULL_WIDE, O.C_FREQS_FIELDS_WIDE,
            O.C_FIELDS_ONLY_WIDE, O.C_FIELDS_OFFSETS_WIDE}
WIDE = {O.C_FULL_WIDE, O.C_FREQS_FIELDS_WIDE, O.C_FIELDS_ONLY_WIDE, O.C_FIELDS_OFFSETS_WIDE}


@pytest.fixture(scope="module", autouse=True)
def _bind():
    S.load_iterators(X.handle())   # the harness is "the module": its constructors build the iterators' results


def records(ii):
    """doc -> (freq, mask128, positions) as the term's own reader yields them (reference defaults for what the codec
    does not store: frequency 1, RS_FIELDMASK_ALL -- index_result/src/core/mod.rs:192-197, term.rs:93-97)"""
    ids, fr, mk = ii.decode_all()
    masks = ii.decode_masks128() if ii.codec in WIDE else [int(m) for m in mk.tolist()]
    out, rd = {}, ii.reader()
    for i, doc in enumerate(ids.tolist()):
        got = rd.next()
        assert got is not None and got[0] == doc
        out[doc] = (int(fr[i]) if ii.codec in HAS_FREQ else 1, masks[i] if ii.codec in HAS_MASK else ALL, rd.offsets())
    return out


def check_drain(it, lists, order, want_ids, cap=None, union=False):
    """Drains the iterator and compares every hit with the oracle's records of the lists (children in `order`)."""
    recs = [records(l) for l in lists]
    d = X.iter_drain(it, len(want_ids) + 8 if cap is None else cap, len(lists))
    assert d["n"] == len(want_ids)
    assert d["ids"].tolist() == list(want_ids)
    for h, doc in enumerate(want_ids):
        kids = [c for c in order if doc in recs[c]] if union else list(order)
        assert d["n_children"][h] == len(kids)
        f, m = 0, 0
        for slot, c in enumerate(kids):
            fr, mk, pos = recs[c][doc]
            assert d["c_same_doc"][slot][h] == 1
            assert d["c_freq"][slot][h] == fr, (doc, c)
            assert (int(d["c_mask_lo"][slot][h]) | (int(d["c_mask_hi"][slot][h]) << 64)) == mk, (doc, c)
            assert d["c_npos"][slot][h] == len(pos) and d["c_hash"][slot][h] == X.positions_hash(pos), (doc, c)
            f += fr
            m |= mk
        assert d["freq"][h] == f and d["mask"][h] == m, doc


def size_order(lists, in_order=False):
    # children iterate by ascending estimate, stable (intersection.rs:94-119); in_order keeps the query's order
    return list(range(len(lists))) if in_order else sorted(range(len(lists)), key=lambda i: lists[i].unique_docs)


@pytest.mark.parametrize("codec", [O.C_FULL, O.C_FREQS_FIELDS, O.C_FREQS_ONLY, O.C_FIELDS_ONLY, O.C_FIELDS_OFFSETS,
                                   O.C_OFFSETS_ONLY, O.C_FREQS_OFFSETS, O.C_DOCIDS_ONLY, O.C_RAW_DOCIDS, O.C_FULL_WIDE,
                                   O.C_FIELDS_OFFSETS_WIDE])
@pytest.mark.parametrize("block", [7, 65536])
def test_intersection_iterator_yields_the_reference_result_tree(codec, block):
    rng = np.random.default_rng(100 + codec)
    lists = [rand_list(rng, codec, n, 900, wide=codec in WIDE) for n in (700, 400, 650)]
    g = [gpu(l) for l in lists]
    S.load_iterators().RSGPU_Iterators_SetBlock(block)
    try:
        it = S.new_iterator("and", g, weight=2.5)
    finally:
        S.load_iterators().RSGPU_Iterators_SetBlock(65536)
    try:
        st = S.QueryIteratorStruct.from_address(it)
        assert st.type == 7 and not st.atEOF and st.lastDocId == 0 and not st.current      # IteratorType_Intersect
        want, _, _ = O.intersect(lists)
        assert len(want) > 20
        (est,) = X.iter_script(it, [(X.OP_ESTIMATE, 0)])
        assert est[0] == min(l.unique_docs for l in lists)                                 # intersection.rs:144-146
        check_drain(it, lists, size_order(lists), want.tolist())
        # past the end: EOF again, lastDocId stays on the last result, current stays NULL (iterator_api.h:96-99)
        assert X.iter_script(it, [(X.OP_READ, 0)]) == [(S.IT_EOF, int(want[-1]), True, 0)]
        # Rewind resets atEOF / lastDocId (iterator_api.h:139) and the second pass is the first
        assert X.iter_script(it, [(X.OP_REWIND, 0)]) == [(0, 0, False, 0)]
        check_drain(it, lists, size_order(lists), want.tolist())
    finally:
        X.iter_free(it)


@pytest.mark.parametrize("max_slop,in_order", [(0, False), (2, True), (-1, True), (5, False)])
def test_proximity_iterator(max_slop, in_order):
    rng = np.random.default_rng(7)
    lists = [rand_list(rng, O.C_FULL, n, 500, max_pos=12) for n in (420, 380)]
    g = [gpu(l) for l in lists]
    it = S.new_iterator("and", g, max_slop=max_slop, in_order=in_order)
    try:
        want, _, _ = O.intersect_ex(lists, max_slop=None if max_slop < 0 else max_slop, in_order=in_order)
        assert 0 < len(want) < min(l.unique_docs for l in lists)
        check_drain(it, lists, size_order(lists, in_order), want.tolist())
    finally:
        X.iter_free(it)


def test_cpp_index_read_skip_kat():
    """tests/cpptests/test_cpp_index.cpp:542-601: two lists over 100 000 entries with doc ids 4(i+1) and 2(i+1), frequency 1:
    50 000 hits (2c+2)*2 with total frequency 2; SkipTo(8) -> OK, Read -> 12, SkipTo(200000) -> OK, then EOF."""
    a, b = O.InvertedIndex(O.C_FULL), O.InvertedIndex(O.C_FULL)
    n = 100000
    a.add_many(np.arange(1, n + 1, dtype=np.uint64) * 4, np.ones(n, np.uint32))
    b.add_many(np.arange(1, n + 1, dtype=np.uint64) * 2, np.ones(n, np.uint32))
    ga, gb = gpu(a), gpu(b)
    it = S.new_iterator("and", [ga, gb])
    try:
        d = X.iter_drain(it, 50001, 2)
        assert d["n"] == 50000
        assert d["ids"].tolist() == [(2 * c + 2) * 2 for c in range(50000)]
        assert set(d["freq"].tolist()) == {2} and set(d["n_children"].tolist()) == {2}
        got = X.iter_script(it, [(X.OP_REWIND, 0), (X.OP_SKIP, 8), (X.OP_READ, 0), (X.OP_SKIP, 200000), (X.OP_READ, 0),
                                 (X.OP_SKIP, 200004)])
        assert got[1] == (S.IT_OK, 8, False, 8)
        assert got[2] == (S.IT_OK, 12, False, 12)
        assert got[3] == (S.IT_OK, 200000, False, 200000)
        assert got[4] == (S.IT_EOF, 200000, True, 0)
        assert got[5][0] == S.IT_EOF and got[5][1] == 200000
    finally:
        X.iter_free(it)


def model_script(ids, ops):
    """The observable behaviour of a reference iterator over the result set `ids` (iterator_api.h:86-120,
    intersection.rs:428-530): Read yields the next result; SkipTo(d) the first result >= d (OK when equal, NOTFOUND when
    greater) and consumes it; EOF sets atEOF, clears current and leaves lastDocId; Rewind starts over."""
    pos, last, eof, out = 0, 0, False, []
    for op, arg in ops:
        if op == X.OP_REWIND:
            pos, last, eof = 0, 0, False
            out.append((0, 0, False, 0))
            continue
        if op == X.OP_SKIP and not eof:
            while pos < len(ids) and ids[pos] < arg:
                pos += 1
        if eof or pos >= len(ids):
            eof = True
            out.append((S.IT_EOF, last, True, 0))
            continue
        last = ids[pos]
        status = S.IT_OK if (op == X.OP_READ or last == arg) else S.IT_NOTFOUND
        pos += 1
        out.append((status, last, False, last))
    return out


@pytest.mark.parametrize("num_children", IC.NUM_CHILDREN_CASES)
@pytest.mark.parametrize("case", range(len(IC.RESULT_SET_CASES)))
def test_reference_fixture_read_skip_rewind_scripts(num_children, case):
    """rqe_iterators/tests/integration/intersection.rs read / skip_to / rewind over 2 / 5 / 25 children: every id of the
    expected set and every gap in between is skipped to, from the start and from the middle, with reads in between."""
    result = IC.RESULT_SET_CASES[case]
    kids = IC.create_children(num_children, result)
    lists = [IC.to_index(k) for k in kids]
    g = [gpu(l) for l in lists]
    S.load_iterators().RSGPU_Iterators_SetBlock(4)
    it = S.new_iterator("and", g)
    S.load_iterators().RSGPU_Iterators_SetBlock(65536)
    try:
        ops = []
        targets = sorted({t for r in result for t in (r - 1, r, r + 1) if t > 0} | set(range(1, 60)) | {max(result) + 2})
        for d in targets:     # skip_to every expected id, its neighbours and a dense prefix (intersection.rs skip_to tests)
            ops += [(X.OP_REWIND, 0), (X.OP_SKIP, d), (X.OP_READ, 0)]
        ops += [(X.OP_REWIND, 0)] + [(X.OP_READ, 0)] * (len(result) + 2)
        ops += [(X.OP_REWIND, 0), (X.OP_READ, 0), (X.OP_SKIP, result[2]), (X.OP_SKIP, result[2] + 1), (X.OP_READ, 0),
                (X.OP_SKIP, result[-1]), (X.OP_READ, 0), (X.OP_READ, 0)]
        assert X.iter_script(it, ops) == model_script(result, ops)
    finally:
        X.iter_free(it)


@pytest.mark.parametrize("seed", range(4))
def test_random_scripts(seed):
    rng = np.random.default_rng(300 + seed)
    lists = [rand_list(rng, O.C_FREQS_ONLY, n, 5000) for n in (3000, 2500)]
    g = [gpu(l) for l in lists]
    S.load_iterators().RSGPU_Iterators_SetBlock(int(rng.integers(3, 200)))
    it = S.new_iterator("and", g)
    S.load_iterators().RSGPU_Iterators_SetBlock(65536)
    try:
        want = O.intersect(lists)[0].tolist()
        ops, last = [], 0
        for _ in range(600):
            r = rng.random()
            if r < 0.45:
                ops.append((X.OP_READ, 0))
            elif r < 0.95:
                last = last + int(rng.integers(1, 60))      # SkipTo is only legal beyond lastDocId (iterator_api.h:105)
                ops.append((X.OP_SKIP, last))
            else:
                ops.append((X.OP_REWIND, 0))
                last = 0
            exp = model_script(want, ops)
            last = max(last, exp[-1][1]) if ops[-1][0] != X.OP_REWIND else 0
        assert X.iter_script(it, ops) == model_script(want, ops)
    finally:
        X.iter_free(it)


@pytest.mark.parametrize("codec", [O.C_FULL, O.C_FREQS_ONLY, O.C_DOCIDS_ONLY])
def test_union_iterator(codec):
    rng = np.random.default_rng(40 + codec)
    lists = [rand_list(rng, codec, n, 700) for n in (200, 350, 120)]
    g = [gpu(l) for l in lists]
    S.load_iterators().RSGPU_Iterators_SetBlock(50)
    it = S.new_iterator("or", g, weight=0.5)
    S.load_iterators().RSGPU_Iterators_SetBlock(65536)
    try:
        assert S.QueryIteratorStruct.from_address(it).type == 6                                # IteratorType_Union
        want = O.union_lists(lists)[0].tolist()
        (est,) = X.iter_script(it, [(X.OP_ESTIMATE, 0)])
        assert est[0] == sum(l.unique_docs for l in lists)                                     # union_flat.rs:102
        hits_order = list(range(len(lists)))
        check_drain(it, lists, hits_order, want, union=True)
        ops = [(X.OP_REWIND, 0), (X.OP_SKIP, want[5]), (X.OP_SKIP, want[9] + 1), (X.OP_READ, 0), (X.OP_SKIP, want[-1] + 1)]
        assert X.iter_script(it, ops) == model_script(want, ops)
    finally:
        X.iter_free(it)


@pytest.mark.parametrize("with_universe", [False, True])
def test_not_iterator(with_universe):
    rng = np.random.default_rng(5)
    child = rand_list(rng, O.C_DOCIDS_ONLY, 300, 600)
    uni = rand_list(rng, O.C_DOCIDS_ONLY, 450, 640) if with_universe else None
    gc, gu = gpu(child), gpu(uni) if uni is not None else None
    it = S.new_iterator("not", [gc], universe=gu, max_doc_id=620, weight=3.0)
    try:
        assert S.QueryIteratorStruct.from_address(it).type == 8                                # IteratorType_Not
        want = O.not_list(child, 620, uni).tolist()
        (est,) = X.iter_script(it, [(X.OP_ESTIMATE, 0)])
        assert est[0] == 620                                                                   # not.rs:301-303
        d = X.iter_drain(it, len(want) + 4, 1)
        assert d["ids"].tolist() == want and d["n"] == len(want)
        assert set(d["n_children"].tolist()) == {0} and set(d["mask"]) == {ALL} and set(d["freq"].tolist()) == {0}   # not.rs:112-115
        ops = [(X.OP_REWIND, 0), (X.OP_SKIP, want[3]), (X.OP_SKIP, want[10] + 1), (X.OP_READ, 0), (X.OP_SKIP, 10 ** 9)]
        assert X.iter_script(it, ops) == model_script(want, ops)
    finally:
        X.iter_free(it)


def test_empty_intersection_is_at_eof_from_the_first_read():
    a, b = O.InvertedIndex(O.C_FREQS_ONLY), O.InvertedIndex(O.C_FREQS_ONLY)
    a.add_many(np.arange(1, 50, dtype=np.uint64) * 2)
    b.add_many(np.arange(1, 50, dtype=np.uint64) * 2 + 1)
    it = S.new_iterator("and", [gpu(a), gpu(b)])
    try:
        assert X.iter_script(it, [(X.OP_READ, 0), (X.OP_REWIND, 0), (X.OP_SKIP, 3)]) == [(S.IT_EOF, 0, True, 0), (0, 0, False, 0),
                                                                                     (S.IT_EOF, 0, True, 0)]
    finally:
        X.iter_free(it)


@pytest.mark.parametrize("scorer", ["BM25STD", "TFIDF", "TFIDF.DOCNORM", "BM25", "DISMAX", "DOCSCORE", "BM25STD.TANH"])
def test_results_scored_one_by_one_equal_the_batched_scorer(scorer):
    """rpscoreNext over the iterator (the reference's compiled default.c scorers where oracle/_ref has them, else the
    plugin) == RSGPU_Hits_Score over the iterator's own hit list: same ids, same fp64 scores."""
    rng = np.random.default_rng(11)
    lists = [rand_list(rng, O.C_FULL, n, 1500, max_pos=30) for n in (1200, 1000)]
    g = [gpu(l) for l in lists]
    n_docs = 1500
    doc_len = rng.integers(5, 300, n_docs + 1).astype(np.uint32)
    doc_score = rng.random(n_docs + 1).astype(np.float32)
    max_freq = rng.integers(1, 60, n_docs + 1).astype(np.uint32)
    avg = float(doc_len.mean())
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists]
    bidf = [S.calculate_idf_bm25(n_docs, l.unique_docs) for l in lists]
    w = [1.0, 1.75]
    host = X.Host()
    if X.have_ref():
        host.load_ref()
    else:
        host.load_plugin()
    terms = [X.new_term(idf[i], bidf[i], "t%d" % i) for i in range(2)]
    it = S.new_iterator("and", g, terms=terms, weights=w, weight=1.25)
    try:
        ids, sc = X.iter_score_all(it, scorer, doc_len, doc_score, max_freq, 2000, num_docs=n_docs, avg_doc_len=avg,
                                   slop=0 if X.have_ref() else 1)
        hp = S.load_iterators().RSGPU_Iterator_Hits(it)
        assert hp
        table = S.DocTable(doc_len, doc_score, max_freq)
        h = S.Hits.__new__(S.Hits)          # a view of the iterator's own hit list (not owned)
        h.lib, h.ptr, h.n_lists, h._lists = S.load(), hp, 2, g
        try:
            gid = h.read()[0]
            gsc = h.score(table, scorer, idf, bidf, w, n_docs, avg, root_weight=1.25)
        finally:
            h.ptr = None
        assert ids.tolist() == gid.tolist() and len(ids) > 100
        if X.have_ref():
            assert np.allclose(sc, gsc, rtol=1e-12, atol=0)
            assert np.array_equal(sc, gsc) or scorer == "BM25STD.TANH"
    finally:
        X.iter_free(it)


# ---- two-level trees: Intersection{Union{..}, Term, ..} behind one iterator ------------------------------------------------
from tests.test_gpu_tree import I as OP_I, T as OP_T, U as OP_U, OracleTree, rand_list as tree_rand_list  # noqa: E402


@pytest.mark.parametrize("name,root,shape,slop,in_order", [
    ("and_of_ors", OP_I, [(OP_U, 1.0, [0, 1, 2]), (OP_U, 0.5, [3, 4])], None, False),          # (a|a'|a'') (b|b')
    ("or_of_ands", OP_U, [(OP_I, 1.0, [0, 1]), (OP_I, 2.0, [2, 3])], None, False),              # (a b) | (c d)
    ("term_and_or", OP_I, [(OP_T, 1.0, [0]), (OP_U, 1.0, [1, 2, 3]), (OP_T, 1.0, [4])], None, False),
    ("or_of_term_and_and", OP_U, [(OP_T, 1.0, [0]), (OP_I, 0.7, [1, 2, 3])], None, False),      # a | (b c d)
    ("phrase_over_a_union", OP_I, [(OP_T, 1.0, [0]), (OP_U, 1.0, [1, 2, 3]), (OP_T, 1.0, [4])], 3, True),
])
def test_tree_iterator_results_score_like_the_batched_tree_scorer(name, root, shape, slop, in_order):
    """The iterator's `current` for a two-level tree, scored one by one by the reference's compiled scorers (which walk
    the nested aggregates and ask IndexResult_MinOffsetDelta for the slop) == RSGPU_Hits_Score over the tree's hit list;
    doc ids, the number of root children per document and the aggregate frequency follow the oracle's set algebra."""
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000 + 17)
    n_lists = sum(len(g[2]) for g in shape)
    built = [tree_rand_list(rng, O.C_FULL, int(rng.integers(300, 1500)), 2500, True) for _ in range(n_lists)]
    lists, recs = [b[0] for b in built], [b[1] for b in built]
    sizes = [l.unique_docs for l in lists]
    g = [gpu(l) for l in lists]
    groups = [(op, w, [g[i] for i in idx]) for op, w, idx in shape]
    ot = OracleTree(root, shape, recs, sizes, slop, in_order)
    assert len(ot.docs) > 5
    n_docs = 2500
    doc_len = rng.integers(5, 200, n_docs + 1).astype(np.uint32)
    doc_score = rng.choice([1.0, 0.5], n_docs + 1).astype(np.float32)
    max_freq = rng.integers(1, 40, n_docs + 1).astype(np.uint32)
    idf = [S.calculate_idf(n_docs, s) for s in sizes]
    bidf = [S.calculate_idf_bm25(n_docs, s) for s in sizes]
    w = [float(x) for x in rng.choice([1.0, 0.5, 2.0], n_lists)]
    avg = float(doc_len[1:].mean())
    host = X.Host()
    if not X.have_ref():
        pytest.skip("needs oracle/_ref/libref_default_ext.so (the reference's scorers and IndexResult_MinOffsetDelta)")
    host.load_ref()
    S.load_iterators().RSGPU_Iterators_SetBlock(37)
    h = S.TreeHits(root, groups, max_slop=slop, in_order=in_order)
    table = S.DocTable(doc_len, doc_score, max_freq)
    try:
        for scorer in ("BM25STD", "TFIDF", "BM25", "DISMAX", "TFIDF.DOCNORM"):
            terms = [X.new_term(idf[i], bidf[i], "t%d" % i) for i in range(n_lists)]
            it = S.new_tree_iterator(root, groups, terms=terms, weights=w, weight=1.5, max_slop=slop, in_order=in_order)
            try:
                (est,) = X.iter_script(it, [(X.OP_ESTIMATE, 0)])
                ge = [min(sizes[i] for i in idx) if op == OP_I else sum(sizes[i] for i in idx) for op, _, idx in shape]
                assert est[0] == (min(ge) if root == OP_I else sum(ge))
                ids, sc = X.iter_score_all(it, scorer, doc_len, doc_score, max_freq, len(ot.docs) + 8, num_docs=n_docs,
                                           avg_doc_len=avg, slop=0)
                assert ids.tolist() == ot.docs
                gs = h.score(table, scorer, idf, bidf, w, n_docs, avg, root_weight=1.5)
                assert np.array_equal(sc, gs), (scorer, np.max(np.abs(sc - gs)))
                if scorer == "BM25STD":
                    X.iter_script(it, [(X.OP_REWIND, 0)])
                    d = X.iter_drain(it, len(ot.docs) + 8, len(shape))
                    for j, doc in enumerate(ot.docs):
                        matched = [gr for gr in ot.groups if doc in gr["docs"]]
                        assert d["n_children"][j] == len(matched)
                        assert d["freq"][j] == sum(recs[i][doc][0] for gr in matched for i in gr["idx"] if doc in recs[i])
            finally:
                X.iter_free(it)
    finally:
        S.load_iterators().RSGPU_Iterators_SetBlock(65536)


# ---- query trees of any depth behind the vtable (RSGPU_NewTreeNodesIterator) --------------------------------------------
@pytest.mark.parametrize("name,tree,n_lists", [
    ("and_or_and", ("and", 1.0, [("or", 0.5, [("and", 2.0, [("t", 0), ("t", 1)]), ("t", 2)]), ("t", 3)]), 4),
    ("or_and_or_and", ("or", 1.0, [("t", 0), ("and", 0.7, [("t", 1), ("or", 1.0, [("t", 2), ("and", 3.0, [("t", 3), ("t", 4)])])])]), 5),
    ("phrase_inside", ("and", 1.0, [("or", 1.0, [("and", 1.5, [("t", 0), ("t", 1)], 4, True), ("t", 2)]), ("t", 3)]), 4),
])
def test_deep_tree_iterator_results_score_like_the_batched_scorer(name, tree, n_lists):
    """`current` of a depth-3 / depth-4 tree -- nested aggregates rebuilt per document, matched children only under a
    union -- scored one by one by the reference's COMPILED scorers (src/ext/default.c walking the nested aggregates,
    IndexResult_MinOffsetDelta for the slop) == RSGPU_Hits_Score's post-order evaluation over the same hit list, bit for
    bit; with the in-place rebuild forced through small record blocks."""
    if not X.have_ref():
        pytest.skip("needs oracle/_ref/libref_default_ext.so (the reference's scorers and IndexResult_MinOffsetDelta)")
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000 + 3)
    built = [tree_rand_list(rng, O.C_FULL, int(rng.integers(900, 1900)), 2500, True) for _ in range(n_lists)]
    lists = [b[0] for b in built]
    sizes = [l.unique_docs for l in lists]
    g = [gpu(l) for l in lists]
    n_docs = 2500
    doc_len = rng.integers(5, 200, n_docs + 1).astype(np.uint32)
    doc_score = rng.choice([1.0, 0.5], n_docs + 1).astype(np.float32)
    max_freq = rng.integers(1, 40, n_docs + 1).astype(np.uint32)
    idf = [S.calculate_idf(n_docs, s) for s in sizes]
    bidf = [S.calculate_idf_bm25(n_docs, s) for s in sizes]
    w = [float(x) for x in rng.choice([1.0, 0.5, 2.0], n_lists)]
    avg = float(doc_len[1:].mean())
    host = X.Host()
    host.load_ref()
    S.load_iterators().RSGPU_Iterators_SetBlock(41)
    h = S.NodeHits(tree, g)
    table = S.DocTable(doc_len, doc_score, max_freq)
    want_ids = h.read()[0].tolist()
    assert len(want_ids) > 5
    try:
        for scorer in ("BM25STD", "TFIDF", "BM25", "DISMAX", "TFIDF.DOCNORM"):
            terms = [X.new_term(idf[i], bidf[i], "t%d" % i) for i in range(n_lists)]
            it = S.new_tree_nodes_iterator(tree, g, terms=terms, weights=w, weight=1.5)
            try:
                ids, sc = X.iter_score_all(it, scorer, doc_len, doc_score, max_freq, len(want_ids) + 8, num_docs=n_docs,
                                           avg_doc_len=avg, slop=0)
                assert ids.tolist() == want_ids
                gs = h.score(table, scorer, idf, bidf, w, n_docs, avg, root_weight=1.5)
                assert np.array_equal(sc, gs), (scorer, np.max(np.abs(sc - gs)))
                # rewind + skip around: the same documents again, whatever the wiring left behind
                X.iter_script(it, [(X.OP_REWIND, 0)])
                some = want_ids[:: max(len(want_ids) // 7, 1)]
                res = X.iter_script(it, [(X.OP_SKIP, d) for d in some])
                assert [r[3] for r in res] == some
            finally:
                X.iter_free(it)
    finally:
        S.load_iterators().RSGPU_Iterators_SetBlock(0)
        h.free()
