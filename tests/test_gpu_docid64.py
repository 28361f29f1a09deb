"""64-bit doc ids (t_docId) through the search seam: lists whose ids lie above 2^32 keep 32-bit offsets from a per-list
base on the device and are re-based onto one frame per query.  Everything -- decode, intersection (plain, with
proximity), union, two-level trees, NOT over a universe, scorers over a doc-table window, top-N, the ad-hoc KNN step
and the fused hybrid query -- must equal the CPU oracle (which is 64-bit throughout) on the same lists."""
import numpy as np
import pytest

import oracle as O
from redisearch_amd import search as S
from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu
BIG = (1 << 40) + 12345            # far above 2^32


def make_list(rng, codec, n_docs, lo, span):
    docs = (np.unique(rng.integers(0, span, n_docs)).astype(np.uint64) + np.uint64(lo))
    freqs = rng.integers(1, 200, docs.size).astype(np.uint32)
    ii = O.InvertedIndex(codec)
    if codec in (O.C_FULL, O.C_FIELDS_OFFSETS, O.C_OFFSETS_ONLY, O.C_FREQS_OFFSETS, O.C_FREQS_FIELDS, O.C_FIELDS_ONLY):
        for d, f in zip(docs.tolist(), freqs.tolist()):
            pos = np.cumsum(rng.integers(1, 6, int(rng.integers(1, 5))))
            ii.add(d, f, int(rng.integers(1, 2 ** 32 - 1)), b"".join(O.varint_encode(int(x)) for x in np.diff(pos, prepend=0)))
    else:
        ii.add_many(docs, freqs)
    return ii


def up(ii):
    return S.Postings.from_flat(ii.flatten())


@pytest.mark.parametrize("codec", range(9))
def test_decode_above_2_32(codec):
    rng = np.random.default_rng(codec)
    ii = make_list(rng, codec, 4000, BIG, 3_000_000)
    gi, gf, gm = up(ii).decode()
    oi, of, om = ii.decode_all()
    assert gi.dtype == np.uint64 and gi.tolist() == oi.tolist() and int(gi[0]) >= BIG
    assert gf.tolist() == of.tolist() and gm.tolist() == om.tolist()


@pytest.mark.parametrize("codec", [O.C_FREQS_ONLY, O.C_DOCIDS_ONLY, O.C_RAW_DOCIDS, O.C_FULL])
@pytest.mark.parametrize("nl", [2, 3, 5])
def test_intersection_and_union_lists_with_different_bases(codec, nl):
    # every list starts somewhere else (its own base on the device); the query re-bases them onto one frame
    rng = np.random.default_rng(10 * nl + codec)
    lists = [make_list(rng, codec, int(rng.integers(3000, 30000)), BIG + int(rng.integers(0, 5000)), 60_000) for _ in range(nl)]
    ps = [up(l) for l in lists]
    oi, of, _ = O.intersect(lists)
    gi, gf = S.intersect(ps).read()
    assert len(oi) > 0 and gi.tolist() == oi.tolist() and gf.tolist() == of.tolist()
    ui, uf = O.union_lists(lists)[:2]
    hi, hf = S.union(ps).read()
    assert hi.tolist() == ui.tolist() and hf.tolist() == uf.tolist()


def test_a_list_below_2_32_meets_a_list_that_crosses_it():
    # list A ends just below 2^32 (base 0 on the device), list B starts below and ends above (base = its first id):
    # the shared frame starts at the smaller first id, A's offset into it is negative
    rng = np.random.default_rng(3)
    lo = (1 << 32) - 40_000
    a = make_list(rng, O.C_FREQS_ONLY, 20_000, lo, 39_000)          # entirely below 2^32
    b = make_list(rng, O.C_FREQS_ONLY, 30_000, lo + 10_000, 80_000)  # crosses 2^32
    assert int(a.decode_all()[0][-1]) < (1 << 32) < int(b.decode_all()[0][-1])
    oi, of, _ = O.intersect([a, b])
    gi, gf = S.intersect([up(a), up(b)]).read()
    assert len(oi) > 100 and gi.tolist() == oi.tolist() and gf.tolist() == of.tolist()
    ui, uf = O.union_lists([b, a])[:2]
    hi, hf = S.union([up(b), up(a)]).read()
    assert hi.tolist() == ui.tolist() and hf.tolist() == uf.tolist()


def test_spans_of_2_32_or_more_are_refused_loudly():
    ii = O.InvertedIndex(O.C_DOCIDS_ONLY)
    for d in (5, (1 << 32) + 100):                     # one list spanning more than 2^32 ids: two blocks
        ii.add(d, 1, 1, b"")
    fl = ii.flatten()
    with pytest.raises(RuntimeError, match="2\\^32"):
        S.Postings.from_flat(fl)
    lo, hi = O.InvertedIndex(O.C_DOCIDS_ONLY), O.InvertedIndex(O.C_DOCIDS_ONLY)
    lo.add_many(np.arange(1, 100, dtype=np.uint64), np.ones(99, np.uint32))
    hi.add_many(np.arange(BIG, BIG + 100, dtype=np.uint64), np.ones(100, np.uint32))
    with pytest.raises(RuntimeError, match="2\\^32"):
        S.intersect([up(lo), up(hi)])


@pytest.mark.parametrize("scorer", ["TFIDF", "BM25", "BM25STD", "DISMAX", "DOCSCORE"])
def test_scorers_topn_over_a_doc_table_window(scorer):
    rng = np.random.default_rng(77)
    N, first = 150_000, BIG - 1000                      # the table window starts below the lists' ids
    lists = [make_list(rng, O.C_FREQS_ONLY, n, BIG + off, N) for n, off in ((40_000, 0), (20_000, 17), (90_000, 5))]
    doc_len = (50 + rng.poisson(150, N + 2000)).astype(np.uint32)
    doc_score = rng.uniform(0.2, 1.0, N + 2000).astype(np.float32)
    max_freq = np.maximum(doc_len // 7, 1).astype(np.uint32)
    table = S.DocTable(doc_len, doc_score, max_freq, first_doc_id=first)
    oi, of, _ = O.intersect(lists)
    assert len(oi) > 300
    h = S.intersect([up(l) for l in lists])
    idf = [S.calculate_idf(N, l.unique_docs) for l in lists]
    bidf = [S.calculate_idf_bm25(N, l.unique_docs) for l in lists]
    w = [1.0, 0.5, 2.0]
    avg = float(doc_len.mean())
    gs = h.score(table, scorer, idf, bidf, w, N, avg, root_weight=0.7)
    sel = (oi - np.uint64(first)).astype(np.int64)
    os_ = O.score_flat(scorer, of, doc_len[sel], max_freq[sel], doc_score[sel], idf, bidf, w, 0.7, N, avg)
    assert np.array_equal(gs, os_)
    for n in (1, 10, 1000):
        ti, ts = h.topn(n)
        order = np.lexsort((oi, -os_))[:n]
        assert ti.tolist() == oi[order].tolist() and np.array_equal(ts, os_[order])


def test_proximity_and_tree_above_2_32():
    rng = np.random.default_rng(9)
    lists = [make_list(rng, O.C_FULL, 6000, BIG + 3 * i, 9000) for i in range(4)]
    ps = [up(l) for l in lists]
    for max_slop, in_order in ((0, False), (2, True), (5, False)):
        oi, of = O.intersect_ex(lists[:3], max_slop, in_order)[:2]
        gi, gf = S.intersect(ps[:3], max_slop=max_slop, in_order=in_order).read()
        assert gi.tolist() == oi.tolist() and gf.tolist() == of.tolist()
    # (l0 | l1) (l2 | l3): ids of the tree = intersection of the two unions
    t = S.TreeHits(S.OP_INTERSECT, [(S.OP_UNION, 1.0, ps[:2]), (S.OP_UNION, 1.0, ps[2:])])
    u01 = set(O.union_lists(lists[:2])[0].tolist())
    u23 = set(O.union_lists(lists[2:])[0].tolist())
    assert t.read()[0].tolist() == sorted(u01 & u23)


def test_not_over_a_universe_above_2_32():
    rng = np.random.default_rng(4)
    universe = O.InvertedIndex(O.C_DOCIDS_ONLY)
    ids = np.arange(BIG, BIG + 50_000, dtype=np.uint64)
    universe.add_many(ids, np.ones(ids.size, np.uint32))
    child = O.InvertedIndex(O.C_DOCIDS_ONLY)
    held = np.sort(rng.choice(ids[100:], 20_000, replace=False))
    child.add_many(held, np.ones(held.size, np.uint32))
    # (the child's base on the device is its own first id, not the universe's: the kernel translates)
    h = S.negate(up(child), int(ids[-1]), up(universe))
    assert h.read()[0].tolist() == sorted(set(ids.tolist()) - set(held.tolist()))


def test_knn_rerank_and_fused_hybrid_query_above_2_32():
    rng = np.random.default_rng(49)
    n_docs, n_vec, dim, k = 200_000, 30_000, 64, 10
    lists = [make_list(rng, O.C_FREQS_ONLY, n, BIG + 1, n_docs) for n in (60_000, 30_000)]
    data = rng.uniform(-1, 1, (n_vec, dim)).astype(np.float32)
    g = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2)
    g.add_bulk(data, first_label=BIG + 1)               # identity labels from BIG+1: docs beyond have no vector
    o = O.FlatIndex(O.F32, dim, O.L2)
    o.add_bulk(data)
    q = rng.uniform(-1, 1, dim).astype(np.float32)
    oi, of, _ = O.intersect(lists)
    ps = [up(l) for l in lists]
    gi, gd = S.intersect(ps).knn_rerank(g, q, k)
    nq = o.normalized_query(q)
    want = sorted((o.distance_from(int(i) - BIG, nq), int(i)) for i in oi if int(i) - BIG <= n_vec)[:k]
    assert gi.tolist() == [i for _, i in want]
    assert np.allclose(gd, [d for d, _ in want], rtol=1e-5, atol=1e-4)
    doc_len = (50 + rng.poisson(150, n_docs + 10)).astype(np.uint32)
    doc_score = rng.uniform(0.2, 1.0, n_docs + 10).astype(np.float32)
    table = S.DocTable(doc_len, doc_score, None, first_doc_id=BIG)
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists]
    bidf = [S.calculate_idf_bm25(n_docs, l.unique_docs) for l in lists]
    r = S.hybrid_query(ps, table, "BM25STD", idf, bidf, [1.0, 1.0], n_docs, float(doc_len.mean()), top_n=10, index=g, q=q, k=k)
    assert r["n_hits"] == len(oi) and r["knn"][0].tolist() == gi.tolist() and np.array_equal(r["knn"][1], gd)
    sel = (oi - np.uint64(BIG)).astype(np.int64)
    os_ = O.score_flat("BM25STD", of, doc_len[sel], np.ones(sel.size, np.uint32), doc_score[sel], idf, bidf, [1.0, 1.0], 1.0,
                       n_docs, float(doc_len.mean()))
    order = np.lexsort((oi, -os_))[:10]
    assert r["top"][0].tolist() == oi[order].tolist() and np.array_equal(r["top"][1], os_[order])
