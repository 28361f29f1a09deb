"""GPU: term offsets on the device -- max_slop / in_order intersections (RSGPU_IntersectEx), the scorers' slop
(IndexResult_MinOffsetDelta) computed from the offset bytes, and the *Wide codecs -- against the CPU oracle and the
reference's known answers (rqe_iterators/tests/integration/intersection.rs:1164-1370, tests/pytests/test_scorers.py)."""
import math

import numpy as np
import pytest

import oracle as O
from redisearch_amd import search as S

pytestmark = pytest.mark.gpu


def mock(codec, docs, positions):
    ii = O.InvertedIndex(codec)
    for d, p in zip(docs, positions):
        ii.add(d, 1, 1, O.varint_encode(p))
    return ii


def gpu(ii):
    return S.Postings.from_flat(ii.flatten())


@pytest.mark.parametrize("codec", [O.C_FULL, O.C_OFFSETS_ONLY, O.C_FREQS_OFFSETS, O.C_FIELDS_OFFSETS, O.C_FULL_WIDE,
                                   O.C_FIELDS_OFFSETS_WIDE])
def test_slop_and_order_kats(codec):
    # intersection.rs:1173-1194: foo docs 1..4 at positions 1 1 2 1, bar docs 1 3 4 at positions 2 1 3
    foo, bar = gpu(mock(codec, [1, 2, 3, 4], [1, 1, 2, 1])), gpu(mock(codec, [1, 3, 4], [2, 1, 3]))
    ids = lambda **kw: S.intersect([foo, bar], **kw).read()[0].tolist()
    assert ids(max_slop=0) == [1, 3]                      # :1196-1248
    assert ids(in_order=True) == [1, 4]                   # :1250-1296
    assert ids(max_slop=0, in_order=True) == [1]          # :1298-1340
    assert ids() == [1, 3, 4] and ids(max_slop=1) == [1, 3, 4] and ids(max_slop=100, in_order=True) == [1, 4]
    f2, b2 = gpu(mock(codec, [1, 2], [3, 1])), gpu(mock(codec, [1], [1]))
    assert S.intersect([f2, b2], in_order=True).read()[0].tolist() == []     # :1342-1368


def rand_list(rng, codec, n_docs, max_doc, max_pos=40, max_off=6, wide=False):
    docs = np.unique(rng.integers(1, max_doc, n_docs))
    ii = O.InvertedIndex(codec)
    for d in docs.tolist():
        npos = int(rng.integers(0, max_off))
        pos = sorted(set(int(x) for x in rng.integers(1, max_pos, npos)))
        offs, last = b"", 0
        for p in pos:
            offs += O.varint_encode(p - last)
            last = p
        f = int(rng.integers(1, 50))
        if wide:
            ii.add_wide(d, f, int(rng.integers(1, 2 ** 60)) << int(rng.integers(0, 60)), offs)
        else:
            ii.add(d, f, int(rng.integers(1, 2 ** 32 - 1)), offs)
    return ii


@pytest.mark.parametrize("n_lists", [2, 3, 5])
@pytest.mark.parametrize("seed", range(3))
def test_random_slop_order_intersections_match_the_oracle(n_lists, seed):
    """Random lists with 0..5 positions per record (some records carry none), mixed offset codecs, every
    (max_slop, in_order) combination: ids, freqs identical to the oracle's restatement of Intersection + proximity."""
    rng = np.random.default_rng(100 * n_lists + seed)
    codecs = [O.C_FULL, O.C_FREQS_OFFSETS, O.C_OFFSETS_ONLY, O.C_FIELDS_OFFSETS, O.C_FULL_WIDE]
    lists = [rand_list(rng, codecs[(seed + i) % len(codecs)], int(rng.integers(800, 4000)), 5000,
                       wide=codecs[(seed + i) % len(codecs)] in O.WIDE_CODECS) for i in range(n_lists)]
    g = [gpu(l) for l in lists]
    for max_slop, in_order in ((None, False), (0, False), (1, False), (3, False), (10, False), (None, True), (0, True),
                               (2, True), (30, True)):
        h = S.intersect(g, max_slop=max_slop, in_order=in_order)
        gi, gf = h.read()
        oi, of, _ = O.intersect_ex(lists, max_slop, in_order)
        assert gi.tolist() == oi.tolist(), (max_slop, in_order)
        assert gf.tolist() == of.tolist()


def test_lists_without_offsets_do_not_take_part():
    """proximity.rs:284-292: children without offsets are left out of the check; fewer than two remaining -> every
    consensus document passes."""
    rng = np.random.default_rng(3)
    a = rand_list(rng, O.C_FULL, 2000, 4000)
    b = O.InvertedIndex(O.C_FREQS_ONLY)
    b.add_many(np.unique(rng.integers(1, 4000, 2000)).astype(np.uint64))
    g = [gpu(a), gpu(b)]
    plain = S.intersect(g).read()[0].tolist()
    assert S.intersect(g, max_slop=0, in_order=True).read()[0].tolist() == plain
    assert plain == O.intersect_ex([a, b], 0, True)[0].tolist()


@pytest.mark.parametrize("scorer", ["TFIDF", "TFIDF.DOCNORM", "BM25"])
def test_slop_dependent_scorers_use_the_real_offsets(scorer):
    """IndexResult_MinOffsetDelta from the term offsets (index_result.c:51-103): the three slop-dependent scorers over
    offset-carrying lists divide by the hit's real slop, bit-identical to the oracle."""
    rng = np.random.default_rng(11)
    lists = [rand_list(rng, O.C_FULL, 3000, 6000, max_pos=60), rand_list(rng, O.C_FREQS_OFFSETS, 3500, 6000, max_pos=60),
             rand_list(rng, O.C_FULL, 2500, 6000, max_pos=60)]
    g = [gpu(l) for l in lists]
    h = S.intersect(g)
    gi, gf = h.read()
    oi, of, osl = O.intersect_ex(lists)
    assert gi.tolist() == oi.tolist() and len(gi) > 50
    assert len(set(osl.tolist())) > 3                       # the slops really vary
    n_docs = 6000
    doc_len = rng.integers(10, 300, n_docs + 1).astype(np.uint32)
    doc_score = rng.choice([1.0, 0.5], n_docs + 1).astype(np.float32)
    max_freq = rng.integers(1, 60, n_docs + 1).astype(np.uint32)
    table = S.DocTable(doc_len, doc_score, max_freq)
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists]
    bidf = [S.calculate_idf_bm25(n_docs, l.unique_docs) for l in lists]
    w = [1.0, 0.5, 2.0]
    avg = float(doc_len[1:].mean())
    gs = h.score(table, scorer, idf, bidf, w, n_docs, avg)
    sel = oi.astype(np.int64)
    base = O.score_flat(scorer, of, doc_len[sel], max_freq[sel], doc_score[sel], idf, bidf, w, 1.0, n_docs, avg)
    const_slop = len(lists) - 1                              # what score_flat divided by
    exp = base * const_slop / osl                            # undo, apply the real slop
    # bit-identical needs the same operation order: recompute through the tree oracle for a sample
    # the aggregate's children are in ITERATION order: sorted by estimated size, stable (intersection.rs:94-119) -- the
    # slop walks consecutive pairs of that order (the summation order of the term scores follows it too)
    order = sorted(range(len(lists)), key=lambda i: lists[i].unique_docs)
    for j in rng.choice(len(oi), 40, replace=False):
        kids = []
        for li in order:
            l = lists[li]
            r = O.Reader(l)
            r.seek(int(oi[j]))
            pos = r.offsets()
            kids.append(O.term(int(of[li][j]), idf[li], bidf[li], w[li], offsets=pos))
        node = O.intersection(kids)
        want = O.score(scorer, node, float(doc_score[sel[j]]), int(max_freq[sel[j]]), int(doc_len[sel[j]]), n_docs, avg)
        assert gs[j] == want, (scorer, j, gs[j], want)
    assert np.allclose(gs, exp, rtol=1e-12, atol=0)


def test_test_scorers_py_slop_kats():
    """reference tests/pytests/test_scorers.py slop cases: 'hello world' adjacent -> slop 1; one / two words between
    -> the TFIDF score is divided by 2 / 3."""
    n_docs, avg = 3, 4.0
    table = S.DocTable(np.array([0, 2, 3, 4], np.uint32), np.ones(4, np.float32), np.array([0, 1, 1, 1], np.uint32))
    hello = O.InvertedIndex(O.C_FULL)
    world = O.InvertedIndex(O.C_FULL)
    for d, gap in ((1, 1), (2, 2), (3, 3)):
        hello.add(d, 1, 1, O.varint_encode(1))
        world.add(d, 1, 1, O.varint_encode(1 + gap))
    h = S.intersect([gpu(hello), gpu(world)])
    idf = [S.calculate_idf(n_docs, 3)] * 2
    gs = h.score(table, "TFIDF", idf, idf, [1.0, 1.0], n_docs, avg)
    raw = 2 * idf[0]
    assert gs.tolist() == [raw / 1, raw / 2, raw / 3]


@pytest.mark.parametrize("codec", O.WIDE_CODECS)
def test_wide_codecs_decode(codec):
    rng = np.random.default_rng(codec)
    ii = rand_list(rng, codec, 6000, 3_000_000, wide=True)
    g = gpu(ii)
    gi, gf, gm = g.decode()
    oi, of, om = ii.decode_all()
    assert gi.tolist() == oi.tolist() and gf.tolist() == of.tolist() and gm.tolist() == om.tolist()
    assert g.decode_wide_masks() == ii.decode_masks128()
