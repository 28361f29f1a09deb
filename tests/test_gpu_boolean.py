"""GPU parity of the union / NOT operators over device-resident posting lists (RSGPU_Union, RSGPU_Not) and of
scoring their hit lists, against the CPU oracle: doc ids and frequencies IDENTICAL, fp64 scores bit-exact
(tanh within 1e-12)."""
import numpy as np
import pytest

import oracle as O
from redisearch_amd import search as S
from tests.test_gpu_search import make_list, zipf_setup
from tests.test_oracle_boolean import ids_list

pytestmark = pytest.mark.gpu
P = lambda ii: S.Postings.from_flat(ii.flatten())


@pytest.mark.parametrize("nl", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("codec", [O.C_FREQS_ONLY, O.C_FULL, O.C_RAW_DOCIDS])
def test_union_parity_random(nl, codec):
    rng = np.random.default_rng(nl * 10 + codec)
    lists = [make_list(rng, codec, int(rng.integers(200, 30000)), 60000) for _ in range(nl)]
    oi, of, _ = O.union_lists(lists)
    h = S.union([P(l) for l in lists])
    gi, gf = h.read()
    assert len(h) == len(oi) and gi.tolist() == oi.tolist()
    assert gf.tolist() == of.tolist()


def test_union_and_not_edge_cases():
    e, a, b = ids_list([]), ids_list([10, 20, 30], [1, 2, 3]), ids_list([15, 25, 35], [4, 5, 6])
    ids, fr = S.union([P(e), P(a), P(b)]).read()
    assert ids.tolist() == [10, 15, 20, 25, 30, 35] and fr.tolist() == [[0] * 6, [1, 0, 2, 0, 3, 0], [0, 4, 0, 5, 0, 6]]
    assert len(S.union([P(e), P(e)])) == 0
    assert S.union([P(a), P(a)]).read()[1].tolist() == [[1, 2, 3], [1, 2, 3]]
    # reference rqe_iterators/tests/integration/not.rs:49-123
    assert S.negate(P(ids_list([2, 4, 7])), 10).read()[0].tolist() == [1, 3, 5, 6, 8, 9, 10]
    assert S.negate(P(e), 5).read()[0].tolist() == [1, 2, 3, 4, 5]
    assert len(S.negate(P(ids_list([1, 2, 3, 4, 5])), 5)) == 0
    assert S.negate(P(ids_list([2, 50])), 4).read()[0].tolist() == [1, 3, 4]
    assert len(S.negate(P(a), 0)) == 0
    uni = ids_list([1, 2, 3, 5, 8, 13, 21])
    assert S.negate(P(ids_list([2, 3, 4, 13])), 20, universe=P(uni)).read()[0].tolist() == [1, 5, 8]


@pytest.mark.parametrize("with_universe", [False, True])
def test_not_parity_random(with_universe):
    rng = np.random.default_rng(3 + with_universe)
    child = make_list(rng, O.C_FREQS_ONLY, 40_000, 300_000)
    uni = make_list(rng, O.C_DOCIDS_ONLY, 150_000, 320_000) if with_universe else None
    for max_doc in (1, 1000, 250_000, 400_000):
        want = O.not_list(child, max_doc, universe=uni)
        h = S.negate(P(child), max_doc, universe=P(uni) if uni is not None else None)
        ids, fr = h.read()
        assert len(h) == len(want) and ids.tolist() == want.tolist() and (fr == 1).all()


@pytest.mark.parametrize("scorer", list(S.SCORERS))
def test_union_scoring_parity(scorer):
    """The reference scores a union hit over the children that matched the document only (union_flat.rs:297-320):
    absent terms add nothing, the slop divisor is (matched - 1, at least 1), DISMAX takes the maximum."""
    rng = np.random.default_rng(78)
    N = 60_000
    lists, doc_len, doc_score = zipf_setup(rng, N, [9_000, 4_000, 15_000])
    oi, of, _ = O.union_lists(lists)
    h = S.union([P(l) for l in lists])
    idf = [S.calculate_idf(N, l.unique_docs) for l in lists]
    bidf = [S.calculate_idf_bm25(N, l.unique_docs) for l in lists]
    w = [1.0, 0.5, 2.0]
    max_freq = np.maximum(doc_len // 7, 1).astype(np.uint32)
    avg = float(doc_len[1:].mean())
    gs = h.score(S.DocTable(doc_len, doc_score, max_freq), scorer, idf, bidf, w, N, avg, root_weight=0.7)
    assert len(gs) == len(oi)
    pick = np.unique(np.concatenate([np.arange(200), rng.integers(0, len(oi), 600), np.flatnonzero((of > 0).sum(0) == 3)[:200]]))
    for j in pick.tolist():
        d = int(oi[j])
        kids = [O.term(int(of[t, j]), idf[t], bidf[t], w[t]) for t in range(3) if of[t, j] > 0]
        want = O.score(scorer, O.union(kids, 0.7), float(doc_score[d]), int(max_freq[d]), int(doc_len[d]), N, avg)
        if scorer == "BM25STD.TANH":
            assert abs(gs[j] - want) <= 1e-12 * max(1.0, abs(want))
        else:
            assert gs[j] == want, (scorer, j, gs[j], want)
    # and the top-N over a union's scores
    ti, ts = h.topn(25)
    order = np.lexsort((oi, -gs))[:25]
    assert ti.tolist() == oi[order].tolist()


def test_not_hits_score_as_virtual_results():
    # src/ext/default.c:289-293: a wildcard / virtual result is scored with idf = f = 1
    rng = np.random.default_rng(9)
    N = 5000
    doc_len = (50 + rng.poisson(150, N + 1)).astype(np.uint32)
    doc_score = rng.uniform(0.2, 1.0, N + 1).astype(np.float32)
    child = make_list(rng, O.C_FREQS_ONLY, 1500, N)
    h = S.negate(P(child), N)
    ids, _ = h.read()
    avg = float(doc_len[1:].mean())
    gs = h.score(S.DocTable(doc_len, doc_score, np.ones(N + 1, np.uint32)), "BM25STD", [1.0], [1.0], [0.8], N, avg)
    for j in (0, 1, 17, len(ids) - 1):
        d = int(ids[j])
        want = O.score("BM25STD", O.Node(O.R_VIRTUAL, 0.8, 1), float(doc_score[d]), 1, int(doc_len[d]), N, avg)
        assert gs[j] == want
