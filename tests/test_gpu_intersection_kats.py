"""GPU twins of tests/test_oracle_intersection_kats.py (reference rqe_iterators/tests/integration/intersection.rs):
2 / 5 / 25 children through RSGPU_Intersect -- ids AND the per-child frequencies identical to the oracle -- plus the
same fixtures through RSGPU_Union (union_common.rs fixtures share create_children) and BM25STD over 25 children."""
import numpy as np
import pytest

import oracle as O
from redisearch_amd import search as S
from tests.intersection_cases import NUM_CHILDREN_CASES, RESULT_SET_CASES, create_children, to_index

pytestmark = pytest.mark.gpu
P = lambda ii: S.Postings.from_flat(ii.flatten())


@pytest.mark.parametrize("num_children", NUM_CHILDREN_CASES)
@pytest.mark.parametrize("case", range(len(RESULT_SET_CASES)))
def test_read_all_combinations(num_children, case):
    rs = RESULT_SET_CASES[case]
    lists = [to_index(c) for c in create_children(num_children, rs)]
    oi, of, _ = O.intersect(lists)
    gi, gf = S.intersect([P(l) for l in lists]).read()
    assert gi.tolist() == rs == oi.tolist()
    assert gf.tolist() == of.tolist()


@pytest.mark.parametrize("num_children", NUM_CHILDREN_CASES)
def test_union_of_the_same_children(num_children):
    rs = RESULT_SET_CASES[1]
    children = create_children(num_children, rs)
    lists = [to_index(c) for c in children]
    want = sorted(set().union(*map(set, children)))
    gi, gf = S.union([P(l) for l in lists]).read()
    assert gi.tolist() == want
    for li, c in enumerate(children):                     # freq 0 where the child does not hold the doc
        cs = set(c)
        assert gf[li].tolist() == [(1 + d % 7) if d in cs else 0 for d in want]


def test_edge_cases_and_limits():
    for rs in ([], [3000]):
        lists = [to_index(c) for c in create_children(3, rs)]
        assert S.intersect([P(l) for l in lists]).read()[0].tolist() == rs
    rs = [5000, 6000, 7000]
    lists = [to_index(c) for c in create_children(32, rs)]
    assert S.intersect([P(l) for l in lists]).read()[0].tolist() == rs
    with pytest.raises(RuntimeError):
        S.intersect([P(l) for l in lists] + [P(lists[0])])           # 33 children: refused loudly
    a, b = to_index([1, 1_000_000, 2_000_000_000, 4_000_000_000]), to_index([1, 500, 1_000_000, 3_000_000_000, 4_000_000_000])
    assert S.intersect([P(a), P(b)]).read()[0].tolist() == [1, 1_000_000, 4_000_000_000]


def test_bm25std_over_25_children_matches_oracle():
    rng = np.random.default_rng(25)
    rs = RESULT_SET_CASES[2]
    lists = [to_index(c) for c in create_children(25, rs)]
    n_docs = 3000
    doc_len = (50 + rng.poisson(150, n_docs + 1)).astype(np.uint32)
    doc_score = rng.uniform(0.2, 1.0, n_docs + 1).astype(np.float32)
    max_freq = np.maximum(doc_len // 7, 1).astype(np.uint32)
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists]
    bidf = [S.calculate_idf_bm25(n_docs, l.unique_docs) for l in lists]
    w = rng.uniform(0.5, 2.0, 25).tolist()
    avg = float(doc_len[1:].mean())
    oi, of, _ = O.intersect(lists)
    h = S.intersect([P(l) for l in lists])
    table = S.DocTable(doc_len, doc_score, max_freq)
    sel = oi.astype(np.int64)
    for scorer in ("BM25STD", "TFIDF", "DISMAX"):
        gs = h.score(table, scorer, idf, bidf, w, n_docs, avg)
        os_ = O.score_flat(scorer, of, doc_len[sel], max_freq[sel], doc_score[sel], idf, bidf, w, 1.0, n_docs, avg)
        assert np.array_equal(gs, os_), scorer
