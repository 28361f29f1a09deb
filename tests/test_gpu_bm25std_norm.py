"""`SCORER BM25STD.NORM` on the device: BM25STD over the hit list, then every score divided by the largest one
(reference RPMaxScoreNormalizer, src/result_processor.c:1770-1812; KATs tests/pytests/test_scorers.py:244-291 on the
oracle side).  fp64, one division per hit: bit-identical to the oracle; the ranking is BM25STD's."""
import numpy as np
import pytest

import oracle as O
from redisearch_amd import search as S
from tests.test_gpu_search import zipf_setup

pytestmark = pytest.mark.gpu
P = lambda ii: S.Postings.from_flat(ii.flatten())


@pytest.mark.parametrize("op", ["and", "or"])
def test_norm_is_bm25std_over_its_maximum(op):
    rng = np.random.default_rng(244)
    N = 120_000
    lists, doc_len, doc_score = zipf_setup(rng, N, [30_000, 12_000, 50_000])
    idf = [S.calculate_idf(N, l.unique_docs) for l in lists]
    bidf = [S.calculate_idf_bm25(N, l.unique_docs) for l in lists]
    w, avg = [1.0, 0.5, 2.0], float(doc_len[1:].mean())
    table = S.DocTable(doc_len, doc_score, np.ones(N + 1, np.uint32))
    h = (S.intersect if op == "and" else S.union)([P(l) for l in lists])
    raw = h.score(table, "BM25STD", idf, bidf, w, N, avg, root_weight=0.7).copy()
    ti0, _ = h.topn(50)
    norm = h.score(table, "BM25STD.NORM", idf, bidf, w, N, avg, root_weight=0.7)
    want = O.max_normalize(raw)
    assert np.array_equal(norm, want) and norm.max() == 1.0
    ti, ts = h.topn(50)                                   # the keys were rewritten too: same ranking, normalised scores
    assert ti.tolist() == ti0.tolist()
    ids, _ = h.read()
    pos = {int(d): i for i, d in enumerate(ids.tolist())}
    assert ts.tolist() == [want[pos[int(d)]] for d in ti.tolist()]


def test_norm_of_all_zero_scores_is_a_no_op():
    rng = np.random.default_rng(245)
    N = 20_000
    lists, doc_len, _ = zipf_setup(rng, N, [5_000, 3_000])
    table = S.DocTable(doc_len, np.zeros(N + 1, np.float32), np.ones(N + 1, np.uint32))   # dmd->score = 0 everywhere
    h = S.intersect([P(l) for l in lists])
    s = h.score(table, "BM25STD.NORM", [1.0, 1.0], [1.0, 1.0], [1.0, 1.0], N, 200.0)
    assert len(s) > 0 and not s.any()
