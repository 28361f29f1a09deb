"""GPU: RSGPU_HybridQuery (the fused, two-sync pipeline) against the stage-by-stage entry points and the CPU oracle,
and the windowed intersection probe on list shapes that stress its window logic (skewed lengths, clustered ids, gaps
larger than the LDS window, windows that overflow it)."""
import zlib

import numpy as np
import pytest

import oracle as O
from redisearch_amd import search as S
from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu


def postings(docs, freqs=None, codec=O.C_FREQS_ONLY):
    docs = np.asarray(docs, np.uint64)
    ii = O.InvertedIndex(codec)
    ii.add_many(docs, np.asarray(freqs if freqs is not None else np.ones(docs.size), np.uint32))
    return ii


def check_intersection(lists_o):
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    h = S.intersect(g)
    gi, gf = h.read()
    oi, of, _ = O.intersect(lists_o)
    assert gi.tolist() == oi.tolist()
    assert gf.tolist() == of.tolist()
    return len(gi)


@pytest.mark.parametrize("shape", ["balanced", "skewed", "clustered", "window_overflow", "sparse_driver", "three_lists",
                                   "disjoint_ranges", "tiny"])
def test_windowed_intersection_probe_shapes(shape):
    rng = np.random.default_rng(zlib.crc32(shape.encode()) % 1000)
    U = 3_000_000
    if shape == "balanced":
        ls = [np.unique(rng.integers(1, U, 400_000)), np.unique(rng.integers(1, U, 500_000))]
    elif shape == "skewed":            # 2 000 candidates against 1.5 M entries: windows of ~190 k entries overflow LDS
        ls = [np.unique(rng.integers(1, U, 2_000)), np.unique(rng.integers(1, U, 1_500_000))]
    elif shape == "clustered":         # the long list is dense where the driver is empty and vice versa, plus overlap
        a = np.concatenate([np.arange(1, 50_000), np.arange(2_000_000, 2_060_000, 3)])
        b = np.concatenate([np.arange(40_000, 1_000_000), np.arange(2_000_000, 2_050_000, 2)])
        ls = [a, b]
    elif shape == "window_overflow":   # 256 consecutive candidates spanning > 4096 entries of the other list
        a = np.arange(1, 600_000, 40)
        b = np.arange(1, 600_000)
        ls = [a, b]
    elif shape == "sparse_driver":
        ls = [np.array([5, 999_999, 2_999_999]), np.unique(rng.integers(1, U, 800_000))]
    elif shape == "three_lists":
        ls = [np.unique(rng.integers(1, 400_000, 150_000)), np.unique(rng.integers(1, 400_000, 200_000)),
              np.unique(rng.integers(1, 400_000, 250_000))]
    elif shape == "disjoint_ranges":
        ls = [np.arange(1, 100_000), np.arange(200_000, 300_000)]
    else:
        ls = [np.array([7]), np.array([7])]
    lists_o = [postings(l, rng.integers(1, 9, len(l))) for l in ls]
    n = check_intersection(lists_o)
    if shape == "disjoint_ranges":
        assert n == 0
    if shape == "tiny":
        assert n == 1


@pytest.mark.parametrize("shape", ["balanced", "window_mix", "three_lists", "ragged_tail", "dense_equal"])
@pytest.mark.parametrize("dpt", [4, 1])
def test_wide_probe_tiles_on_long_driving_lists(shape, dpt):
    """driving lists of >= 2^18 entries take tiles of 1 024 drivers (four per thread, a window of 8 Ki entries of the other
    list in LDS): same hits and frequencies as the oracle and as the 256-driver tiles (knob probe_dpt = 1), on windows that
    fit 4 Ki, fit only 8 Ki, overflow both, and on a last tile that is mostly empty"""
    lib = V.load()
    rng = np.random.default_rng(zlib.crc32(shape.encode()) % 1000)
    if shape == "balanced":
        ls = [np.unique(rng.integers(1, 6_000_000, 700_000)), np.unique(rng.integers(1, 6_000_000, 900_000))]
    elif shape == "window_mix":      # 1 024 drivers span 40 960 entries (overflow), then 5 851 (8 Ki only), then 2 048
        a = np.arange(1, 12_000_000, 40)
        b = np.concatenate([np.arange(1, 3_000_000), np.arange(3_000_000, 8_000_000, 7), np.arange(8_000_000, 12_000_000, 20)])
        ls = [a, b]
    elif shape == "three_lists":
        ls = [np.unique(rng.integers(1, 900_000, 450_000)), np.unique(rng.integers(1, 900_000, 500_000)),
              np.unique(rng.integers(1, 900_000, 600_000))]
    elif shape == "ragged_tail":     # 2^18 + 3 drivers: the last tile holds three
        a = np.arange(10, 10 + 3 * ((1 << 18) + 3), 3)
        b = np.unique(rng.integers(1, 800_000, 500_000))
        ls = [a, b]
    else:                            # the same list twice: every driver is a hit
        a = np.unique(rng.integers(1, 2_000_000, 400_000))
        ls = [a, a.copy()]
    assert min(len(l) for l in ls) >= 1 << 18
    lists_o = [postings(l, rng.integers(1, 9, len(l))) for l in ls]
    assert lib.RSGPU_SetTuning(b"probe_dpt", dpt) == 0
    try:
        n = check_intersection(lists_o)
    finally:
        lib.RSGPU_SetTuning(b"probe_dpt", 4)
    if shape == "dense_equal":
        assert n == len(ls[0])


def setup_hybrid(n_docs, n_vec, dim, seed, dfs):
    rng = np.random.default_rng(seed)
    lists_o = []
    for df in dfs:
        docs = np.flatnonzero(rng.random(n_docs + 1) < df).astype(np.uint64)
        docs = docs[docs > 0]
        lists_o.append(postings(docs, np.minimum(1 + rng.geometric(0.5, docs.size), 255)))
    doc_len = (50 + rng.poisson(150, n_docs + 1)).astype(np.uint32)
    doc_score = rng.choice([1.0, 0.5, 0.25], n_docs + 1).astype(np.float32)
    table = S.DocTable(doc_len, doc_score, rng.integers(1, 50, n_docs + 1).astype(np.uint32))
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2)
    idx.add_philox_rows(11, 0, n_vec, 1)
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists_o]
    bidf = [S.calculate_idf_bm25(n_docs, l.unique_docs) for l in lists_o]
    return lists_o, g, table, idx, idf, bidf, float(doc_len[1:].mean()), O.philox_rows(11, 1 << 40, 1, dim)[0]


@pytest.mark.parametrize("n_docs,n_vec,dfs", [(2_000_000, 300_000, (0.1, 0.05)),      # 10 k hits: radix top-N path
                                              (4_000_000, 600_000, (0.3, 0.2)),       # 240 k hits: prefilter path
                                              (300_000, 300_000, (0.5, 0.4, 0.3))])   # three lists, every doc has a vector
@pytest.mark.parametrize("scorer", ["BM25STD", "TFIDF", "BM25STD.NORM"])
@pytest.mark.parametrize("tiles", [1, 0])
def test_fused_query_equals_staged_pipeline(n_docs, n_vec, dfs, scorer, tiles):
    """tiles = 1: the query in two launches (hybrid_kernels.hip) -- BM25STD.NORM too since round 4: ranked as BM25STD, the
    division by the largest score (the first entry's) done on the host; tiles = 0: the staged pipeline behind the same entry point"""
    V.load().RSGPU_SetTuning(b"hybrid_tiles", tiles)
    try:
        _fused_query_equals_staged_pipeline(n_docs, n_vec, dfs, scorer, tiles)
    finally:
        V.load().RSGPU_SetTuning(b"hybrid_tiles", 1)


def _fused_query_equals_staged_pipeline(n_docs, n_vec, dfs, scorer, tiles):
    lists_o, g, table, idx, idf, bidf, avg, q = setup_hybrid(n_docs, n_vec, 64, 5, dfs)
    w = [1.0] * len(g)
    # stage by stage
    h = S.intersect(g)
    h.score(table, scorer, idf, bidf, w, n_docs, avg, want_scores=False)
    ti, ts = h.topn(10)
    ki, kd = h.knn_rerank(idx, q, 10)
    # fused
    r = S.hybrid_query(g, table, scorer, idf, bidf, w, n_docs, avg, top_n=10, index=idx, q=q, k=10)
    assert S.hybrid_path() == (1 if tiles else 0)
    assert r["n_hits"] == len(h)
    assert r["top"][0].tolist() == ti.tolist() and r["top"][1].tolist() == ts.tolist()
    assert r["knn"][0].tolist() == ki.tolist() and r["knn"][1].tolist() == kd.tolist()
    # the oracle on the same inputs
    oi, of, _ = O.intersect(lists_o)
    assert r["n_hits"] == len(oi)
    # halves on their own
    r2 = S.hybrid_query(g, table, scorer, idf, bidf, w, n_docs, avg, top_n=10)
    assert r2["top"][0].tolist() == ti.tolist() and len(r2["knn"][0]) == 0
    r3 = S.hybrid_query(g, index=idx, q=q, k=10)
    assert r3["knn"][0].tolist() == ki.tolist() and len(r3["top"][0]) == 0


def test_fused_query_with_mass_score_ties_falls_back_to_the_exact_select():
    """DOCSCORE with three distinct document scores: the 32-bit prefilter cannot separate the hits (all tie at the
    threshold), the candidate list overflows the host cap and the device radix select decides -- by doc id."""
    lists_o, g, table, idx, idf, bidf, avg, q = setup_hybrid(1_000_000, 1000, 16, 9, (0.5, 0.5))
    w = [1.0, 1.0]
    h = S.intersect(g)
    h.score(table, "DOCSCORE", idf, bidf, w, 1_000_000, avg, want_scores=False)
    ti, ts = h.topn(10)
    r = S.hybrid_query(g, table, "DOCSCORE", idf, bidf, w, 1_000_000, avg, top_n=10)
    assert r["top"][0].tolist() == ti.tolist() and r["top"][1].tolist() == ts.tolist()
    assert set(ts.tolist()) == {1.0} and ti.tolist() == sorted(ti.tolist())


def test_fused_query_empty_and_tiny_intersections():
    a, b = postings([1, 5, 9]), postings([2, 6, 10])
    g = [S.Postings.from_flat(a.flatten()), S.Postings.from_flat(b.flatten())]
    table = S.DocTable(np.full(16, 10, np.uint32), np.ones(16, np.float32))
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 4, V.VecSimMetric_L2)
    idx.add_bulk(np.arange(40, dtype=np.float32).reshape(10, 4), 1)
    r = S.hybrid_query(g, table, "BM25STD", [1, 1], [1, 1], [1, 1], 10, 10.0, top_n=5, index=idx, q=np.zeros(4), k=5)
    assert r["n_hits"] == 0 and len(r["top"][0]) == 0 and len(r["knn"][0]) == 0
    c = postings([1, 2, 6, 12])
    g2 = [g[1], S.Postings.from_flat(c.flatten())]
    r = S.hybrid_query(g2, table, "BM25STD", [1, 1], [1, 1], [1, 1], 10, 10.0, top_n=5, index=idx, q=np.zeros(4, np.float32), k=5)
    assert r["n_hits"] == 2 and r["top"][0].tolist() == [2, 6] and r["knn"][0].tolist() == [2, 6]


def test_fused_query_when_the_index_covers_a_window_of_the_doc_ids():
    """the FLAT index holds labels 150 001 .. 450 000 only (identity labels with a base): hits below and above have no
    vector; and repeated calls reuse the re-armed device counters"""
    n_docs, dim = 600_000, 32
    rng = np.random.default_rng(3)
    lists_o = []
    for df in (0.3, 0.25):
        docs = np.flatnonzero(rng.random(n_docs + 1) < df).astype(np.uint64)
        lists_o.append(postings(docs[docs > 0], np.minimum(1 + rng.geometric(0.5, (docs > 0).sum()), 255)))
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    table = S.DocTable((50 + rng.poisson(150, n_docs + 1)).astype(np.uint32), np.ones(n_docs + 1, np.float32))
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2)
    idx.add_philox_rows(11, 0, 300_000, 150_001)
    q = O.philox_rows(11, 1 << 40, 1, dim)[0]
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists_o]
    bidf = [S.calculate_idf_bm25(n_docs, l.unique_docs) for l in lists_o]
    h = S.intersect(g)
    h.score(table, "BM25STD", idf, bidf, [1.0, 1.0], n_docs, 200.0, want_scores=False)
    ti, ts = h.topn(10)
    ki, kd = h.knn_rerank(idx, q, 10)
    assert 150_001 <= ki.min() and ki.max() <= 450_000
    for rep in range(3):
        r = S.hybrid_query(g, table, "BM25STD", idf, bidf, [1.0, 1.0], n_docs, 200.0, top_n=10, index=idx, q=q, k=10)
        assert r["n_hits"] == len(h)
        assert r["top"][0].tolist() == ti.tolist() and r["top"][1].tolist() == ts.tolist()
        assert r["knn"][0].tolist() == ki.tolist() and r["knn"][1].tolist() == kd.tolist()
