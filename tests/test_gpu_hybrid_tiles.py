"""GPU: RSGPU_HybridQuery in two launches (hybrid_kernels.hip: one tile kernel -- probe, scores, distances, per-tile winners --
and one reduce kernel) against the staged pipeline behind the same entry point (knob hybrid_tiles = 0) and, through it, the
stage-by-stage entry points the oracle tests pin: hit count, top-N ids and scores, KNN ids and distances, BIT FOR BIT -- on every
scorer, element type, metric, row shape (lanes per row / chunks per lane of the scan kernels), list count, codec without
frequencies, 64-bit doc ids, skewed lists whose windows overflow LDS, mass ties (decided by doc id) and N / k from 1 to 64."""
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
    if codec == O.C_FREQS_ONLY:
        ii.add_many(docs, np.asarray(freqs if freqs is not None else np.ones(docs.size), np.uint32))
    else:
        for d in docs.tolist():
            ii.add(d, 1, 1 + (d % 7))
    return ii


def both_paths(make):
    """make() -> HybridQuery; returns (two-launch results, staged results) of the same argument block"""
    lib = V.load()
    hq = make()
    try:
        lib.RSGPU_SetTuning(b"hybrid_tiles", 1)
        hq.run()
        path = S.hybrid_path()
        a = hq.results()
        hq.run()                      # (again: the tickets were put back)
        a2 = hq.results()
        lib.RSGPU_SetTuning(b"hybrid_tiles", 0)
        hq.run()
        assert S.hybrid_path() == 0
        b = hq.results()
    finally:
        lib.RSGPU_SetTuning(b"hybrid_tiles", 1)
    for x in (a, a2):
        assert x["n_hits"] == b["n_hits"]
        assert x["top"][0].tolist() == b["top"][0].tolist(), "top-N ids"
        assert x["top"][1].tolist() == b["top"][1].tolist(), "top-N scores"
        assert x["knn"][0].tolist() == b["knn"][0].tolist(), "KNN ids"
        assert x["knn"][1].tolist() == b["knn"][1].tolist(), "KNN distances"
    return a, b, path


def corpus(n_docs, dfs, seed, codec=O.C_FREQS_ONLY, first=1):
    rng = np.random.default_rng(seed)
    lists_o = []
    for df in dfs:
        docs = np.flatnonzero(rng.random(n_docs) < df).astype(np.uint64) + first
        lists_o.append(postings(docs, np.minimum(1 + rng.geometric(0.5, docs.size), 255), codec))
    return lists_o, rng


SCORERS = ["BM25STD", "BM25STD.TANH", "BM25", "TFIDF", "TFIDF.DOCNORM", "DOCSCORE", "DISMAX"]


@pytest.mark.parametrize("scorer", SCORERS)
@pytest.mark.parametrize("n_lists", [1, 2, 3, 4])
def test_every_scorer_and_list_count(scorer, n_lists):
    n_docs = 700_000
    lists_o, rng = corpus(n_docs, (0.5, 0.45, 0.6, 0.4)[:n_lists], 100 + n_lists)
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    table = S.DocTable((50 + rng.poisson(150, n_docs + 1)).astype(np.uint32), rng.choice([1.0, 0.5, 0.25], n_docs + 1).astype(np.float32),
                       rng.integers(1, 50, n_docs + 1).astype(np.uint32))
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 48, V.VecSimMetric_L2)
    idx.add_philox_rows(7, 0, 200_000, 1)
    q = O.philox_rows(7, 1 << 40, 1, 48)[0]
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists_o]
    bidf = [S.calculate_idf_bm25(n_docs, l.unique_docs) for l in lists_o]
    w = [1.0, 0.5, 2.0, 1.5][:n_lists]
    a, b, path = both_paths(lambda: S.HybridQuery(g, table, scorer, idf, bidf, w, n_docs, 200.0, top_n=10, index=idx, q=q, k=10,
                                                  root_weight=1.25, min_score=0.0))
    assert path == 1
    assert a["n_hits"] == len(O.intersect(lists_o)[0])
    assert len(a["top"][0]) == 10 and len(a["knn"][0]) == 10


@pytest.mark.parametrize("vtype,metric", [(V.VecSimType_FLOAT32, V.VecSimMetric_L2), (V.VecSimType_FLOAT32, V.VecSimMetric_IP),
                                          (V.VecSimType_FLOAT32, V.VecSimMetric_Cosine), (V.VecSimType_FLOAT16, V.VecSimMetric_L2),
                                          (V.VecSimType_FLOAT16, V.VecSimMetric_Cosine), (V.VecSimType_BFLOAT16, V.VecSimMetric_L2),
                                          (V.VecSimType_BFLOAT16, V.VecSimMetric_IP)])
@pytest.mark.parametrize("dim", [4, 24, 100, 384, 768, 1000, 1536])
def test_every_element_type_metric_and_row_shape(vtype, metric, dim):
    """dims chosen for the shapes of pick_shape: one chunk per lane with 1 .. 64 lanes, 16 / 32 lanes with three chunks, 64 lanes
    with 2 .. 6 chunks, rows whose last chunks are padding"""
    n_docs = 400_000
    lists_o, rng = corpus(n_docs, (0.4, 0.3), zlib.crc32(b"%d %d %d" % (vtype, metric, dim)) % 10_000)
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    idx = V.VecSimIndex(vtype, dim, metric)
    idx.add_philox_rows(13, 0, 150_000, 100_001)     # the index covers a window of the doc ids
    t = {V.VecSimType_FLOAT32: O.F32, V.VecSimType_FLOAT16: O.F16, V.VecSimType_BFLOAT16: O.BF16}[vtype]
    q = O.philox_rows(13, 1 << 40, 1, dim, t)[0]
    a, b, path = both_paths(lambda: S.HybridQuery(g, index=idx, q=q, k=10))
    assert path == 1
    assert len(a["knn"][0]) == 10 and 100_001 <= a["knn"][0].min() and a["knn"][0].max() <= 250_000
    # the per-label ad-hoc seam (VecSimIndex_GetDistanceFrom_Unsafe) gives the same distances
    assert np.array_equal(idx.adhoc_ctx(q).get_exact_distances(a["knn"][0]), a["knn"][1])
    idx.free()


@pytest.mark.parametrize("top_n,k", [(1, 1), (3, 32), (32, 5), (32, 32), (48, 7), (10, 50), (64, 64)])
def test_list_lengths_from_one_to_sixty_four_and_fewer_hits_than_asked_for(top_n, k):
    n_docs = 500_000
    lists_o, rng = corpus(n_docs, (0.3, 0.2), 31)
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    table = S.DocTable((50 + rng.poisson(150, n_docs + 1)).astype(np.uint32), np.ones(n_docs + 1, np.float32))
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 32, V.VecSimMetric_L2)
    idx.add_philox_rows(3, 0, 100_000, 1)
    q = O.philox_rows(3, 1 << 40, 1, 32)[0]
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists_o]
    a, b, path = both_paths(lambda: S.HybridQuery(g, table, "BM25STD", idf, idf, [1, 1], n_docs, 200.0, top_n=top_n, index=idx, q=q, k=k))
    assert path == 1 and len(a["top"][0]) == top_n and len(a["knn"][0]) == k
    # a handful of hits, fewer than asked for: three common documents, two of them with a vector
    sa, sb = postings([5, 77, 90_000, 400_000, 499_999]), postings([4, 77, 90_000, 300_000, 499_999])
    g2 = [S.Postings.from_flat(sa.flatten()), S.Postings.from_flat(sb.flatten())]
    a, b, path = both_paths(lambda: S.HybridQuery(g2, table, "BM25STD", [1, 1], [1, 1], [1, 1], n_docs, 200.0, top_n=top_n, index=idx, q=q, k=k))
    assert path == 1 and a["n_hits"] == 3
    assert len(a["top"][0]) == min(top_n, 3) and set(a["top"][0].tolist()) <= {77, 90_000, 499_999}
    assert len(a["knn"][0]) == min(k, 2) and set(a["knn"][0].tolist()) <= {77, 90_000}


def test_mass_ties_are_decided_by_doc_id():
    """DOCSCORE with one document score: every hit ties; the winners are the smallest doc ids, in order -- across tiles and
    reduce blocks.  Identical vectors: every distance ties as well."""
    n_docs = 1_200_000
    lists_o, rng = corpus(n_docs, (0.6, 0.5), 77)
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    table = S.DocTable(np.full(n_docs + 1, 100, np.uint32), np.ones(n_docs + 1, np.float32))
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 8, V.VecSimMetric_L2)
    idx.add_bulk(np.tile(np.arange(8, dtype=np.float32), (300_000, 1)), 200_001)
    hits = O.intersect(lists_o)[0]
    with_vec = hits[(hits >= 200_001) & (hits <= 500_000)]
    # (64: every entry of the first 63 tiles sits at the reduce kernel's bound -- more than it ranks; the exact select settles it)
    for n in (20, 64):
        a, b, path = both_paths(lambda: S.HybridQuery(g, table, "DOCSCORE", [1, 1], [1, 1], [1, 1], n_docs, 100.0, top_n=n, index=idx,
                                                      q=np.zeros(8, np.float32), k=n))
        assert path == 1
        assert a["top"][0].tolist() == hits[:n].tolist() and set(a["top"][1].tolist()) == {1.0}
        assert a["knn"][0].tolist() == with_vec[:n].tolist() and len(set(a["knn"][1].tolist())) == 1


@pytest.mark.parametrize("shape", ["skewed", "window_overflow", "clustered", "dense_equal", "ragged_tail", "disjoint"])
def test_windows_of_every_kind(shape):
    """the other list's window of a tile of 1 024 drivers: inside 4 Ki entries, beyond them (a search in memory), empty"""
    rng = np.random.default_rng(zlib.crc32(shape.encode()) % 1000)
    U = 3_000_000
    if shape == "skewed":
        ls = [np.unique(rng.integers(1, U, 3_000)), np.unique(rng.integers(1, U, 1_500_000))]
    elif shape == "window_overflow":
        ls = [np.arange(1, 900_000, 40), np.arange(1, 900_000)]
    elif shape == "clustered":
        ls = [np.concatenate([np.arange(1, 50_000), np.arange(2_000_000, 2_060_000, 3)]),
              np.concatenate([np.arange(40_000, 1_000_000), np.arange(2_000_000, 2_050_000, 2)])]
    elif shape == "dense_equal":
        a = np.unique(rng.integers(1, 2_000_000, 400_000))
        ls = [a, a.copy()]
    elif shape == "ragged_tail":
        ls = [np.arange(10, 10 + 3 * (2048 + 3), 3), np.unique(rng.integers(1, 800_000, 500_000))]
    else:
        ls = [np.arange(1, 100_000), np.arange(200_000, 300_000)]
    lists_o = [postings(l, rng.integers(1, 9, len(l))) for l in ls]
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    table = S.DocTable((50 + rng.poisson(150, U + 1)).astype(np.uint32), rng.choice([1.0, 0.5], U + 1).astype(np.float32))
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 16, V.VecSimMetric_L2)
    idx.add_philox_rows(5, 0, 1_000_000, 1)
    q = O.philox_rows(5, 1 << 40, 1, 16)[0]
    a, b, path = both_paths(lambda: S.HybridQuery(g, table, "BM25STD", [1.5, 0.5], [1.2, 0.7], [1, 1], U, 200.0, top_n=10, index=idx, q=q, k=10))
    assert path == 1
    assert a["n_hits"] == len(O.intersect(lists_o)[0])
    if shape == "disjoint":
        assert a["n_hits"] == 0 and len(a["top"][0]) == 0 and len(a["knn"][0]) == 0


def test_lists_without_frequencies_and_sixty_four_bit_doc_ids():
    """FieldsOnly lists store no frequency (the term record's default, 1); doc ids beyond 2^32 (the lists keep 32-bit offsets from
    their own bases, the doc table and the index cover windows that start up there)"""
    first = (1 << 33) + 12_345
    n_docs = 300_000
    lists_o, rng = corpus(n_docs, (0.5, 0.4), 19, first=first)
    lists_o[1] = postings(np.flatnonzero(rng.random(n_docs) < 0.4).astype(np.uint64) + first, codec=O.C_FIELDS_ONLY)
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    table = S.DocTable((50 + rng.poisson(150, n_docs)).astype(np.uint32), np.ones(n_docs, np.float32), first_doc_id=first)
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 20, V.VecSimMetric_L2)
    idx.add_philox_rows(9, 0, 100_000, first + 50_000)
    q = O.philox_rows(9, 1 << 40, 1, 20)[0]
    a, b, path = both_paths(lambda: S.HybridQuery(g, table, "TFIDF.DOCNORM", [1.5, 0.5], [1.2, 0.7], [1, 1], n_docs, 200.0, top_n=10,
                                                  index=idx, q=q, k=10))
    assert path == 1 and a["n_hits"] == len(O.intersect(lists_o)[0])
    assert a["top"][0].min() >= first and a["knn"][0].min() >= first + 50_000


def test_shapes_the_two_launches_leave_to_the_staged_pipeline():
    """N > 64: the staged pipeline answers (round 5: 32 before).  Round 4: five lists run the general tile kernel (path 2,
    tests/test_gpu_hybrid_general.py); BM25STD.NORM -- the maximum over ALL hits is the first entry's score -- is ranked as BM25STD
    by the two launches and divided on the host.  Each against the staged pipeline (knob hybrid_tiles = 0), bit for bit."""
    lib = V.load()
    n_docs = 200_000
    lists_o, rng = corpus(n_docs, (0.6, 0.5, 0.6, 0.5, 0.6), 23)
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    table = S.DocTable((50 + rng.poisson(150, n_docs + 1)).astype(np.uint32), rng.choice([1.0, 0.5], n_docs + 1).astype(np.float32))
    ones = [1.0] * 5
    for args, path in ((dict(lists=g[:2], scorer="BM25STD.NORM", top_n=10), 1), (dict(lists=g[:2], scorer="BM25STD.NORM", top_n=63), 1),
                       (dict(lists=g[:2], scorer="BM25STD.NORM", top_n=64), 0),
                       (dict(lists=g, scorer="BM25STD", top_n=10), 2), (dict(lists=g, scorer="BM25STD.NORM", top_n=31), 2),
                       (dict(lists=g, scorer="BM25STD", top_n=64), 2),
                       (dict(lists=g[:2], scorer="BM25STD", top_n=40), 1), (dict(lists=g[:2], scorer="BM25STD", top_n=80), 0)):
        n = len(args["lists"])
        r = S.hybrid_query(args["lists"], table, args["scorer"], ones[:n], ones[:n], ones[:n], n_docs, 200.0, top_n=args["top_n"])
        assert S.hybrid_path() == path and len(r["top"][0]) == args["top_n"], args
        if path:
            try:
                lib.RSGPU_SetTuning(b"hybrid_tiles", 0)
                r0 = S.hybrid_query(args["lists"], table, args["scorer"], ones[:n], ones[:n], ones[:n], n_docs, 200.0, top_n=args["top_n"])
                assert S.hybrid_path() == 0
            finally:
                lib.RSGPU_SetTuning(b"hybrid_tiles", 1)
            assert r0["n_hits"] == r["n_hits"] and r0["top"][0].tolist() == r["top"][0].tolist() and r0["top"][1].tolist() == r["top"][1].tolist()
            if args["scorer"] == "BM25STD.NORM":
                assert r["top"][1][0] == 1.0 and np.all(r["top"][1] <= 1.0)


def test_more_candidates_at_the_bound_than_the_reduce_kernel_ranks():
    """the way out of the reduce kernel (an adversarial arrangement of the tiles' lists leaves more than 2 048 entries at its
    bound; forced here with a cap of 4): since round 5 the exact radix select settles it over the tiles' lists, still in HBM --
    same answers, the query stays on the tile path (rounds 3-4 re-ran it through the staged pipeline; a NOT query failed)"""
    lib = V.load()
    n_docs = 300_000
    lists_o, rng = corpus(n_docs, (0.5, 0.4), 41)
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    table = S.DocTable((50 + rng.poisson(150, n_docs + 1)).astype(np.uint32), np.ones(n_docs + 1, np.float32))
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 16, V.VecSimMetric_L2)
    idx.add_philox_rows(3, 0, 100_000, 1)
    q = O.philox_rows(3, 1 << 40, 1, 16)[0]
    for scorer, top_n, k in (("BM25STD", 10, 10), ("DOCSCORE", 32, 3), ("BM25STD", 1, 32)):   # DOCSCORE: every hit ties -- by doc id
        hq = S.HybridQuery(g, table, scorer, [1.5, 0.5], [1.2, 0.7], [1, 1], n_docs, 200.0, top_n=top_n, index=idx, q=q, k=k)
        hq.run()
        assert S.hybrid_path() == 1
        want = hq.results()
        try:
            lib.RSGPU_SetTuning(b"hybrid_surv_cap", 4)
            for rep in range(2):
                hq.run()
                assert S.hybrid_path() == 1
                got = hq.results()
                assert got["n_hits"] == want["n_hits"]
                for key in ("top", "knn"):
                    assert got[key][0].tolist() == want[key][0].tolist() and got[key][1].tolist() == want[key][1].tolist()
            lib.RSGPU_SetTuning(b"hybrid_tiles", 0)                      # ... and the staged pipeline agrees
            hq.run()
            assert S.hybrid_path() == 0
            st = hq.results()
            for key in ("top", "knn"):
                assert st[key][0].tolist() == want[key][0].tolist() and st[key][1].tolist() == want[key][1].tolist()
        finally:
            lib.RSGPU_SetTuning(b"hybrid_surv_cap", 4096)
            lib.RSGPU_SetTuning(b"hybrid_tiles", 1)
        hq.run()
        assert S.hybrid_path() == 1


@pytest.mark.parametrize("scorer", SCORERS)
def test_top_n_against_the_cpu_oracle_directly(scorer):
    """not through the staged pipeline: the oracle's intersection (find_consensus restated), its scoring loop (default.c restated,
    oracle_score_flat) and the reference's order (score descending, equal scores by doc id: result_processor.c:849) against the
    two-launch answer -- ids exact, scores bit-exact (BM25STD.TANH: 1e-12, the libm tanh)"""
    n_docs = 400_000
    lists_o, rng = corpus(n_docs, (0.35, 0.5, 0.45), 7)
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    doc_len = (50 + rng.poisson(150, n_docs + 1)).astype(np.uint32)
    doc_score = rng.uniform(0.2, 1.0, n_docs + 1).astype(np.float32)
    max_freq = np.maximum(doc_len // 7, 1).astype(np.uint32)
    table = S.DocTable(doc_len, doc_score, max_freq)
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists_o]
    bidf = [S.calculate_idf_bm25(n_docs, l.unique_docs) for l in lists_o]
    w = [1.0, 0.5, 2.0]
    avg = float(doc_len[1:].mean())
    r = S.hybrid_query(g, table, scorer, idf, bidf, w, n_docs, avg, top_n=25, root_weight=0.7)
    assert S.hybrid_path() == 1
    oi, of, _ = O.intersect(lists_o)
    sel = oi.astype(np.int64)
    # the intersection's children -- and so the terms of the scorers' sums -- come in the iterator's order: by estimated size,
    # ascending and stable (intersection.rs:94-119); fp64 addition is not associative, (a + c) + b is not (a + b) + c
    it = np.argsort([l.unique_docs for l in lists_o], kind="stable")
    os_ = O.score_flat(scorer, of[it], doc_len[sel], max_freq[sel], doc_score[sel], [idf[i] for i in it], [bidf[i] for i in it],
                       [w[i] for i in it], 0.7, n_docs, avg)
    order = np.lexsort((oi, -os_))[:25]
    assert r["n_hits"] == len(oi)
    assert r["top"][0].tolist() == oi[order].tolist()
    if scorer == "BM25STD.TANH":
        assert np.allclose(r["top"][1], os_[order], rtol=1e-12, atol=0)
    else:
        assert np.array_equal(r["top"][1], os_[order])


@pytest.mark.parametrize("vtype,otype,metric,ometric,dim", [
    (V.VecSimType_FLOAT32, O.F32, V.VecSimMetric_L2, O.L2, 768), (V.VecSimType_FLOAT32, O.F32, V.VecSimMetric_Cosine, O.COSINE, 128),
    (V.VecSimType_FLOAT16, O.F16, V.VecSimMetric_IP, O.IP, 256)])
def test_knn_branch_against_the_cpu_oracle_directly(vtype, otype, metric, ometric, dim):
    """Round 4: the KNN answer of the two-launch query against the ORACLE, not against the product's other
    paths -- the oracle's intersection gives the candidates, the documents among them that have a vector go into an oracle FLAT
    index of their own (labels = doc ids), its top-k is the answer: ids identical, distances inside the parity tolerance."""
    n_docs, n_vec, k = 300_000, 120_000, 10
    lists_o, rng = corpus(n_docs, (0.3, 0.4), 11)
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    gi = V.VecSimIndex(vtype, dim, metric)
    assert gi.add_philox_rows(5, 0, n_vec, 1) == n_vec           # documents 1 .. n_vec have a vector
    rows = O.philox_rows(5, 0, n_vec, dim, otype)                # (the generator's own rows of that type: what the device made)
    q = O.philox_rows(5, 1 << 40, 1, dim, O.F32)[0]
    try:
        r = S.hybrid_query(g, index=gi, q=q, k=k)
        assert S.hybrid_path() == 1
        oi, _, _ = O.intersect(lists_o)
        cand = oi[oi <= n_vec].astype(np.int64)
        assert r["n_hits"] == len(oi) and len(cand) > 100
        o = O.FlatIndex(otype, dim, ometric)
        o.add_bulk(rows[cand - 1], 1)                            # oracle labels 1 .. m  <->  cand[0 .. m)
        li, ls = o.topk(q, k)
        want_ids = cand[li.astype(np.int64) - 1]
        assert r["knn"][0].tolist() == want_ids.tolist()
        assert np.all(np.abs(r["knn"][1] - ls) <= 1e-4 + 1e-5 * np.abs(ls))
    finally:
        gi.free()
        for x in g:
            x.free()


def test_lean_decode_first_then_the_paths_that_need_the_rest():
    """round 5: the two-launch form decodes a list's doc ids and frequencies only (a Full-codec list: 8 of 20 bytes per posting);
    the field masks and the offsets index are decoded when a path first asks for them.  FRESH Full-codec lists: (1) the
    two-launch query (BM25STD: nothing but ids + freqs), (2) a slop-dependent scorer over the offsets on the general tile kernel,
    (3) the staged intersection with term records (masks, offsets) -- each against the same query on lists that were decoded
    whole from the start (knob decode_lean = 0), bit for bit; and the decode-per-query mode (cache_decoded = 0) the same."""
    from tests.test_gpu_tree import rand_list
    lib = V.load()
    rng = np.random.default_rng(12)
    built = [rand_list(rng, O.C_FULL, n, 60_000, True) for n in (20_000, 30_000)]
    flat = [b[0].flatten() for b in built]
    n_docs = 60_000
    table = S.DocTable(rng.integers(5, 200, n_docs + 1).astype(np.uint32), rng.choice([1.0, 0.5], n_docs + 1).astype(np.float32),
                       rng.integers(1, 40, n_docs + 1).astype(np.uint32))
    idf = [S.calculate_idf(n_docs, b[0].unique_docs) for b in built]
    bidf = [S.calculate_idf_bm25(n_docs, b[0].unique_docs) for b in built]
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 16, V.VecSimMetric_L2)
    idx.add_philox_rows(5, 0, 30_000, 1)
    q = O.philox_rows(5, 1 << 40, 1, 16)[0]

    def run_all(cache):
        out = []
        lib.RSGPU_SetTuning(b"cache_decoded", cache)
        g = [S.Postings.from_flat(f) for f in flat]                    # fresh lists: nothing decoded yet
        hq = S.HybridQuery(g, table, "BM25STD", idf, bidf, [1.0, 2.0], n_docs, 100.0, top_n=10, index=idx, q=q, k=10)
        hq.run()
        assert S.hybrid_path() == 1
        out.append(hq.results())
        hq2 = S.HybridQuery(g, table, "TFIDF", idf, bidf, [1.0, 2.0], n_docs, 100.0, top_n=10, index=idx, q=q, k=10)
        hq2.run()
        assert S.hybrid_path() == 2
        out.append(hq2.results())
        h = S.intersect(g, max_slop=5)
        ids, fr = h.read()
        rec = h.read_records(1)
        out.append(dict(n_hits=len(ids), top=(ids[:50], fr[0][:50].astype(np.float64)), knn=(rec["off_pos"][:50], rec["off_len"][:50].astype(np.float64))))
        assert rec["mask"] == h.read_records(1)["mask"]
        hq.run()                                                        # ... and the lean form again, afterwards
        out.append(hq.results())
        # the general tile kernel decodes lean too when the query walks no offsets: fresh lists again, BM25STD forced through it
        g2 = [S.Postings.from_flat(f) for f in flat]
        try:
            lib.RSGPU_SetTuning(b"hybrid_force_general", 1)
            hq3 = S.HybridQuery(g2, table, "BM25STD", idf, bidf, [1.0, 2.0], n_docs, 100.0, top_n=10, index=idx, q=q, k=10)
            hq3.run()
            assert S.hybrid_path() == 2
            out.append(hq3.results())
        finally:
            lib.RSGPU_SetTuning(b"hybrid_force_general", 0)
        h2 = S.intersect(g2, max_slop=2, in_order=True)                 # ... and then the offsets, on demand
        i2, f2 = h2.read()
        out.append(dict(n_hits=len(i2), top=(i2[:50], f2[0][:50].astype(np.float64)), knn=(i2[:1], f2[1][:1].astype(np.float64))))
        return out

    try:
        for cache in (1, 0):
            lib.RSGPU_SetTuning(b"decode_lean", 1)
            a = run_all(cache)
            lib.RSGPU_SetTuning(b"decode_lean", 0)
            b = run_all(cache)
            for x, y in zip(a, b):
                assert x["n_hits"] == y["n_hits"]
                for key in ("top", "knn"):
                    assert np.asarray(x[key][0]).tolist() == np.asarray(y[key][0]).tolist() and np.asarray(x[key][1]).tolist() == np.asarray(y[key][1]).tolist()
    finally:
        lib.RSGPU_SetTuning(b"decode_lean", 1)
        lib.RSGPU_SetTuning(b"cache_decoded", 1)
    idx.free()
