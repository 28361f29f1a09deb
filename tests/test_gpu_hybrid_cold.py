"""GPU: the in-tile decode of the two-launch hybrid query (hybrid_kernels.hip hybrid_tile_kernel<.., COLD>; round-4 verdict, next 3).
Decode-per-query mode (cache_decoded = 0: every query pays what a first touch of a term pays): once a list's first decode has left
its sync points behind, a tile's blocks go from the ENCODED bytes straight into LDS -- the driving list's ten blocks, then the
blocks of every probed list that can hold the tile's doc-id range (a bucket directory over the blocks' first doc ids) -- and the
decoded arrays are never written (the reference's reader decodes as it intersects: inverted_index/src/reader/core.rs skip_to /
next_record; record layouts codec/freqs_only.rs, codec/full.rs:117-190).  RSGPU_HybridQueryColdFused() tells that it ran.
Held BIT FOR BIT -- hit count, top-N ids and scores, KNN ids and distances -- to (a) the same query with the knob
hybrid_cold_fused = 0 (the decode kernel + the tile kernel over decoded arrays), (b) the staged pipeline (hybrid_tiles = 0), which
tests/test_gpu_hybrid_query.py pins to the CPU oracle; the hit count to the oracle's intersection directly."""
import numpy as np
import pytest

import bench as B
import oracle as O
from redisearch_amd import search as S
from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu


def knob(name, value):
    V.load().RSGPU_SetTuning(name.encode(), int(value))


def term_list(rng, n_docs, df, first=1):
    docs, freqs, masks, offs = B._term_list(rng, n_docs, df * n_docs)
    return docs + np.uint64(first - 1), freqs, masks, offs


def encode(raw, codec, block_entries=100):
    d, f, m, o = raw
    return B.encode_freqs_only(d, f, block_entries) if codec == "freqs_only" else B.encode_full(d, f, m, o, block_entries)


def cold_and_the_others(enc, make):
    """enc: upload dicts; make(lists) -> HybridQuery.  -> (fused results, decode-kernel results, staged results, lists)"""
    try:
        knob("cache_decoded", 0)                      # BEFORE the upload: sync points + block directory are laid out for it
        lists = [S.Postings.from_flat(e) for e in enc]
        hq = make(lists)
        hq.run()                                      # a list's FIRST decode leaves its sync points (the decode kernel)
        assert S.hybrid_path() == 1 and S.hybrid_cold_fused() == 0
        first = hq.results()
        hq.run()
        assert S.hybrid_path() == 1 and S.hybrid_cold_fused() == 1, "the tile kernel did not decode the lists itself"
        a = hq.results()
        hq.run()
        assert S.hybrid_cold_fused() == 1
        a2 = hq.results()
        knob("hybrid_cold_fused", 0)
        hq.run()
        assert S.hybrid_path() == 1 and S.hybrid_cold_fused() == 0
        b = hq.results()
        knob("hybrid_tiles", 0)
        hq.run()
        assert S.hybrid_path() == 0
        c = hq.results()
    finally:
        knob("hybrid_tiles", 1)
        knob("hybrid_cold_fused", 1)
        knob("cache_decoded", 1)
    for x in (first, a2, b, c):
        assert x["n_hits"] == a["n_hits"], ("hit count", x["n_hits"], a["n_hits"])
        for key in ("top", "knn"):
            assert x[key][0].tolist() == a[key][0].tolist(), (key, "ids", x[key][0][:8], a[key][0][:8])
            assert x[key][1].tolist() == a[key][1].tolist(), (key, "values")
    return a, b, c, lists


def table_of(rng, n_docs, first=1):
    return S.DocTable((50 + rng.poisson(150, n_docs + 1)).astype(np.uint32), rng.choice([1.0, 0.5, 0.25], n_docs + 1).astype(np.float32),
                      rng.integers(1, 50, n_docs + 1).astype(np.uint32), first_doc_id=first - 1)


@pytest.mark.parametrize("codec", ["freqs_only", "full"])
@pytest.mark.parametrize("n_lists", [1, 2, 3, 4])
@pytest.mark.parametrize("scorer", ["BM25STD", "DISMAX"])
def test_in_tile_decode_against_the_decode_kernel_and_the_staged_form(codec, n_lists, scorer):
    n_docs = 600_000
    rng = np.random.default_rng(100 + n_lists + (7 if codec == "full" else 0))
    raws = [term_list(rng, n_docs, df) for df in (0.3, 0.45, 0.5, 0.4)[:n_lists]]
    table = table_of(rng, n_docs)
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 48, V.VecSimMetric_L2)
    idx.add_philox_rows(7, 0, 150_000, 1)
    q = O.philox_rows(7, 1 << 40, 1, 48)[0]
    idf = [S.calculate_idf(n_docs, r[0].size) for r in raws]
    bidf = [S.calculate_idf_bm25(n_docs, r[0].size) for r in raws]
    w = [1.0, 0.5, 2.0, 1.5][:n_lists]
    a, _, _, lists = cold_and_the_others([encode(r, codec) for r in raws],
                                         lambda g: S.HybridQuery(g, table, scorer, idf, bidf, w, n_docs, 200.0, top_n=10, index=idx, q=q, k=10))
    want = raws[0][0]
    for r in raws[1:]:
        want = np.intersect1d(want, r[0])
    assert a["n_hits"] == want.size and len(a["top"][0]) == 10 and len(a["knn"][0]) == 10
    idx.free()


@pytest.mark.parametrize("vtype,metric,dim", [(V.VecSimType_FLOAT16, V.VecSimMetric_IP, 64), (V.VecSimType_BFLOAT16, V.VecSimMetric_L2, 96),
                                              (V.VecSimType_FLOAT16, V.VecSimMetric_L2, 128), (V.VecSimType_FLOAT32, V.VecSimMetric_Cosine, 32)])
def test_in_tile_decode_on_every_element_type(vtype, metric, dim):
    n_docs = 300_000
    rng = np.random.default_rng(dim)
    raws = [term_list(rng, n_docs, df) for df in (0.4, 0.5)]
    table = table_of(rng, n_docs)
    idx = V.VecSimIndex(vtype, dim, metric)
    idx.add_philox_rows(9, 0, 100_000, 1)
    q = O.philox_rows(9, 1 << 40, 1, dim)[0]
    idf = [S.calculate_idf(n_docs, r[0].size) for r in raws]
    a, _, _, _ = cold_and_the_others([encode(r, "full") for r in raws],
                                     lambda g: S.HybridQuery(g, table, "BM25STD", idf, idf, [1.0, 1.0], n_docs, 200.0, top_n=32, index=idx, q=q, k=64))
    assert len(a["top"][0]) == 32 and len(a["knn"][0]) == 64
    idx.free()


@pytest.mark.parametrize("shape", ["skewed", "sparse_driver", "tiny", "short_blocks", "disjoint", "sixty_four_bit_ids"])
@pytest.mark.parametrize("codec", ["freqs_only", "full"])
def test_in_tile_decode_window_shapes(shape, codec):
    """windows of many block groups (a sparse driver over a dense list: one tile spans the whole list), lists of a few postings,
    blocks shorter than 100 entries (the tile's slots stay compact), lists that share no document, doc ids above 2^32"""
    n_docs = 400_000
    rng = np.random.default_rng(sum(map(ord, shape)) + len(codec))
    first, blk = 1, 100
    if shape == "skewed":
        raws = [term_list(rng, n_docs, 0.02), term_list(rng, n_docs, 0.7)]
    elif shape == "sparse_driver":
        raws = [term_list(rng, n_docs, 0.0008), term_list(rng, n_docs, 0.6), term_list(rng, n_docs, 0.5)]
    elif shape == "tiny":
        few = np.sort(rng.choice(n_docs, 17, replace=False)).astype(np.uint64) + 1
        ff = np.full(17, 3, np.uint32)
        raws = [(few, ff, np.ones(17, np.uint32), rng.integers(1, 128, 51).astype(np.uint8)), term_list(rng, n_docs, 0.9)]
    elif shape == "short_blocks":
        raws, blk = [term_list(rng, n_docs, 0.3), term_list(rng, n_docs, 0.4)], 37
    elif shape == "disjoint":
        d0, f0, m0, o0 = term_list(rng, n_docs, 0.3)
        raws = [(d0[d0 % 2 == 0], f0[d0 % 2 == 0], m0[d0 % 2 == 0], None), None]
        d1, f1, m1, o1 = term_list(rng, n_docs, 0.3)
        raws[1] = (d1[d1 % 2 == 1], f1[d1 % 2 == 1], m1[d1 % 2 == 1], None)
        raws = [(d, f, m, rng.integers(1, 128, int(f.sum())).astype(np.uint8)) for d, f, m, _ in raws]
    else:
        first = (1 << 33) + 12_345
        raws = [term_list(rng, n_docs, 0.3, first), term_list(rng, n_docs, 0.4, first)]
    table = table_of(rng, n_docs, first)
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 32, V.VecSimMetric_L2)
    idx.add_philox_rows(3, 0, 100_000, first)
    q = O.philox_rows(3, 1 << 40, 1, 32)[0]
    n = len(raws)
    idf = [S.calculate_idf(n_docs, max(r[0].size, 1)) for r in raws]
    a, _, _, _ = cold_and_the_others([encode(r, codec, blk) for r in raws],
                                     lambda g: S.HybridQuery(g, table, "BM25STD", idf, idf, [1.0] * n, n_docs, 200.0, top_n=10, index=idx, q=q, k=10))
    want = raws[0][0]
    for r in raws[1:]:
        want = np.intersect1d(want, r[0])
    assert a["n_hits"] == want.size
    if shape == "disjoint":
        assert a["n_hits"] == 0 and len(a["top"][0]) == 0
    if want.size:
        assert a["top"][0].min() >= first
    idx.free()


def test_lists_the_in_tile_decode_leaves_to_the_decode_kernel():
    """other codecs (no frequency / another layout), lists uploaded with the decoded arrays cached: the decode kernel as before"""
    n_docs = 200_000
    rng = np.random.default_rng(5)
    raws = [term_list(rng, n_docs, 0.3), term_list(rng, n_docs, 0.4)]
    table = table_of(rng, n_docs)
    ones = [1.0, 1.0]
    try:
        knob("cache_decoded", 0)
        mixed = [S.Postings.from_flat(encode(raws[0], "freqs_only")), S.Postings.from_flat(encode(raws[1], "full"))]
        hq = S.HybridQuery(mixed, table, "BM25STD", ones, ones, ones, n_docs, 200.0, top_n=10)
        for _ in range(3):
            hq.run()
            assert S.hybrid_path() == 1 and S.hybrid_cold_fused() == 0        # two layouts in one query
        r_mixed = hq.results()
        knob("cache_decoded", 1)
        cached = [S.Postings.from_flat(encode(r, "freqs_only")) for r in raws]
        hq = S.HybridQuery(cached, table, "BM25STD", ones, ones, ones, n_docs, 200.0, top_n=10)
        for _ in range(3):
            hq.run()
            assert S.hybrid_path() == 1 and S.hybrid_cold_fused() == 0
        assert hq.results()["top"][0].tolist() == r_mixed["top"][0].tolist()
    finally:
        knob("cache_decoded", 1)
