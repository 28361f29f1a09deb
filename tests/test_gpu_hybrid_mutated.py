"""GPU: the hybrid entry points over an index that is NOT pristine -- the state every live RediSearch index is in.

In the reference a label is a doc id: documents without the vector field have no row, an update is delete + a NEW id
(src/indexer.c:179-190), deletes arrive through VecSimIndex_DeleteVector (src/spec.c:3533-3541), and the hybrid iterator looks
every candidate up by label (src/iterators/hybrid_reader.c:309-327).  Rounds 1-4 kept the label -> row map on the host once the
labels stopped being `base + row`; round 5 keeps it in HBM (csrc/label_table.hpp), so RSGPU_HybridQuery / RSGPU_HybridTreeQuery
stay on their tile kernels (RSGPU_HybridQueryPath 1 / 2) after deletes, re-adds under new labels, overwrites and on multi-value
indexes.  Every case below is held to the CPU ORACLE DIRECTLY -- the oracle's intersection / result-tree scorers in the
reference's order (result_processor.c:849) and an O.FlatIndex that received the same mutations, distance by distance
(VecSimIndex_GetDistanceFrom_Unsafe semantics: a multi-value label's distance is the minimum over its vectors) -- and, bit for
bit, to the staged pipeline behind the same entry point."""
import math
import time
import zlib

import numpy as np
import pytest

import oracle as O
from redisearch_amd import search as S
from redisearch_amd import vecsim as V
from tests.test_gpu_tree import OracleTree, rand_list

pytestmark = pytest.mark.gpu
T, U, I = S.OP_TERM, S.OP_UNION, S.OP_INTERSECT
SCORERS = ["BM25STD", "BM25STD.TANH", "BM25", "TFIDF", "TFIDF.DOCNORM", "DOCSCORE", "DISMAX"]
G_TYPE = {O.F32: V.VecSimType_FLOAT32, O.F16: V.VecSimType_FLOAT16, O.BF16: V.VecSimType_BFLOAT16}
G_METRIC = {O.L2: V.VecSimMetric_L2, O.IP: V.VecSimMetric_IP, O.COSINE: V.VecSimMetric_Cosine}
SEED = 23


def knob(name, value):
    V.load().RSGPU_SetTuning(name.encode(), int(value))


def rows(seed, first_index, n, dim, vtype=O.F32):
    """rows of the keyed corpus as VALUES (O.philox_rows hands BFLOAT16 rows out as bit patterns; to_blob takes values)"""
    r = O.philox_rows(seed, first_index, n, dim, vtype)
    if vtype == O.BF16:
        r = (r.astype(np.uint32) << 16).view(np.float32)
    return r


def mutated_pair(rng, n0, dim, vtype=O.F32, metric=O.L2, multi=False, frac=0.01, first=1):
    """a GPU index and the oracle index after the SAME history: n0 bulk rows labelled first .. first+n0-1, `frac` of them deleted
    at random, documents re-added under NEW labels with gaps between them (documents without a vector), labels overwritten
    (single-value: delete + add; multi-value: a second / third vector), some of the new labels deleted again"""
    g = V.VecSimIndex(G_TYPE[vtype], dim, G_METRIC[metric], multi=multi)
    o = O.FlatIndex(vtype, dim, metric, multi=multi)
    assert g.add_philox_rows(SEED, 0, n0, first) == n0
    o.add_bulk(rows(SEED, 0, n0, dim, vtype), first)
    assert g.label_table() == 0
    m = max(8, int(n0 * frac))
    victims = rng.choice(n0, m, replace=False) + first
    for lab in victims.tolist():
        assert g.delete_vector(lab) == o.delete(lab) == 1
    assert g.label_table() == 1, "the first delete builds the device label table"
    fresh = rows(SEED, 1 << 30, 2 * m, dim, vtype)
    new_labels = first + n0 + 3 * np.arange(m)                      # two of three new doc ids carry no vector
    for j, lab in enumerate(new_labels.tolist()):
        assert g.add_vector(fresh[j], lab) == 1
        o.add(fresh[j], lab)
    alive = np.setdiff1d(np.arange(first, first + n0), victims)
    again = rng.choice(alive, m, replace=False)
    for j, lab in enumerate(again.tolist()):                          # overwrite (single) / one more vector (multi)
        assert g.add_vector(fresh[m + j], lab) == (1 if multi else 0)
        o.add(fresh[m + j], lab)
    if multi:                                                         # a third vector for some, so that chains have length 3
        for j, lab in enumerate(again[: m // 2].tolist()):
            g.add_vector(fresh[j] * np.float32(0.5), lab)
            o.add(fresh[j] * np.float32(0.5), lab)
    for lab in new_labels[:: 7].tolist():
        assert g.delete_vector(lab) == o.delete(lab) == 1
    assert g.index_size() == len(o)
    assert g.label_table() == 1
    return g, o


def oracle_knn(o, doc_ids, q, k):
    """the k nearest of the candidates by (distance, doc id) from the oracle's per-label distances"""
    nq = o.normalized_query(q)
    out = []
    for d in doc_ids:
        x = o.distance_from(int(d), nq)
        if not math.isnan(x):
            out.append((x, int(d)))
    out.sort()
    return out[:k]


def check_knn(ans, want):
    assert ans[0].tolist() == [d for _, d in want], (ans[0][:10], want[:10])
    wd = np.asarray([x for x, _ in want])
    assert np.all(np.abs(ans[1] - wd) <= 1e-4 + 1e-5 * np.abs(wd)), (ans[1][:5], wd[:5])


def run_both(make, want_paths):
    """tile path and the staged pipeline behind the same entry point: bit for bit"""
    hq = make()
    try:
        knob("hybrid_tiles", 1)
        hq.run()
        path = S.hybrid_path()
        a = hq.results()
        hq.run()
        a2 = hq.results()
        knob("hybrid_tiles", 0)
        hq.run()
        assert S.hybrid_path() == 0
        b = hq.results()
    finally:
        knob("hybrid_tiles", 1)
    assert path in want_paths, "the query left the tile kernels (RSGPU_HybridQueryPath %d)" % path
    for x in (a, a2):
        assert x["n_hits"] == b["n_hits"]
        for key in ("top", "knn"):
            assert x[key][0].tolist() == b[key][0].tolist(), key + " ids"
            assert x[key][1].tolist() == b[key][1].tolist(), key + " values"
    return a, path


def freqs_only_lists(rng, n_docs, dfs):
    out = []
    for df in dfs:
        docs = np.flatnonzero(rng.random(n_docs) < df).astype(np.uint64) + 1
        ii = O.InvertedIndex(O.C_FREQS_ONLY)
        ii.add_many(docs, np.minimum(1 + rng.geometric(0.5, docs.size), 255).astype(np.uint32))
        out.append(ii)
    return out


@pytest.mark.parametrize("scorer", SCORERS)
@pytest.mark.parametrize("multi", [False, True])
def test_flat_query_freqs_only_on_a_mutated_index(scorer, multi):
    """RSGPU_HybridQuery, two-launch form (path 1): every scorer family, single- and multi-value indexes"""
    rng = np.random.default_rng(zlib.crc32(scorer.encode()) % 1000 + int(multi))
    n_docs, n0, dim = 400_000, 120_000, 48
    lists_o = freqs_only_lists(rng, n_docs, (0.5, 0.4))
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    doc_len = (50 + rng.poisson(150, n_docs + 1)).astype(np.uint32)
    doc_score = rng.choice([1.0, 0.5, 0.25], n_docs + 1).astype(np.float32)
    max_freq = rng.integers(1, 50, n_docs + 1).astype(np.uint32)
    table = S.DocTable(doc_len, doc_score, max_freq)
    idx, o = mutated_pair(rng, n0, dim, multi=multi)
    q = O.philox_rows(SEED, 1 << 40, 1, dim)[0]
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists_o]
    bidf = [S.calculate_idf_bm25(n_docs, l.unique_docs) for l in lists_o]
    w = [1.0, 0.5]
    a, path = run_both(lambda: S.HybridQuery(g, table, scorer, idf, bidf, w, n_docs, 200.0, top_n=10, index=idx, q=q, k=10,
                                             root_weight=1.25), {1})
    oi, of, _ = O.intersect(lists_o)
    assert a["n_hits"] == len(oi)
    sel = oi.astype(np.int64)
    sc = O.score_flat(scorer, of, doc_len[sel], max_freq[sel], doc_score[sel], idf, bidf, w, 1.25, n_docs, 200.0)
    order = np.lexsort((oi, -sc))[:10]
    assert a["top"][0].tolist() == oi[order].tolist()
    assert np.allclose(a["top"][1], sc[order], rtol=1e-12, atol=0)
    check_knn(a["knn"], oracle_knn(o, oi, q, 10))
    # the per-label ad-hoc seam agrees with the tile kernel's distances, bit for bit
    assert np.array_equal(idx.adhoc_ctx(q).get_exact_distances(a["knn"][0]), a["knn"][1])
    idx.free()


@pytest.mark.parametrize("vtype,metric,dim", [(O.F32, O.COSINE, 768), (O.F16, O.L2, 100), (O.BF16, O.IP, 384), (O.F32, O.IP, 24)])
@pytest.mark.parametrize("multi", [False, True])
def test_element_types_and_row_shapes_on_a_mutated_index(vtype, metric, dim, multi):
    rng = np.random.default_rng(dim + int(multi))
    n_docs, n0 = 200_000, 40_000
    lists_o = freqs_only_lists(rng, n_docs, (0.6, 0.5))
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    idx, o = mutated_pair(rng, n0, dim, vtype, metric, multi=multi, first=20_001)   # vectors cover a window of the doc ids
    q = rows(SEED, 1 << 40, 1, dim, vtype)[0]
    a, path = run_both(lambda: S.HybridQuery(g, index=idx, q=q, k=16), {1})
    oi = O.intersect(lists_o)[0]
    want = oracle_knn(o, oi, q, 16)
    assert a["knn"][0].tolist() == [d for _, d in want]
    wd = np.asarray([x for x, _ in want])
    tol = 1e-4 if vtype == O.F32 else 1e-2                             # (the reference's own fp16 tolerance, test_vecsim.py:14)
    assert np.all(np.abs(a["knn"][1] - wd) <= tol + tol * np.abs(wd))
    idx.free()


@pytest.mark.parametrize("scorer", ["BM25STD", "BM25", "TFIDF", "TFIDF.DOCNORM", "DISMAX"])
@pytest.mark.parametrize("multi", [False, True])
def test_flat_query_full_codec_on_a_mutated_index(scorer, multi):
    """Full-codec lists (FT.CREATE's default): the scorers that divide by the slop take the general tile kernel (path 2), the
    others the two-launch form (path 1)"""
    rng = np.random.default_rng(zlib.crc32(scorer.encode()) % 1000 + 7 * int(multi))
    n_docs = 9000
    built = [rand_list(rng, O.C_FULL, n, n_docs, True) for n in (3000, 4200, 5000)]
    lists_o, recs = [x[0] for x in built], [x[1] for x in built]
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    doc_len = rng.integers(5, 200, n_docs + 1).astype(np.uint32)
    doc_score = rng.choice([1.0, 0.5], n_docs + 1).astype(np.float32)
    max_freq = rng.integers(1, 40, n_docs + 1).astype(np.uint32)
    table = S.DocTable(doc_len, doc_score, max_freq)
    sizes = [l.unique_docs for l in lists_o]
    idf = [S.calculate_idf(n_docs, s) for s in sizes]
    bidf = [S.calculate_idf_bm25(n_docs, s) for s in sizes]
    w = [1.0, 0.5, 2.0]
    idx, o = mutated_pair(rng, 5000, 24, multi=multi, frac=0.02, first=100)
    q = O.philox_rows(SEED, 1 << 40, 1, 24)[0]
    a, path = run_both(lambda: S.HybridQuery(g, table, scorer, idf, bidf, w, n_docs, 180.0, top_n=15, index=idx, q=q, k=10,
                                             root_weight=1.5), {1, 2})
    ot = OracleTree(I, [(T, 1.0, [0]), (T, 1.0, [1]), (T, 1.0, [2])], recs, sizes)
    assert a["n_hits"] == len(ot.docs)
    scored = []
    for d in ot.docs:
        node = ot.node(d, idf, bidf, w)
        node.c.weight = 1.5
        scored.append((O.score(scorer, node, float(doc_score[d]), int(max_freq[d]), int(doc_len[d]), n_docs, 180.0), d))
    scored.sort(key=lambda t: (-t[0], t[1]))
    assert a["top"][0].tolist() == [d for _, d in scored[:15]]
    assert a["top"][1].tolist() == [x for x, _ in scored[:15]]
    check_knn(a["knn"], oracle_knn(o, ot.docs, q, 10))
    idx.free()


TREE_SHAPES = [
    ("term_and_or", [(T, 1.0, [0]), (U, 0.5, [1, 2, 3])]),
    ("or_term_or", [(U, 2.0, [0, 1]), (T, 1.0, [2]), (U, 1.0, [3, 4])]),
    ("or_and_and_term", [(U, 1.0, [0, 1, 2]), (I, 0.7, [3, 4]), (T, 1.0, [5])]),
]


@pytest.mark.parametrize("with_offsets", [False, True])
@pytest.mark.parametrize("multi", [False, True])
@pytest.mark.parametrize("name,shape", TREE_SHAPES)
def test_tree_query_on_a_mutated_index(name, shape, multi, with_offsets):
    """RSGPU_HybridTreeQuery (path 2): FreqsOnly and Full lists, every scorer family, a max_slop window on the Full lists"""
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 10000 + 2 * int(multi) + int(with_offsets))
    codec = O.C_FULL if with_offsets else O.C_FREQS_ONLY
    n_lists = sum(len(gp[2]) for gp in shape)
    max_doc = 2500
    built = [rand_list(rng, codec, int(rng.integers(900, 2200)), max_doc, with_offsets) for _ in range(n_lists)]
    lists_o, recs = [x[0] for x in built], [x[1] for x in built]
    sizes = [l.unique_docs for l in lists_o]
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    groups = [(op, wt, [g[i] for i in ix]) for op, wt, ix in shape]
    doc_len = rng.integers(5, 200, max_doc + 1).astype(np.uint32)
    doc_score = rng.choice([1.0, 0.5], max_doc + 1).astype(np.float32)
    max_freq = rng.integers(1, 40, max_doc + 1).astype(np.uint32)
    table = S.DocTable(doc_len, doc_score, max_freq)
    idf = [S.calculate_idf(max_doc, s) for s in sizes]
    bidf = [S.calculate_idf_bm25(max_doc, s) for s in sizes]
    w = [float(x) for x in rng.choice([1.0, 0.5, 2.0], n_lists)]
    avg = float(doc_len[1:].mean())
    idx, o = mutated_pair(rng, 1200, 24, multi=multi, frac=0.03, first=100)
    q = O.philox_rows(SEED, 1 << 40, 1, 24)[0]
    max_slop = 8 if with_offsets else None
    ot = OracleTree(I, shape, recs, sizes, max_slop, False)
    for scorer in SCORERS:
        a, path = run_both(lambda: S.HybridTreeQuery(I, groups, max_slop=max_slop, table=table, scorer=scorer, idf=idf, bm25_idf=bidf,
                                                     weight=w, num_docs=max_doc, avg_doc_len=avg, top_n=10, index=idx, q=q, k=10,
                                                     root_weight=1.5), {2})
        assert a["n_hits"] == len(ot.docs), scorer
        scored = []
        for d in ot.docs:
            node = ot.node(d, idf, bidf, w)
            node.c.weight = 1.5
            scored.append((O.score(scorer, node, float(doc_score[d]), int(max_freq[d]), int(doc_len[d]), max_doc, avg), d))
        scored.sort(key=lambda t: (-t[0], t[1]))
        assert a["top"][0].tolist() == [d for _, d in scored[:10]], scorer
        if scorer == "BM25STD.TANH":
            assert a["top"][1] == pytest.approx([s for s, _ in scored[:10]], rel=1e-12)
        else:
            assert a["top"][1].tolist() == [s for s, _ in scored[:10]], scorer
        if scorer == SCORERS[0]:
            check_knn(a["knn"], oracle_knn(o, ot.docs, q, 10))
    idx.free()


def test_staged_entry_points_translate_on_the_device_too():
    """RSGPU_Hits_KnnRerank / the staged RSGPU_HybridQuery over a mutated multi-value index: labels -> first rows through the
    device table, the minimum over a label's chain by knn_chain_min_kernel -- the oracle's answer"""
    rng = np.random.default_rng(77)
    n_docs = 60_000
    lists_o = freqs_only_lists(rng, n_docs, (0.5,))
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    for multi in (False, True):
        idx, o = mutated_pair(rng, 20_000, 32, multi=multi)
        q = O.philox_rows(SEED, 1 << 40, 1, 32)[0]
        h = S.intersect(g)
        ki, kd = h.knn_rerank(idx, q, 12)
        check_knn((ki, kd), oracle_knn(o, O.intersect(lists_o)[0], q, 12))
        idx.free()


def test_labels_too_sparse_for_a_direct_table_take_the_device_hash_table():
    """a label 10^12 away from the others: no direct table -- the open-addressing table in HBM (label_table.hpp HASH, round 6;
    rounds 1-5: host hash maps and the staged pipeline's host translation).  The hybrid query stays on its tile kernels."""
    rng = np.random.default_rng(5)
    n_docs, dim = 50_000, 16
    lists_o = freqs_only_lists(rng, n_docs, (0.5, 0.5))
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2)
    o = O.FlatIndex(O.F32, dim, O.L2)
    idx.add_philox_rows(SEED, 0, 10_000, 1)
    o.add_bulk(O.philox_rows(SEED, 0, 10_000, dim), 1)
    v = O.philox_rows(SEED, 99, 1, dim)[0]
    idx.add_vector(v, 10 ** 12)
    o.add(v, 10 ** 12)
    assert idx.label_table() == 2
    idx.delete_vector(17)
    o.delete(17)
    q = O.philox_rows(SEED, 1 << 40, 1, dim)[0]
    hq = S.HybridQuery(g, index=idx, q=q, k=10)
    hq.run()
    assert S.hybrid_path() == 1
    check_knn(hq.results()["knn"], oracle_knn(o, O.intersect(lists_o)[0], q, 10))
    gi, _ = idx.topk_query(v, 1).results()
    assert gi.tolist() == [10 ** 12]
    idx.free()


def test_label_table_follows_a_long_random_history():
    """adds, overwrites and deletes in random order (single- and multi-value), labels below the table's base, labels that make it
    grow: after every burst the device answers (GetDistanceFrom through the table's host copy, a tile-kernel KNN through its
    device copy) equal the oracle's"""
    for multi in (False, True):
        rng = np.random.default_rng(11 + int(multi))
        dim = 8
        idx = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2, multi=multi)
        o = O.FlatIndex(O.F32, dim, O.L2, multi=multi)
        base = 2_000_000                                               # (above 2^20: the table starts at its first label)
        idx.add_philox_rows(SEED, 0, 3000, base)
        o.add_bulk(O.philox_rows(SEED, 0, 3000, dim), base)
        pool = list(range(base - 500, base + 9000))
        docs = np.asarray(pool, np.uint64)
        ii = O.InvertedIndex(O.C_DOCIDS_ONLY)
        ii.add_many(docs, np.ones(docs.size, np.uint32))
        g = [S.Postings.from_flat(ii.flatten())]
        q = O.philox_rows(SEED, 1 << 40, 1, dim)[0]
        for burst in range(6):
            for _ in range(400):
                lab = int(rng.choice(pool))
                if rng.random() < 0.45:
                    assert idx.delete_vector(lab) == o.delete(lab)
                else:
                    v = rng.uniform(-1, 1, dim).astype(np.float32)
                    idx.add_vector(v, lab)
                    o.add(v, lab)
            assert idx.index_size() == len(o)
            assert idx.label_table() == 1
            hq = S.HybridQuery(g, index=idx, q=q, k=20)
            hq.run()
            assert S.hybrid_path() == 1
            check_knn(hq.results()["knn"], oracle_knn(o, pool, q, 20))
            nq = idx.normalized_query(q)
            for lab in rng.choice(pool, 50).tolist():
                a, b = idx.get_distance_from_unsafe(lab, nq), o.distance_from(lab, o.normalized_query(q))
                assert (math.isnan(a) and math.isnan(b)) or a == pytest.approx(b, rel=1e-5, abs=1e-5)
        idx.free()


def test_the_first_delete_is_cheap():
    """leaving identity labelling on a large index: O(1) on the host (a calloc'ed table that encodes "unchanged" as zero), one fill
    kernel on the device -- rounds 1-4 built a hash map of every row under the writer lock"""
    n = 2_000_000
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 16, V.VecSimMetric_L2)
    idx.add_philox_rows(SEED, 0, n, 1)
    idx.topk_query(np.zeros(16, np.float32), 1)
    t0 = time.perf_counter()
    assert idx.delete_vector(n // 2) == 1
    dt = (time.perf_counter() - t0) * 1e3
    assert idx.label_table() == 1
    assert dt < 50.0, "first delete took %.1f ms" % dt
    t0 = time.perf_counter()
    assert idx.delete_vector(n // 3) == 1
    dt2 = (time.perf_counter() - t0) * 1e3
    assert dt2 < 20.0
    idx.free()
