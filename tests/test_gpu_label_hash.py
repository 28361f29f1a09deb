"""GPU: the label -> row map of an index whose doc ids are FAR APART (round 6, csrc/label_table.hpp HASH).

In the reference a label is a doc id of the WHOLE document table -- only documents that carry the vector field get a row
(src/document.c:712-725), an update takes a new id (src/indexer.c:179-190) -- so "1 M vectors in a 10^9-document index" is an
ordinary deployment, and the span of a long-lived index only grows.  Rounds 1-5 sent such an index to host hash maps and every
hybrid entry point with it to the staged pipeline's host translation (RSGPU_HybridQueryPath 0).  Round 6: an open-addressing
table in HBM (16-byte slots, linear probing, at most half full) that the tile kernels, labels_to_rows and the gather read through
the same label_first_row -- the tile paths survive whatever the labels look like.  Everything is held to the CPU oracle: an
O.FlatIndex with the same history, the oracle's intersection, distance by distance."""
import math
import time

import numpy as np
import pytest

import oracle as O
from redisearch_amd import search as S
from redisearch_amd import vecsim as V
from tests.test_gpu_hybrid_mutated import check_knn, oracle_knn

pytestmark = pytest.mark.gpu
SEED = 29
F32, L2 = V.VecSimType_FLOAT32, V.VecSimMetric_L2


def docids_list(docs, codec=O.C_FREQS_ONLY, rng=None):
    ii = O.InvertedIndex(codec)
    f = np.ones(docs.size, np.uint32) if rng is None else np.minimum(1 + rng.geometric(0.5, docs.size), 255).astype(np.uint32)
    ii.add_many(np.asarray(docs, np.uint64), f)
    return ii


def test_one_million_rows_in_a_billion_document_index_stay_on_the_tile_path():
    """the verdict's case: 1 M vectors (dim 16) whose doc ids are scattered over [1, 10^9]; a 2-term filter whose intersection
    holds ~250 k of them + documents without a vector.  The query takes the two-launch form (path 1), the tree query the general
    tile kernel (path 2); answers == oracle; and the same vectors under DENSE labels (a direct table) are not much faster."""
    rng = np.random.default_rng(1)
    n_rows, span, dim, k = 1_000_000, 1_000_000_000, 16, 10
    labels = np.unique(rng.integers(1, span + 1, int(n_rows * 1.01)))[:n_rows]
    assert labels.size == n_rows
    g = V.VecSimIndex(F32, dim, L2)
    g.add_philox_rows(SEED, 0, n_rows, 1)                      # identity labels first ...
    assert g.label_table() == 0
    g.free()
    # ... the scattered index proper: rows by label through the C ABI (AddVector), in blocks
    x = O.philox_rows(SEED, 0, n_rows, dim)
    g = V.VecSimIndex(F32, dim, L2)
    t0 = time.perf_counter()
    for i in range(n_rows):
        g.add_vector(x[i], int(labels[i]))
    add_s = time.perf_counter() - t0
    assert g.index_size() == n_rows and g.label_table() == 2, g.label_table()
    # filter: two term lists over the document space; each holds half of the vector documents + 1.5 M others
    others = np.unique(rng.integers(1, span + 1, 1_600_000))
    la = np.union1d(labels[rng.random(n_rows) < 0.5], others[: 800_000 * 2: 2])
    lb = np.union1d(labels[rng.random(n_rows) < 0.5], others[: 800_000 * 2: 2])
    lists_o = [docids_list(la, rng=rng), docids_list(lb, rng=rng)]
    gl = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    inter = O.intersect(lists_o)[0]
    assert 150_000 < np.intersect1d(inter, labels).size < 350_000 and inter.size > 700_000
    # oracle distances of the candidates that have a vector (an O.FlatIndex of 1 M labelled rows would take minutes to fill)
    row_of = dict(zip(labels.tolist(), range(n_rows)))
    for qi in range(3):
        q = O.philox_rows(SEED, (1 << 40) + qi, 1, dim)[0]
        cand = [d for d in inter.tolist() if d in row_of]
        dist = ((x[[row_of[d] for d in cand]].astype(np.float32) - q.astype(np.float32)) ** 2).sum(axis=1, dtype=np.float32)
        order = np.lexsort((np.asarray(cand), dist))[:k]
        hq = S.HybridQuery(gl, index=g, q=q, k=k)
        hq.run()
        assert S.hybrid_path() == 1
        ki, kd = hq.results()["knn"]
        assert hq.results()["n_hits"] == inter.size
        assert ki.tolist() == [cand[j] for j in order]
        assert np.allclose(kd, dist[order], rtol=1e-5, atol=1e-5)
    # single lookups through the host copy of the same table
    nq = g.normalized_query(q)
    for lab in rng.choice(labels, 20).tolist():
        want = float(((x[row_of[lab]] - q) ** 2).sum(dtype=np.float32))
        assert g.get_distance_from_unsafe(int(lab), nq) == pytest.approx(want, rel=1e-5, abs=1e-5)
    assert math.isnan(g.get_distance_from_unsafe(int(labels[0]) + 1 if int(labels[0]) + 1 not in row_of else 0, nq))
    # timing against the same rows under DENSE doc ids (a direct table after one delete): same lists' shape, same hit counts
    hq = S.HybridQuery(gl, index=g, q=q, k=k)
    for _ in range(20):
        hq.run()
    t = []
    for _ in range(100):
        t0 = time.perf_counter()
        hq.run()
        t.append(time.perf_counter() - t0)
    p50_hash = float(np.percentile(t, 50) * 1e3)
    g.free()
    d = V.VecSimIndex(F32, dim, L2)
    d.add_philox_rows(SEED, 0, n_rows, 1)
    d.delete_vector(n_rows)                                    # identity ends: the direct table
    assert d.label_table() == 1
    dense_a = np.union1d(np.flatnonzero(rng.random(n_rows) < 0.5) + 1, n_rows + 1 + np.arange(800_000))
    dense_b = np.union1d(np.flatnonzero(rng.random(n_rows) < 0.5) + 1, n_rows + 1 + np.arange(800_000))
    dl = [S.Postings.from_flat(docids_list(a, rng=rng).flatten()) for a in (dense_a, dense_b)]
    hd = S.HybridQuery(dl, index=d, q=q, k=k)
    for _ in range(20):
        hd.run()
    assert S.hybrid_path() == 1
    t = []
    for _ in range(100):
        t0 = time.perf_counter()
        hd.run()
        t.append(time.perf_counter() - t0)
    p50_direct = float(np.percentile(t, 50) * 1e3)
    print("hybrid p50: hash table %.4f ms, direct table %.4f ms (x %.2f); 1 M AddVector calls %.1f s" % (p50_hash, p50_direct, p50_hash / p50_direct, add_s))
    assert p50_hash <= 1.5 * p50_direct + 0.02, (p50_hash, p50_direct)     # (the bench records the ratio; 1.1 x is the aim)
    d.free()


@pytest.mark.parametrize("multi", [False, True])
def test_hash_table_follows_a_long_random_history(multi):
    """adds, overwrites, deletes and re-adds in random order under labels near 10^11, spread over 4 x 10^9 (single- and multi-value: chains through
    next[row]); rebuilds happen on the way (the table stays at most half full, tombstones included).  After every burst: the tile
    kernel's KNN through the device copy and GetDistanceFrom through the host copy equal the oracle's."""
    rng = np.random.default_rng(41 + int(multi))
    dim = 8
    idx = V.VecSimIndex(F32, dim, L2, multi=multi)
    o = O.FlatIndex(O.F32, dim, O.L2, multi=multi)
    pool = (10 ** 11 + np.unique(rng.integers(1, 4_000_000_000, 6000))).astype(np.uint64)   # (a list spans < 2^32 doc ids)
    docs = pool.copy()
    g = [S.Postings.from_flat(docids_list(docs, O.C_DOCIDS_ONLY).flatten())]
    q = O.philox_rows(SEED, 1 << 40, 1, dim)[0]
    for burst in range(8):
        for _ in range(700):
            lab = int(rng.choice(pool))
            if rng.random() < 0.4:
                assert idx.delete_vector(lab) == o.delete(lab)
            else:
                v = rng.uniform(-1, 1, dim).astype(np.float32)
                idx.add_vector(v, lab)
                o.add(v, lab)
        assert idx.index_size() == len(o)
        assert idx.label_table() == 2
        hq = S.HybridQuery(g, index=idx, q=q, k=20)
        hq.run()
        assert S.hybrid_path() == 1
        check_knn(hq.results()["knn"], oracle_knn(o, pool.tolist(), q, 20))
        nq = idx.normalized_query(q)
        for lab in rng.choice(pool, 60).tolist():
            a, b = idx.get_distance_from_unsafe(int(lab), nq), o.distance_from(int(lab), o.normalized_query(q))
            assert (math.isnan(a) and math.isnan(b)) or a == pytest.approx(b, rel=1e-5, abs=1e-5)
        ti, _ = idx.topk_query(q, 5).results()
        oi, _ = o.topk(q, 5)
        assert ti.tolist() == oi.tolist()
    idx.free()


def test_a_growing_span_moves_the_direct_table_into_the_hash_table():
    """a long-lived index: doc ids only grow, old documents die.  While the span fits, the direct table; once it passes the limit
    (max(4 MiB, a quarter of the row matrix)), ONE rebuild into the hash table -- and the queries stay on the tile path across it
    (round-5 advisor: the index went to host maps for good at that point)."""
    rng = np.random.default_rng(7)
    dim, k = 8, 10
    idx = V.VecSimIndex(F32, dim, L2)
    o = O.FlatIndex(O.F32, dim, O.L2)
    idx.add_philox_rows(SEED, 0, 5000, 1)
    o.add_bulk(O.philox_rows(SEED, 0, 5000, dim), 1)
    idx.delete_vector(1)
    o.delete(1)
    assert idx.label_table() == 1
    q = O.philox_rows(SEED, 1 << 40, 1, dim)[0]
    next_label, seen_modes = 5001, set()
    live = list(range(2, 5001))
    for step in range(12):
        for _ in range(300):                                   # updates: the oldest documents die, new ids far ahead
            lab = live.pop(0)
            assert idx.delete_vector(lab) == o.delete(lab) == 1
            next_label += int(rng.integers(1, 4000))
            v = rng.uniform(-1, 1, dim).astype(np.float32)
            idx.add_vector(v, next_label)
            o.add(v, next_label)
            live.append(next_label)
        seen_modes.add(idx.label_table())
        docs = np.asarray(sorted(live), np.uint64)
        g = [S.Postings.from_flat(docids_list(docs, O.C_DOCIDS_ONLY).flatten())]
        hq = S.HybridQuery(g, index=idx, q=q, k=k)
        hq.run()
        assert S.hybrid_path() == 1, (step, idx.label_table())
        check_knn(hq.results()["knn"], oracle_knn(o, docs.tolist(), q, k))
    assert seen_modes == {1, 2}, seen_modes                     # the span outgrew 2^20 entries on the way
    idx.free()
