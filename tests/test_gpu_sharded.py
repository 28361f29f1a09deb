"""GPU: RSGPU_ShardedIndex_* -- one FLAT index row-partitioned over several shards of one process.  On a
1-GPU box all shards sit on device 0 (the code path -- worker threads, concurrent per-shard queries, K-way merge -- is the
one an 8-GPU node runs; with more devices visible the shards spread over them).  Parity: identical to one unsharded index
and to the oracle over the same vectors."""
import math

import ctypes as C

import numpy as np
import pytest
import torch

import oracle as O
from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu
F32 = V.VecSimType_FLOAT32


def _devices(n):
    nd = torch.cuda.device_count()
    return [i % nd for i in range(n)]


@pytest.mark.parametrize("shards", [2, 3, 8])
@pytest.mark.parametrize("metric,om", [(V.VecSimMetric_L2, O.L2), (V.VecSimMetric_Cosine, O.COSINE)])
def test_sharded_topk_equals_unsharded_and_oracle(shards, metric, om):
    n, dim, seed = 30_000, 64, 21
    per = n // shards
    s = V.ShardedIndex(F32, dim, metric, shards, devices=_devices(shards))
    assert s.num_shards() == shards
    for i in range(shards):                       # shard i holds rows [i*per, (i+1)*per) under labels 1 + row
        cnt = per if i < shards - 1 else n - per * (shards - 1)
        assert s.shard(i).add_philox_rows(seed, i * per, cnt, 1 + i * per) == cnt
    assert s.index_size() == n
    one = V.VecSimIndex(F32, dim, metric)
    one.add_philox_rows(seed, 0, n, 1)
    o = O.FlatIndex(O.F32, dim, om)
    o.add_bulk(O.philox_rows(seed, 0, n, dim), 1)
    for qi in range(6):
        q = O.philox_rows(seed, n + qi, 1, dim)[0]
        for k in (1, 10, 100):
            si, ss = s.topk_query(q, k).results()
            ui, us = one.topk_query(q, k).results()
            oi, os_ = o.topk(q, k)
            assert si.tolist() == ui.tolist() == oi.tolist()
            assert ss.tolist() == us.tolist()
            assert np.allclose(ss, os_, rtol=1e-5, atol=1e-4)
        bi, _ = s.topk_query(q, 10, order=V.BY_ID).results()
        assert bi.tolist() == sorted(si[:10].tolist())
    # k larger than one shard's share, and larger than the index
    q = O.philox_rows(seed, n + 50, 1, dim)[0]
    assert s.topk_query(q, per + 5).results()[0].tolist() == one.topk_query(q, per + 5).results()[0].tolist()
    # range: the union of the shards' ranges
    radius = float(one.topk_query(q, 40).results()[1][-1])
    ri, rs = s.range_query(q, radius, order=V.BY_SCORE).results()
    ei, es = one.range_query(q, radius, order=V.BY_SCORE).results()
    assert ri.tolist() == ei.tolist() and rs.tolist() == es.tolist() and len(ri) >= 40


def test_sharded_add_delete_and_distance_routing():
    rng = np.random.default_rng(4)
    dim = 12
    s = V.ShardedIndex(F32, dim, V.VecSimMetric_L2, 3, devices=_devices(3))
    o = O.FlatIndex(O.F32, dim, O.L2)
    x = rng.uniform(-1, 1, (900, dim)).astype(np.float32)
    for i in range(900):
        assert s.add_vector(x[i], i + 1) == 1
        o.add(x[i], i + 1)
    sizes = [s.shard(i).index_size() for i in range(3)]
    assert sizes == [300, 300, 300]                 # new labels go to the emptiest shard
    assert s.add_vector(x[5] * 2, 6) == 0           # overwrite stays on the shard that holds the label
    o.add(x[5] * 2, 6)
    assert s.index_size() == 900
    for lab in (1, 6, 450, 900):
        nq = s.normalized_query(x[0])
        assert s.get_distance_from_unsafe(lab, nq) == pytest.approx(o.distance_from(lab, o.normalized_query(x[0])), rel=1e-5, abs=1e-6)
    assert math.isnan(s.get_distance_from_unsafe(5000, s.normalized_query(x[0])))
    for lab in (10, 11, 12, 500):
        assert s.delete_vector(lab) == 1 and o.delete(lab) == 1
    assert s.delete_vector(10) == 0 and s.index_size() == 896
    for qi in range(4):
        q = rng.uniform(-1, 1, dim).astype(np.float32)
        si, ss = s.topk_query(q, 15).results()
        oi, os_ = o.topk(q, 15)
        assert si.tolist() == oi.tolist() and np.allclose(ss, os_, rtol=1e-5, atol=1e-6)


def test_multi_value_labels_stay_on_one_shard():
    rng = np.random.default_rng(6)
    dim = 8
    s = V.ShardedIndex(F32, dim, V.VecSimMetric_L2, 2, devices=_devices(2), multi=True)
    o = O.FlatIndex(O.F32, dim, O.L2, multi=True)
    for lab in range(1, 201):
        for _ in range(int(rng.integers(1, 4))):
            v = rng.uniform(-1, 1, dim).astype(np.float32)
            s.add_vector(v, lab)
            o.add(v, lab)
    for _ in range(5):
        q = rng.uniform(-1, 1, dim).astype(np.float32)
        si, ss = s.topk_query(q, 20).results()
        oi, os_ = o.topk(q, 20)
        assert si.tolist() == oi.tolist() and len(set(si.tolist())) == 20
        assert np.allclose(ss, os_, rtol=1e-5, atol=1e-6)


def test_replica_mode_answers_from_any_replica():
    n, dim, seed = 20_000, 32, 8
    r = V.ShardedIndex(F32, dim, V.VecSimMetric_L2, 2, devices=_devices(2), replicas=True)
    for i in range(2):
        r.shard(i).add_philox_rows(seed, 0, n, 1)
    assert r.index_size() == n
    one = V.VecSimIndex(F32, dim, V.VecSimMetric_L2)
    one.add_philox_rows(seed, 0, n, 1)
    v = np.full(dim, 0.25, np.float32)
    assert r.add_vector(v, n + 1) == 1 and one.add_vector(v, n + 1) == 1     # writes reach every replica
    assert r.shard(0).index_size() == r.shard(1).index_size() == n + 1
    for qi in range(4):                                                        # round-robin over the replicas
        q = O.philox_rows(seed, n + 10 + qi, 1, dim)[0]
        assert r.topk_query(q, 10).results()[0].tolist() == one.topk_query(q, 10).results()[0].tolist()
    assert r.topk_query(v, 1).results()[0].tolist() == [n + 1]
    assert r.delete_vector(n + 1) == 1 and r.index_size() == n


def test_concurrent_callers_on_a_sharded_index():
    import threading
    n, dim, seed = 40_000, 48, 13
    s = V.ShardedIndex(F32, dim, V.VecSimMetric_L2, 4, devices=_devices(4))
    per = n // 4
    for i in range(4):
        s.shard(i).add_philox_rows(seed, i * per, per, 1 + i * per)
    one = V.VecSimIndex(F32, dim, V.VecSimMetric_L2)
    one.add_philox_rows(seed, 0, n, 1)
    qs = O.philox_rows(seed, n, 16, dim)
    exp = [one.topk_query(q, 10).results()[0].tolist() for q in qs]
    bad = []

    def run(t):
        for it in range(20):
            j = (t * 7 + it) % 16
            if s.topk_query(qs[j], 10).results()[0].tolist() != exp[j]:
                bad.append((t, it))
    th = [threading.Thread(target=run, args=(t,)) for t in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not bad


def test_eight_shards_eight_callers_share_shard_passes():
    """8 shards (oversubscribed on the visible devices), 8 caller threads mixing top-k, range and batch-iterator calls:
    no caller serialises another (the shard workers queue), the workers answer several callers' top-k in one pass over
    their rows, and every reply equals the unsharded index's -- scores included."""
    import threading
    n, dim, seed, shards = 160_000, 128, 17, 8
    s = V.ShardedIndex(F32, dim, V.VecSimMetric_L2, shards, devices=_devices(shards))
    per = n // shards
    for i in range(shards):
        s.shard(i).add_philox_rows(seed, i * per, per, 1 + i * per)
    one = V.VecSimIndex(F32, dim, V.VecSimMetric_L2)
    one.add_philox_rows(seed, 0, n, 1)
    qs = O.philox_rows(seed, n, 32, dim)
    exp = [one.topk_query(q, 10).results() for q in qs]
    exp_r = [one.range_query(q, float(exp[i][1][3]), order=V.BY_ID).results() for i, q in enumerate(qs)]
    bad, bar = [], threading.Barrier(8)
    V.load().RSGPU_SetTuning(b"coalesce_min_mib", 0)   # (the shards are small: by default their workers would not coalesce)
    V.coalesce_stats(reset=True)

    def run(t):
        try:
            bar.wait()
            for it in range(24):
                j = (t * 5 + it) % 32
                ids, sc = s.topk_query(qs[j], 10).results()
                if ids.tolist() != exp[j][0].tolist() or sc.tolist() != exp[j][1].tolist():
                    bad.append(("topk", t, it))
                if it % 6 == t % 6:
                    rid, rs = s.range_query(qs[j], float(exp[j][1][3]), order=V.BY_ID).results()
                    if rid.tolist() != exp_r[j][0].tolist():
                        bad.append(("range", t, it))
        except Exception as e:  # noqa: BLE001
            bad.append(repr(e))
    th = [threading.Thread(target=run, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not bad, bad[:3]
    V.load().RSGPU_SetTuning(b"coalesce_min_mib", 64)
    st = V.coalesce_stats()
    assert st["mq_passes"] > 0 and st["mq_queries"] > st["mq_passes"], st
    ex = (C.c_uint64 * 2)()
    V.load().RSGPU_ShardedIndex_GetExchangeStats(s.ptr, ex, 0)
    assert ex[0] == 8 * 24


# ---- the VecSim ABI handle itself over several shards ("shards" knob): multi-GPU behind the reference's unchanged seam ----
def abi_sharded(vtype, dim, metric, shards, multi=False, replicas=False):
    lib = V.load()
    lib.RSGPU_SetTuning(b"shards", shards)
    lib.RSGPU_SetTuning(b"shard_replicas", int(replicas))
    try:
        return V.VecSimIndex(vtype, dim, metric, multi=multi)
    finally:
        lib.RSGPU_SetTuning(b"shards", 0)
        lib.RSGPU_SetTuning(b"shard_replicas", 0)


@pytest.mark.parametrize("shards,replicas", [(2, False), (3, False), (2, True)])
@pytest.mark.parametrize("metric", [V.VecSimMetric_L2, V.VecSimMetric_Cosine])
def test_abi_handle_over_shards_is_a_drop_in(shards, replicas, metric):
    rng = np.random.default_rng(100 + shards)
    n, dim = 7000, 32
    data = rng.standard_normal((n, dim)).astype(np.float32)
    g = abi_sharded(F32, dim, metric, shards, replicas=replicas)
    one = V.VecSimIndex(F32, dim, metric)
    labels = rng.permutation(n * 3)[:n] + 1                 # arbitrary labels: AddVector routes them to the emptiest shard
    for row, lab in zip(data, labels):
        assert g.add_vector(row, int(lab)) == 1 and one.add_vector(row, int(lab)) == 1
    assert g.index_size() == one.index_size() == n
    flat_info = g.debug_info()
    info = dict(zip(flat_info[0::2], flat_info[1::2]))
    assert info["ALGORITHM"] == "FLAT" and info["INDEX_SIZE"] == n and info["INDEX_LABEL_COUNT"] == n
    for subset in (10, 500, 2000, 6999):
        assert g.prefer_adhoc_search(subset, 10) == one.prefer_adhoc_search(subset, 10)
    for qi in range(4):
        q = rng.standard_normal(dim).astype(np.float32)
        for k in (1, 10, 333):
            a, b = g.topk_query(q, k).results(), one.topk_query(q, k).results()
            assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()
        a, b = g.topk_query(q, 25, order=V.BY_ID).results(), one.topk_query(q, 25, order=V.BY_ID).results()
        assert a[0].tolist() == b[0].tolist()
        radius = float(one.topk_query(q, 60).results()[1][-1])
        a, b = g.range_query(q, radius, order=V.BY_SCORE).results(), one.range_query(q, radius, order=V.BY_SCORE).results()
        assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()
        # the batch iterator: the same sequence of batches, whatever their sizes; Reset starts over
        ig, io = g.batch_iterator(q), one.batch_iterator(q)
        for size, order in ((7, V.BY_SCORE), (1, V.BY_ID), (300, V.BY_ID), (64, V.BY_SCORE), (5000, V.BY_SCORE)):
            assert ig.has_next() == io.has_next()
            a, b = ig.next(size, order).results(), io.next(size, order).results()
            assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()
        ig.reset()
        io.reset()
        a, b = ig.next(12, V.BY_SCORE).results(), io.next(12, V.BY_SCORE).results()
        assert a[0].tolist() == b[0].tolist()
        while io.has_next():                                  # drain: both end together
            assert ig.has_next()
            a, b = ig.next(3000, V.BY_SCORE).results(), io.next(3000, V.BY_SCORE).results()
            assert a[0].tolist() == b[0].tolist()
        assert not ig.has_next()
        ig.free()
        io.free()
        # ad-hoc: labels of every shard and labels nobody holds, in one call
        probe = np.concatenate([labels[:200], [n * 3 + 5, n * 3 + 6]]).astype(np.uint64)
        cg, co = g.adhoc_ctx(q), one.adhoc_ctx(q)
        dg, do = cg.get_exact_distances(probe), co.get_exact_distances(probe)
        assert np.array_equal(dg, do, equal_nan=True) and np.isnan(dg[-1])
        cg.free()
        co.free()
        nq = one.normalized_query(q)
        for lab in labels[:5]:
            assert g.get_distance_from_unsafe(int(lab), nq) == one.get_distance_from_unsafe(int(lab), nq)
    # deletes and overwrites land on the owning shard
    for lab in labels[:50]:
        assert g.delete_vector(int(lab)) == one.delete_vector(int(lab)) == 1
    assert g.delete_vector(int(labels[0])) == 0
    assert g.add_vector(data[0] * 2, int(labels[100])) == one.add_vector(data[0] * 2, int(labels[100]))
    assert g.index_size() == one.index_size() == n - 50
    q = data[0]
    a, b = g.topk_query(q, 20).results(), one.topk_query(q, 20).results()
    assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()


def test_reference_hybrid_iterator_on_a_sharded_handle():
    """The reference's own compiled HybridIterator (oracle/_ref, tests/test_gpu_reference_hybrid_reader.py) driving a
    2-shard handle through the unchanged VecSim ABI: same results, same modes, same iteration counts as on one index."""
    from tests import test_gpu_reference_hybrid_reader as R
    from tests import hybrid_replay as H
    rng = np.random.default_rng(1583)
    n, dim, k = 6000, 24, 10
    data = rng.standard_normal((n, dim)).astype(np.float32)
    g = abi_sharded(F32, dim, V.VecSimMetric_L2, 2)
    one = V.VecSimIndex(F32, dim, V.VecSimMetric_L2)
    for i, row in enumerate(data):
        g.add_vector(row, i + 1)
        one.add_vector(row, i + 1)
    for trial in range(3):
        qv = rng.standard_normal(dim).astype(np.float32)
        child = sorted(rng.choice(np.arange(1, n + 300), int(rng.choice([40, 700, 4000])), replace=False).tolist())
        for kind in ("reference", "batched"):
            for policy, bs in ((0, 0), (H.HYBRID_BATCHES, 7), (H.HYBRID_ADHOC_BF, 0)):
                a, ao = R.run(kind, g, qv, k, child, policy=policy, batch_size=bs)
                b, bo = R.run(kind, one, qv, k, child, policy=policy, batch_size=bs)
                assert a == b and ao.search_mode_out == bo.search_mode_out and ao.num_iterations == bo.num_iterations
    a, ao = R.run("reference", g, data[5], k, None)             # no child: STANDARD_KNN
    assert a[0][0] == 6 and ao.search_mode_out == H.STANDARD_KNN


def test_abi_handle_philox_rows_split_over_shards_and_multi_value():
    g = abi_sharded(F32, 48, V.VecSimMetric_Cosine, 3)
    one = V.VecSimIndex(F32, 48, V.VecSimMetric_Cosine)
    assert g.add_philox_rows(9, 0, 100_000, 1) == 100_000 and one.add_philox_rows(9, 0, 100_000, 1) == 100_000
    sh = V.load().RSGPU_ShardedIndex_FromHandle(g.ptr)
    assert sh and V.load().RSGPU_ShardedIndex_NumShards(sh) == 3
    for qi in range(3):
        q = O.philox_rows(9, 200_000 + qi, 1, 48)[0]
        a, b = g.topk_query(q, 50).results(), one.topk_query(q, 50).results()
        assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()
    # multi-value labels: a label's vectors stay together on the shard that got the first one
    rng = np.random.default_rng(2)
    gm, om = abi_sharded(F32, 16, V.VecSimMetric_L2, 2, multi=True), V.VecSimIndex(F32, 16, V.VecSimMetric_L2, multi=True)
    for i in range(3000):
        v = rng.standard_normal(16).astype(np.float32)
        lab = int(rng.integers(1, 800))
        assert gm.add_vector(v, lab) == om.add_vector(v, lab)
    q = rng.standard_normal(16).astype(np.float32)
    a, b = gm.topk_query(q, 30).results(), om.topk_query(q, 30).results()
    assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()
    ig, io = gm.batch_iterator(q), om.batch_iterator(q)
    for _ in range(4):
        a, b = ig.next(40, V.BY_SCORE).results(), io.next(40, V.BY_SCORE).results()
        assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()


def test_hybrid_entry_points_on_a_sharded_handle_with_deletes():
    """round 5: RSGPU_HybridQuery / RSGPU_HybridTreeQuery over a handle that spans device shards -- no single row matrix, so the
    KNN branch is the staged one (every label routed to its shard, RSGPU_Hits_KnnRerank) -- with deletes and re-adds on the
    shards (each shard keeps its own label table): the answers of a one-device index that received the same history, bit for
    bit; the tile path used to dereference the missing FlatIndex."""
    from redisearch_amd import search as S
    import oracle as O
    rng = np.random.default_rng(21)
    n_docs, n_vec, dim = 120_000, 40_000, 32
    g = abi_sharded(F32, dim, V.VecSimMetric_L2, 3)
    one = V.VecSimIndex(F32, dim, V.VecSimMetric_L2)
    g.add_philox_rows(9, 0, n_vec, 1)
    one.add_philox_rows(9, 0, n_vec, 1)
    fresh = O.philox_rows(9, 1 << 30, 600, dim)
    for lab in rng.choice(n_vec, 400, replace=False).tolist():
        assert g.delete_vector(lab + 1) == one.delete_vector(lab + 1) == 1
    for j in range(600):
        lab = n_vec + 1 + 2 * j
        assert g.add_vector(fresh[j], lab) == one.add_vector(fresh[j], lab) == 1
    assert g.index_size() == one.index_size()
    lists = []
    for df in (0.5, 0.4):
        docs = np.flatnonzero(rng.random(n_docs) < df).astype(np.uint64) + 1
        ii = O.InvertedIndex(O.C_FREQS_ONLY)
        ii.add_many(docs, np.minimum(1 + rng.geometric(0.5, docs.size), 255).astype(np.uint32))
        lists.append(ii)
    gp = [S.Postings.from_flat(l.flatten()) for l in lists]
    table = S.DocTable((50 + rng.poisson(150, n_docs + 1)).astype(np.uint32), np.ones(n_docs + 1, np.float32))
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists]
    q = O.philox_rows(9, 1 << 40, 1, dim)[0]
    for make in (lambda ix: S.HybridQuery(gp, table, "BM25STD", idf, idf, [1.0, 1.0], n_docs, 200.0, top_n=10, index=ix, q=q, k=10),
                 lambda ix: S.HybridTreeQuery(S.OP_UNION, [(S.OP_TERM, 1.0, gp[:1]), (S.OP_TERM, 1.0, gp[1:])], table=table, scorer="BM25STD",
                                              idf=idf, bm25_idf=idf, weight=[1.0, 1.0], num_docs=n_docs, avg_doc_len=200.0, top_n=10, index=ix,
                                              q=q, k=10)):
        a, b = make(g), make(one)
        a.run()
        assert S.hybrid_path() == 0                       # (sharded: staged)
        b.run()
        assert S.hybrid_path() in (1, 2)
        ra, rb = a.results(), b.results()
        assert ra["n_hits"] == rb["n_hits"]
        for key in ("top", "knn"):
            assert ra[key][0].tolist() == rb[key][0].tolist() and ra[key][1].tolist() == rb[key][1].tolist(), key
    g.free()
    one.free()
