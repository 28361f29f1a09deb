"""Full-size (BASELINE configs[1]: 10M x 768 fp32 cosine; configs[2]: 10M x 768 fp16 IP, 256-query batches) parity.

The corpus is bench.py's: the keyed Philox corpus, seed 47, generated in HBM (RSGPU_FlatIndex_AddPhiloxRows).  The host
regenerates the same 10M rows (oracle.philox_rows, threads), loads them into the CPU ORACLE and whole queries -- bench.py's
own first queries among them -- are compared id for id and distance for distance (`assert_topk_parity`): the oracle pins
the headline configuration itself, not a scaled-down stand-in.  On top of that, size-independent properties: planted
near-duplicates come back first and in order, results are sorted, deterministic, consistent between TopK / batches /
range / ad-hoc distances, and a row-sharded merge equals the single-index answer.  The opt-in two-stage exact scan
(int8 shadow) is held to the same oracle at the same size.
Needs ~31 GB of HBM and ~65 GB of host memory; skipped when the device is smaller."""
import numpy as np
import pytest

import oracle as O
from redisearch_amd import vecsim as V
from redisearch_amd.sharded import merge_topk
from tests.util import assert_topk_parity

pytestmark = pytest.mark.gpu
ROWS, DIM, K = 10_000_000, 768, 10
SEED, QUERY_BASE = 47, 1 << 40          # bench.py's corpus and query keys


def _host_gb():
    for l in open("/proc/meminfo"):
        if l.startswith("MemAvailable"):
            return int(l.split()[1]) / 1e6
    return 0.0


@pytest.fixture(scope="module")
def big():
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 80 * 2 ** 30:
        pytest.skip("needs an MI355X-class device")
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, DIM, V.VecSimMetric_Cosine)
    idx.reserve(ROWS + 64)
    assert idx.add_philox_rows(SEED, 0, ROWS, 1) == ROWS
    q = O.philox_rows(SEED, QUERY_BASE + 5000, 1, DIM)[0]
    # K+2 near-duplicates of the query behind the generated rows (labels ROWS+1 ..), closer for smaller j; the last
    # one is the very last row of the corpus
    noise = O.philox_rows(SEED, QUERY_BASE + 6000, K + 2, DIM)
    planted_vecs = np.stack([q + noise[j] * (0.01 * (j + 1)) for j in range(K + 2)]).astype(np.float32)
    planted = []
    for j in range(K + 2):
        assert idx.add_vector(planted_vecs[j], ROWS + 1 + j) == 1
        planted.append(ROWS + 1 + j)
    assert idx.index_size() == ROWS + K + 2
    return idx, q, planted, planted_vecs


@pytest.fixture(scope="module")
def big_oracle(big):
    """The same 10M + K+2 rows in the CPU oracle (regenerated on the host, nothing read back from the GPU)."""
    if _host_gb() < 80:
        pytest.skip("needs ~65 GB of host memory for the oracle's copy of the corpus")
    _, _, planted, planted_vecs = big
    o = O.FlatIndex(O.F32, DIM, O.COSINE)
    step = 2_500_000
    for a in range(0, ROWS, step):            # bounded transient memory: 7.7 GB at a time
        o.add_bulk(O.philox_rows(SEED, a, min(step, ROWS - a), DIM), a + 1)
    for v, lab in zip(planted_vecs, planted):
        o.add(v, lab)
    assert len(o) == ROWS + K + 2
    return o


def test_oracle_parity_at_the_baseline_size(big, big_oracle):
    """Whole queries over the full bench corpus: bench.py's first three timed queries, a query near one stored row, and
    the planted query -- top-10 ids identical to the CPU oracle's (near-ties at rank K excepted and checked), distances
    within 1e-4 (tests/util.py), BY_SCORE and BY_ID; plus K = 100."""
    idx, q, planted, _ = big
    qs = list(O.philox_rows(SEED, QUERY_BASE, 3, DIM)) + [O.philox_rows(SEED, 4_242_424, 1, DIM)[0] * 1.5, q]
    for i, qq in enumerate(qs):
        gi, gs = assert_topk_parity(idx, big_oracle, qq, K)
        if i == 3:
            assert gi[0] == 4_242_425 and gs[0] <= 1e-6          # the stored row itself (cosine is scale-free)
    assert_topk_parity(idx, big_oracle, qs[0], K, order=V.BY_ID)
    assert_topk_parity(idx, big_oracle, qs[1], 100)
    # the ad-hoc seam at full size: per-label distances of arbitrary rows
    nq = idx.normalized_query(qs[2])
    onq = big_oracle.normalized_query(qs[2])
    for lab in (1, 5_000_000, ROWS, ROWS + 3):
        assert abs(idx.get_distance_from_unsafe(lab, nq) - big_oracle.distance_from(lab, onq)) <= 1e-4


def test_two_stage_int8_shadow_at_the_baseline_size(big, big_oracle):
    """The opt-in two-stage exact scan at full size: a second index over the same 10M keyed rows, created with the
    int8 shadow (+25 % HBM), answers the same queries -- checked against the CPU oracle like the plain index, and
    bit-identical to the plain index (ids and distances), K = 10 / 100 / 1000; then the same pair under L2 (plain vs
    shadow, bit-identical)."""
    idx, q, planted, planted_vecs = big
    lib = V.load()

    def shadowed(metric, with_planted):
        lib.RSGPU_SetTuning(b"shadow8", 1)
        try:
            s = V.VecSimIndex(V.VecSimType_FLOAT32, DIM, metric)
        finally:
            lib.RSGPU_SetTuning(b"shadow8", 0)
        s.reserve(ROWS + 64)
        assert s.add_philox_rows(SEED, 0, ROWS, 1) == ROWS
        if with_planted:
            for v, lab in zip(planted_vecs, planted):
                assert s.add_vector(v, lab) == 1
        return s

    sh = shadowed(V.VecSimMetric_Cosine, True)
    qs = list(O.philox_rows(SEED, QUERY_BASE, 3, DIM)) + [O.philox_rows(SEED, 4_242_424, 1, DIM)[0] * 1.5, q]
    lib.RSGPU_ResetProfile()
    lib.RSGPU_SetProfiling(1)
    for qq in qs:
        assert_topk_parity(sh, big_oracle, qq, K)
    lib.RSGPU_SetProfiling(0)
    launches, _, by = V.scan_profile()
    assert launches == len(qs) and by == launches * (ROWS + K + 2) * (DIM + 8)      # every query ran the shadow scan
    for qq in qs[:3]:
        for k in (K, 100, 1000):
            pi, ps = idx.topk_query(qq, k).results()
            si, ss = sh.topk_query(qq, k).results()
            assert si.tolist() == pi.tolist() and ss.tolist() == ps.tolist()
    assert_topk_parity(sh, big_oracle, qs[1], 100)
    # bench.py's 1 000 timed queries: NONE may leave the two-stage path (a fallback is exact but scans four times the
    # bytes -- RSGPU_GetTwoStageStats counts every way out), and the replies are the plain index's, bit for bit
    bench_qs = O.philox_rows(SEED, QUERY_BASE, 1000, DIM)
    V.two_stage_stats(reset=True)
    replies = [sh.topk_query(qq, K).results() for qq in bench_qs]
    st = V.two_stage_stats()
    assert st["attempts"] == 1000 and st["two_stage"] == 1000 and st["fallbacks"] == 0, st
    for i in range(0, 1000, 25):
        pi, ps = idx.topk_query(bench_qs[i], K).results()
        assert replies[i][0].tolist() == pi.tolist() and replies[i][1].tolist() == ps.tolist(), i
    del sh
    plain = V.VecSimIndex(V.VecSimType_FLOAT32, DIM, V.VecSimMetric_L2)
    plain.reserve(ROWS + 64)
    assert plain.add_philox_rows(SEED, 0, ROWS, 1) == ROWS
    sh = shadowed(V.VecSimMetric_L2, False)
    for qq in qs[:3]:
        for k in (K, 100):
            pi, ps = plain.topk_query(qq, k).results()
            si, ss = sh.topk_query(qq, k).results()
            assert si.tolist() == pi.tolist() and ss.tolist() == ps.tolist()
    # a stored row as the query: L2 distance 0 comes first
    si, ss = sh.topk_query(O.philox_rows(SEED, 777_777, 1, DIM)[0], 3).results()
    assert si[0] == 777_778 and ss[0] == 0.0


def test_fp32_native_matrix_core_batch_at_the_baseline_size(big, big_oracle):
    """Round 4 (K5): 256 queries per corpus pass on the PLAIN fp32 index of the headline configuration -- the matrix-core
    filter reads the fp32 rows themselves (bf16 in flight in rounds 4-5, int8 in flight since round 6: asserted), survivors are re-scored exactly.  Every one of 300 replies is
    bit-identical to VecSimIndex_TopKQuery on the same index, the passes were really taken, and whole queries -- the planted
    one among them -- are held to the CPU oracle on the full 10 M + 12 rows."""
    idx, q, planted, _ = big
    lib = V.load()
    queries = np.concatenate([q[None, :], O.philox_rows(SEED, QUERY_BASE + 7000, 299, DIM)]).astype(np.float32)
    for k in (10, 100):
        before = V.coalesce_stats()["mq_passes"]
        lib.RSGPU_ResetProfile()
        lib.RSGPU_SetProfiling(1)
        ids, sc, cnt = idx.topk_batch(queries, k)
        lib.RSGPU_SetProfiling(0)
        assert V.scan_profile()[0] == 2 and V.coalesce_stats()["mq_passes"] == before, "the matrix-core passes were not taken"
        assert lib.RSGPU_LastBatchRoute() == 6   # (round 6: the fp32 rows quantised to int8 in flight, the default route; rsgpu_ext.h)
        assert (cnt == k).all()
        for i in range(0, 300, 7):
            si, ss = idx.topk_query(queries[i], k).results()
            assert ids[i].tolist() == si.tolist() and sc[i].tolist() == ss.tolist(), (k, i)
        assert ids[0][:min(k, K + 2)].tolist() == planted[:min(k, K + 2)]
        for i in (0, 1, 147, 294):          # (queries whose batched reply was just shown to BE the single-query reply)
            gi, gs = assert_topk_parity(idx, big_oracle, queries[i], k)
            assert ids[i].tolist() == gi.tolist() and sc[i].tolist() == gs.tolist()


def test_fp32_native_l2_batch_at_the_baseline_size():
    """... and on an L2 index of the same rows (per-row band through the half norms): bit-identical to single queries"""
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 80 * 2 ** 30:
        pytest.skip("needs an MI355X-class device")
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, DIM, V.VecSimMetric_L2)
    try:
        idx.reserve(ROWS)
        assert idx.add_philox_rows(SEED, 0, ROWS, 1) == ROWS
        queries = O.philox_rows(SEED, QUERY_BASE + 8000, 256, DIM).astype(np.float32)
        lib = V.load()
        lib.RSGPU_ResetProfile()
        lib.RSGPU_SetProfiling(1)
        ids, sc, cnt = idx.topk_batch(queries, 100)
        lib.RSGPU_SetProfiling(0)
        assert V.scan_profile()[0] == 1, "the matrix-core pass was not taken"
        assert (cnt == 100).all()
        for i in range(0, 256, 5):
            si, ss = idx.topk_query(queries[i], 100).results()
            assert ids[i].tolist() == si.tolist() and sc[i].tolist() == ss.tolist(), i
        # size-independent: ascending scores, and row 12345 is its own nearest neighbour at distance 0
        assert np.all(np.diff(sc, axis=1) >= 0)
        own = O.philox_rows(SEED, 12344, 1, DIM).astype(np.float32)
        oi, os_, _ = idx.topk_batch(np.repeat(own, 2, axis=0), 1)
        assert oi[0][0] == 12345 and os_[0][0] == 0.0
    finally:
        idx.free()


def test_planted_neighbours_found_in_order(big):
    idx, q, planted, _ = big
    ids, sc = idx.topk_query(q, K).results()
    assert ids.tolist() == planted[:K]
    assert np.all(np.diff(sc) > 0) and sc[0] < 1e-3 and sc[-1] < 0.05
    ids2, sc2 = idx.topk_query(q, K).results()          # deterministic
    assert ids2.tolist() == ids.tolist() and sc2.tolist() == sc.tolist()
    ids12, _ = idx.topk_query(q, K + 2).results()       # includes the very last row of the corpus
    assert ids12.tolist() == planted


def test_consistency_between_entry_points(big):
    idx, q, planted, _ = big
    ids, sc = idx.topk_query(q, K).results()
    # ad-hoc distances of the winners == their TopK scores (same kernel family, same order of sums)
    nq = idx.normalized_query(q)
    for i, s in zip(ids[:3], sc[:3]):
        assert abs(idx.get_distance_from_unsafe(int(i), nq) - s) <= 1e-6
    ctx = idx.adhoc_ctx(q)
    assert np.allclose(ctx.get_exact_distances(ids), sc, atol=1e-6)
    ctx.free()
    # range query with radius = k-th distance returns exactly the top-k set
    rid, rsc = idx.range_query(q, float(sc[-1]), order=V.BY_SCORE).results()
    assert rid.tolist() == ids.tolist()
    # batches: first batch BY_ID == sorted top-k ids, second batch continues strictly above
    it = idx.batch_iterator(q)
    b1, s1 = it.next(K, V.BY_ID).results()
    assert b1.tolist() == sorted(ids.tolist())
    b2, s2 = it.next(5, V.BY_SCORE).results()
    assert s2.min() >= sc.max() and len(set(b2.tolist()) & set(ids.tolist())) == 0
    assert b2[:2].tolist() == planted[K:K + 2]
    it.free()


def test_row_sharded_merge_equals_global(big):
    """Weak-scaling shape in one process: 8 contiguous label ranges, per-shard top-k via range-restricted
    candidates, merged by (score,label) == the global top-k."""
    idx, q, planted, _ = big
    ids, sc = idx.topk_query(q, 64).results()
    shards = 8
    per = ROWS // shards
    all_s, all_l = [], []
    for g in range(shards):
        m = (ids > g * per) & ((ids <= (g + 1) * per) | (g == shards - 1))
        l = np.full(K, np.uint64(0xFFFFFFFFFFFFFFFF))
        s = np.full(K, np.inf, dtype=np.float32)
        l[: min(K, m.sum())] = ids[m][:K]
        s[: min(K, m.sum())] = sc[m][:K]
        all_s.append(s)
        all_l.append(l)
    labels, scores = merge_topk(np.concatenate(all_s), np.concatenate(all_l), K)
    assert labels.tolist() == ids[:K].tolist()


def test_delete_and_overwrite_at_full_size(big):
    idx, q, planted, _ = big
    assert idx.delete_vector(planted[0]) == 1               # moves the last row (a planted one) into the hole
    ids, _ = idx.topk_query(q, K).results()
    assert ids.tolist() == planted[1:K + 1]
    assert idx.index_size() == ROWS + K + 2 - 1
    assert idx.add_vector(q, planted[0]) == 1               # exact duplicate of the query: distance ~0, first
    ids, sc = idx.topk_query(q, 3).results()
    assert ids[0] == planted[0] and sc[0] <= 1e-6 and ids[1:].tolist() == planted[1:3]


def test_batched_config3_full_size_against_the_oracle():
    """BASELINE configs[2] at full size: 10M x 768 fp16 IP (the keyed corpus in fp16), 256 queries per pass, top-100.
    Eight of the 256 queries are compared with the CPU oracle over the same 10M rows at exact-id level (a member may
    differ only when its oracle distance ties rank K within the fp32 tolerance); planted rows (scaled copies of a query,
    so their inner product is the largest by far) lead that query's list in order; all lists are sorted and free of
    duplicates; the batch equals the single-query path."""
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 80 * 2 ** 30:
        pytest.skip("needs an MI355X-class device")
    rows, dim, k, nq = 10_000_000, 768, 100, 256
    queries = O.philox_rows(SEED, QUERY_BASE, nq, dim, O.F16)
    idx = V.VecSimIndex(V.VecSimType_FLOAT16, dim, V.VecSimMetric_IP)
    idx.reserve(rows + 64)
    assert idx.add_philox_rows(SEED, 0, rows, 1) == rows
    plant = {7: [3.0, 2.5, 2.0], 200: [4.0, 3.5]}                       # query -> scales of its planted copies
    planted_rows, want = [], {}
    lab = rows + 1
    for qi, scales in plant.items():
        want[qi] = []
        for sc_ in scales:
            v = (queries[qi].astype(np.float32) * sc_).astype(np.float16)
            assert idx.add_vector(v, lab) == 1
            planted_rows.append((v, lab))
            want[qi].append(lab)
            lab += 1
    ids, sc, cnt = idx.topk_batch(queries, k)
    assert V.load().RSGPU_LastBatchRoute() == 5   # (round 6: the fp16 rows quantised to int8 in flight, the default route; rsgpu_ext.h)
    assert (cnt == k).all()
    for qi, labs in want.items():
        assert ids[qi, :len(labs)].tolist() == labs
    assert np.all(np.diff(sc, axis=1) >= 0)
    assert all(len(set(ids[i].tolist())) == k for i in range(nq))
    # round 5: this route re-scores its survivors with the single-query scan's arithmetic like every other batched route --
    # the batch IS the single-query path, id for id and bit for bit
    for qi in (0, 7, 31, 64, 100, 128, 200, 255):
        si, ss = idx.topk_query(queries[qi], k).results()
        assert si.tolist() == ids[qi].tolist(), qi
        assert np.array_equal(ss, sc[qi]), qi
    if _host_gb() < 50:
        pytest.skip("oracle leg needs ~35 GB of host memory")
    o = O.FlatIndex(O.F16, dim, O.IP)
    step = 2_500_000
    for a0 in range(0, rows, step):
        o.add_bulk(O.philox_rows(SEED, a0, min(step, rows - a0), dim, O.F16), a0 + 1)
    for v, l in planted_rows:
        o.add(v, l)
    tol = 1e-4   # the oracle's scalar fp32 loop against the scan's lane order: north_star's fp32 tolerance
    for qi in (0, 7, 31, 64, 100, 128, 200, 255):
        oi, os_ = o.topk(queries[qi], k)
        gset, oset = set(ids[qi].tolist()), set(oi.tolist())
        nqb = o.normalized_query(queries[qi])
        for lab_ in gset ^ oset:                                       # only a true fp32 tie at rank K may differ
            assert abs(o.distance_from(int(lab_), nqb) - os_[-1]) <= tol * max(1.0, abs(os_[-1])), (qi, lab_)
        assert len(gset ^ oset) <= 2, (qi, len(gset ^ oset))
        assert np.allclose(np.sort(sc[qi]), np.sort(os_), rtol=1e-4, atol=tol)
        common = [x for x in ids[qi].tolist() if x in oset]
        god = dict(zip(ids[qi].tolist(), sc[qi].tolist()))
        ood = dict(zip(oi.tolist(), os_.tolist()))
        assert max(abs(god[x] - ood[x]) / max(1.0, abs(ood[x])) for x in common) <= tol


def test_batched_config3_int8_shadow_full_size():
    """BASELINE configs[2] through the opt-in int8 shadow (one index-wide scale, int8 MFMA filter passes, fp16 re-scoring):
    at 10M x 768 the 256 replies must be BIT-IDENTICAL to single queries on the same fp16 index -- 16 of them are compared
    id for id and distance for distance -- planted rows lead their query's list, and the passes must have read the int8
    shadow (one byte per element)."""
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 80 * 2 ** 30:
        pytest.skip("needs an MI355X-class device")
    rows, dim, k, nq = 10_000_000, 768, 100, 256
    lib = V.load()
    queries = O.philox_rows(SEED, QUERY_BASE, nq, dim, O.F16)
    lib.RSGPU_SetTuning(b"shadow8", 1)
    try:
        idx = V.VecSimIndex(V.VecSimType_FLOAT16, dim, V.VecSimMetric_IP)
    finally:
        lib.RSGPU_SetTuning(b"shadow8", 0)
    idx.reserve(rows + 64)
    assert idx.add_philox_rows(SEED, 0, rows, 1) == rows
    lab, want = rows + 1, {}
    for qi, scales in {7: [0.99, 0.97], 200: [0.98]}.items():          # (below 1: the index-wide scale stays put)
        want[qi] = []
        for sc_ in scales:
            assert idx.add_vector((queries[qi].astype(np.float32) * sc_).astype(np.float16), lab) == 1
            want[qi].append(lab)
            lab += 1
    lib.RSGPU_ResetProfile()
    lib.RSGPU_SetProfiling(1)
    ids, sc, cnt = idx.topk_batch(queries, k)
    lib.RSGPU_SetProfiling(0)
    launches, _, by = V.scan_profile()
    assert launches == 1 and by == (rows + 3) * dim                      # one pass over the int8 shadow
    assert (cnt == k).all() and np.all(np.diff(sc, axis=1) >= 0)
    for qi, labs in want.items():
        assert ids[qi, :len(labs)].tolist() == labs
    for qi in list(range(0, 256, 17)) + [255]:
        si, ss = idx.topk_query(queries[qi], k).results()
        assert si.tolist() == ids[qi].tolist() and ss.tolist() == sc[qi].tolist(), qi
