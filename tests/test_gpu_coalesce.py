"""The query coalescer behind VecSimIndex_TopKQuery and its multi-query scan (scan_mq_kernels.hip).

VecSim answers one query per call (reference src/iterators/hybrid_reader.c:374); RediSearch issues the calls from its
worker threads (src/util/workers.c:58,104).  Calls that arrive while a corpus pass is in flight join the next pass, which
scores every row against all of them.  The contract tested here: a coalesced reply is BIT-IDENTICAL (ids and scores)
to the reply of the same query issued alone, for every row shape / type / metric the multi-query kernel has, whatever
the number of queries in the pass and whatever their K; and the oracle agrees with both."""
import threading

import numpy as np
import pytest
import torch

import oracle as O
from redisearch_amd import vecsim as V
from tests.util import assert_topk_parity

pytestmark = pytest.mark.gpu


@pytest.fixture()
def knobs():
    lib = V.load()
    set_ = []

    def setk(key, val):
        assert lib.RSGPU_SetTuning(key.encode(), int(val)) == 0
        set_.append(key)
    yield setk
    defaults = {"coalesce": 1, "coalesce_min_mib": 64, "coalesce_linger_us": -1, "filter_select": 1, "batch_mfma": 1, "mq16": 1, "coalesce_shadow8": 1}
    for key in set_:
        lib.RSGPU_SetTuning(key.encode(), defaults[key])


def _index(vtype, dim, metric, n, seed=5):
    idx = V.VecSimIndex(vtype, dim, metric)
    assert idx.add_philox_rows(seed, 0, n, 1) == n
    return idx


def _queries(vtype, dim, nq, seed=6):
    s = V.VecSimIndex(vtype, dim, V.VecSimMetric_L2)
    try:
        assert s.add_philox_rows(seed, 1 << 40, nq, 1) == nq
        return s.read_rows(0, nq)
    finally:
        s.free()


SHAPES = [  # (type, dim): every (G, ITERS) shape of the multi-query kernel, exact and masked
    (V.VecSimType_FLOAT32, 128), (V.VecSimType_FLOAT32, 100), (V.VecSimType_FLOAT32, 256), (V.VecSimType_FLOAT32, 384),
    (V.VecSimType_FLOAT32, 512), (V.VecSimType_FLOAT32, 768), (V.VecSimType_FLOAT32, 700), (V.VecSimType_FLOAT32, 1024),
    (V.VecSimType_FLOAT16, 256), (V.VecSimType_FLOAT16, 768), (V.VecSimType_FLOAT16, 1024), (V.VecSimType_FLOAT16, 1536),
    (V.VecSimType_FLOAT16, 2048), (V.VecSimType_BFLOAT16, 768), (V.VecSimType_BFLOAT16, 1000), (V.VecSimType_BFLOAT16, 2048),
]


@pytest.mark.parametrize("vtype,dim", SHAPES)
@pytest.mark.parametrize("metric", [V.VecSimMetric_L2, V.VecSimMetric_IP, V.VecSimMetric_Cosine])
def test_multi_query_pass_is_bit_identical_to_single_queries(vtype, dim, metric, knobs):
    """RSGPU_FlatIndex_TopKBatch without the matrix-core passes runs the multi-query scan, sixteen queries per pass: 38
    queries = passes of 16, 16 and 6 -- the sixteen-query kernel (FLOAT32 rows up to 3 KiB; two launches of eight
    otherwise) and the eight-query one -- then 3 queries through the four-query kernel."""
    knobs("batch_mfma", 0)   # (fp16 / bf16 IP batches would take the MFMA passes: equal only up to the summation order)
    n = 70_000   # >= 2^16: the batched threshold-filter selection at K <= 32
    idx = _index(vtype, dim, metric, n)
    try:
        qs = _queries(vtype, dim, 38)
        before = V.coalesce_stats()
        for k in (1, 10, 100):
            ids, sc, cnt = idx.topk_batch(qs, k)
            for i in range(len(qs)):
                si, ss = idx.topk_query(qs[i], k).results()
                assert cnt[i] == len(si) == k
                assert ids[i][:k].tolist() == si.tolist(), (i, k)
                assert sc[i][:k].tolist() == ss.tolist(), (i, k)
        after = V.coalesce_stats()
        assert after["mq_passes"] - before["mq_passes"] == 9 and after["mq_queries"] - before["mq_queries"] == 114
        ids, sc, cnt = idx.topk_batch(qs[:3], 10)
        for i in range(3):
            si, ss = idx.topk_query(qs[i], 10).results()
            assert ids[i].tolist() == si.tolist() and sc[i].tolist() == ss.tolist()
        if vtype == V.VecSimType_FLOAT32 and dim <= 768:      # ... and the same passes as two launches of eight
            knobs("mq16", 0)
            ids, sc, cnt = idx.topk_batch(qs[:16], 10)
            for i in range(16):
                si, ss = idx.topk_query(qs[i], 10).results()
                assert ids[i].tolist() == si.tolist() and sc[i].tolist() == ss.tolist()
    finally:
        idx.free()


@pytest.mark.parametrize("n", [3_000, 30_000, 66_000, 300_000])
def test_multi_query_pass_selection_paths_against_the_oracle(n):
    """short arrays (one select workgroup per query), the batched filter path, the radix levels (K > 32) -- each held to
    the CPU oracle, and to the single-query path bit for bit."""
    dim, metric = 128, V.VecSimMetric_L2
    rng = np.random.default_rng(n)
    data = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    g = V.VecSimIndex(V.VecSimType_FLOAT32, dim, metric)
    o = O.FlatIndex(O.F32, dim, metric)
    try:
        g.add_bulk(data)
        o.add_bulk(data)
        qs = rng.uniform(-1, 1, (5, dim)).astype(np.float32)
        for k in (3, 32, 64):
            ids, sc, cnt = g.topk_batch(qs, k)
            for i, q in enumerate(qs):
                gi, gs = assert_topk_parity(g, o, q, k)
                assert ids[i].tolist() == gi.tolist() and sc[i].tolist() == gs.tolist()
    finally:
        g.free()


def test_equal_rows_tie_break_in_a_multi_query_pass():
    """every row equal: the top-K is decided by the row index alone (first-seen wins), in the batched selection too"""
    n, dim = 70_000, 128
    data = np.tile(np.linspace(-1, 1, dim, dtype=np.float32), (n, 1))
    g = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_IP)
    try:
        g.add_bulk(data)
        qs = np.random.default_rng(3).uniform(-1, 1, (4, dim)).astype(np.float32)
        ids, sc, cnt = g.topk_batch(qs, 10)
        for i in range(4):
            assert ids[i].tolist() == list(range(1, 11))
            si, ss = g.topk_query(qs[i], 10).results()
            assert si.tolist() == ids[i].tolist() and ss.tolist() == sc[i].tolist()
    finally:
        g.free()


def _hammer(idx, qs, ks, want, n_threads, reps, orders=None):
    errors, barrier = [], threading.Barrier(n_threads)

    def worker(t):
        try:
            barrier.wait()
            for rep in range(reps):
                for i in range(t, len(qs), n_threads):
                    order = orders[i] if orders else V.BY_SCORE
                    rep_ = idx.topk_query(qs[i], ks[i], order=order)
                    assert rep_.code == V.VecSim_QueryReply_OK
                    ids, sc = rep_.results()
                    wi, ws = want[i]
                    if order == V.BY_ID:
                        o = np.argsort(wi, kind="stable")
                        wi, ws = wi[o], ws[o]
                    assert ids.tolist() == wi.tolist() and sc.tolist() == ws.tolist(), (i, ks[i])
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    [x.start() for x in th]
    [x.join() for x in th]
    return errors


@pytest.mark.parametrize("vtype,dim,metric", [(V.VecSimType_FLOAT32, 768, V.VecSimMetric_Cosine),
                                              (V.VecSimType_FLOAT32, 128, V.VecSimMetric_L2),
                                              (V.VecSimType_FLOAT16, 768, V.VecSimMetric_IP)])
def test_concurrent_callers_share_passes_and_get_their_serial_answers(vtype, dim, metric, knobs):
    """8 threads through the plain C ABI on one index: the replies equal the serial ones bit for bit, with different K
    and reply orders inside one pass, and the coalescer did put several callers into one pass."""
    knobs("coalesce_min_mib", 0)
    n = 200_000
    idx = _index(vtype, dim, metric, n)
    try:
        qs = _queries(vtype, dim, 48)
        ks = [(1, 10, 10, 10, 32, 100)[i % 6] for i in range(len(qs))]
        orders = [V.BY_ID if i % 5 == 0 else V.BY_SCORE for i in range(len(qs))]
        knobs("coalesce", 0)
        want = [idx.topk_query(q, k).results() for q, k in zip(qs, ks)]
        knobs("coalesce", 1)
        V.coalesce_stats(reset=True)
        errors = _hammer(idx, qs, ks, want, 8, 6, orders)
        assert not errors, errors[:3]
        st = V.coalesce_stats()
        assert st["queries"] == 48 * 6
        assert st["mq_passes"] > 0 and st["mq_queries"] > st["mq_passes"], st
    finally:
        idx.free()


@pytest.mark.parametrize("dim,metric", [(768, V.VecSimMetric_Cosine), (256, V.VecSimMetric_L2), (512, V.VecSimMetric_IP),
                                        (1024, V.VecSimMetric_Cosine)])
def test_concurrent_callers_on_an_int8_shadow_index_share_two_stage_passes(dim, metric, knobs):
    """FLOAT32 index created with shadow8: concurrent K <= 16 callers share multi-query passes over the int8 shadow
    (scan_mq_i8_kernel + the batched form of the two-stage filter + exact re-scoring); the replies equal those of a plain
    index without a shadow, bit for bit; a K = 100 caller in the mix is answered on the single path."""
    lib = V.load()
    n = 300_000                                         # >= 2^18: the two-stage path's own threshold
    qs = _queries(V.VecSimType_FLOAT32, dim, 40)
    ks = [(10, 1, 16, 10, 5)[i % 5] for i in range(len(qs))]
    orders = [V.BY_ID if i % 7 == 0 else V.BY_SCORE for i in range(len(qs))]
    plain = _index(V.VecSimType_FLOAT32, dim, metric, n)
    try:
        knobs("coalesce", 0)
        want = [plain.topk_query(q, k).results() for q, k in zip(qs, ks)]
        want100 = plain.topk_query(qs[0], 100).results()
    finally:
        plain.free()
    lib.RSGPU_SetTuning(b"shadow8", 1)          # (two_stage, the query-time switch, is on by default)
    try:
        idx = _index(V.VecSimType_FLOAT32, dim, metric, n)
    finally:
        lib.RSGPU_SetTuning(b"shadow8", 0)
    try:
        knobs("coalesce", 1)
        knobs("coalesce_min_mib", 0)
        V.coalesce_stats(reset=True)
        lib.RSGPU_ResetTwoStageStats()
        errors = _hammer(idx, qs, ks, want, 8, 5, orders)
        assert not errors, errors[:3]
        st, ts = V.coalesce_stats(), V.two_stage_stats()
        assert st["queries"] == 40 * 5
        assert st["mq_passes"] > 0 and st["mq_queries"] > st["mq_passes"], st
        assert ts["two_stage"] >= st["mq_queries"] - st["mq_redo"] and ts["fallbacks"] == 0, (ts, st)
        # K above the multi-query form's limit: coalesced with nobody, same answer
        ids, sc = idx.topk_query(qs[0], 100).results()
        assert ids.tolist() == want100[0].tolist() and sc.tolist() == want100[1].tolist()
        # ... and with the multi-query form switched off the callers still get their answers (one two-stage scan each)
        knobs("coalesce_shadow8", 0)
        V.coalesce_stats(reset=True)
        assert not _hammer(idx, qs, ks, want, 4, 1, orders)
        assert V.coalesce_stats()["mq_passes"] == 0
    finally:
        idx.free()


def test_coalescer_off_and_small_corpora_do_not_coalesce(knobs):
    n, dim = 50_000, 64      # 12.8 MB: below coalesce_min_mib -> concurrent streams, no shared passes
    idx = _index(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2, n)
    try:
        qs = _queries(V.VecSimType_FLOAT32, dim, 16)
        want = [idx.topk_query(q, 10).results() for q in qs]
        V.coalesce_stats(reset=True)
        assert not _hammer(idx, qs, [10] * 16, want, 4, 3)
        assert V.coalesce_stats()["passes"] == 0
    finally:
        idx.free()


def test_writers_and_timeouts_next_to_coalesced_readers(knobs):
    """AddVector / DeleteVector take the index's write lock between passes; a caller whose timeout callback fires gets
    TimedOut while the others of its pass get their answers."""
    knobs("coalesce_min_mib", 0)
    n, dim = 120_000, 128
    idx = _index(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2, n)
    try:
        qs = _queries(V.VecSimType_FLOAT32, dim, 24)
        stop = threading.Event()
        errors = []

        def writer():
            try:
                lab = 10_000_000
                while not stop.is_set():
                    idx.add_vector(np.full(dim, 5.0, np.float32), lab)   # far from every query: never in a top-10
                    idx.delete_vector(lab)
                    lab += 1
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))
        want = [idx.topk_query(q, 10).results() for q in qs]
        w = threading.Thread(target=writer)
        w.start()
        try:
            errs = _hammer(idx, qs, [10] * len(qs), want, 6, 4)
        finally:
            stop.set()
            w.join()
        assert not errs and not errors, (errs[:2], errors[:2])

        # timeouts: ctx != NULL means "timed out" to this callback
        cb = V.set_timeout_callback(lambda ctx: 1 if ctx else 0)
        try:
            res = [None] * 6
            bar = threading.Barrier(6)

            def caller(t):
                qp = V.VecSimQueryParams()
                qp.timeoutCtx = 1 if t % 2 else None
                bar.wait()
                res[t] = idx.topk_query(qs[t], 10, params=qp)
            th = [threading.Thread(target=caller, args=(t,)) for t in range(6)]
            [x.start() for x in th]
            [x.join() for x in th]
            for t in range(6):
                if t % 2:
                    assert res[t].code == V.VecSim_QueryReply_TimedOut and len(res[t].results()[0]) == 0
                else:
                    assert res[t].code == V.VecSim_QueryReply_OK
                    assert res[t].results()[0].tolist() == want[t][0].tolist()
        finally:
            V.set_timeout_callback(None)
            del cb
    finally:
        idx.free()


def test_a_timeout_that_fires_while_the_call_is_queued(knobs):
    """Round 4: a call parked behind a pass it is not part of polls ITS OWN timeout callback on ITS OWN
    thread -- on arrival, then every millisecond -- and leaves the queue with VecSim_QueryReply_TimedOut instead of waiting
    for that pass and riding in the next one; the callback is never invoked from another caller's thread; calls without a
    deadline get their serial answers.  The deadlines here expire AFTER the arrival poll: the call is in the queue (or in
    a pass) when it finds out."""
    knobs("coalesce_min_mib", 0)
    n, dim = 8_000_000, 256                       # ~1.5 ms per pass: queued calls get to poll while it runs
    idx = _index(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2, n)
    try:
        qs = _queries(V.VecSimType_FLOAT32, dim, 16)
        want = [idx.topk_query(q, 10).results() for q in qs]
        calls = []                                 # (ctx, thread id) of every callback invocation
        polls = {}                                 # ctx -> invocations since the owner's current call began

        def on_timeout(ctx):
            ctx = int(ctx or 0)
            calls.append((ctx, threading.get_ident()))
            if not ctx:
                return 0
            polls[ctx] = polls.get(ctx, 0) + 1
            return 1 if polls[ctx] > 1 else 0      # fine on arrival, expired at the next look
        cb = V.set_timeout_callback(on_timeout)
        try:
            n_threads, reps = 12, 20
            owner = {}
            out = [[] for _ in range(n_threads)]
            bar = threading.Barrier(n_threads)

            def caller(t):
                owner[t + 1] = threading.get_ident()
                qp = V.VecSimQueryParams()
                qp.timeoutCtx = t + 1 if t % 3 == 0 else None      # every third caller carries a deadline
                bar.wait()
                for r in range(reps):
                    polls[t + 1] = 0
                    rep = idx.topk_query(qs[(t + r) % len(qs)], 10, params=qp)
                    out[t].append((rep.code, rep.results()[0].tolist(), (t + r) % len(qs)))
            V.coalesce_stats(reset=True)
            th = [threading.Thread(target=caller, args=(t,)) for t in range(n_threads)]
            [x.start() for x in th]
            [x.join() for x in th]
            st = V.coalesce_stats()
            for t in range(n_threads):
                for code, ids, qi in out[t]:
                    if t % 3 == 0:
                        assert code == V.VecSim_QueryReply_TimedOut and ids == []
                    else:
                        assert code == V.VecSim_QueryReply_OK and ids == want[qi][0].tolist()
            # some of the expired calls were queued behind a pass when they gave up (others found out as they were about to
            # lead, or inside a pass that had already taken them: those still rode in it)
            assert st["left_queue_on_timeout"] > 0, st
            assert st["queries"] >= (n_threads - n_threads // 3) * reps, st
            # the callback ran on the thread that owns the context, never on a leader's or a worker's
            assert calls and all(tid == owner[ctx] for ctx, tid in calls if ctx), [c for c in calls if c[0] and c[1] != owner[c[0]]][:3]
        finally:
            V.set_timeout_callback(None)
            del cb
    finally:
        idx.free()
