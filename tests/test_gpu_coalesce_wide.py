"""The coalescer's WIDE passes (round 4): more than sixteen VecSimIndex_TopKQuery calls queued on an index whose batched
queries are exact -- here plain FLOAT32 cosine / L2 indexes, no shadow -- share ONE matrix-core filter pass over the corpus
(gemm_qs_f32_kernel) + exact re-scoring, up to 256 calls per pass (FlatIndex::topk_pass_wide).  The contract is the
coalescer's: every reply BIT-IDENTICAL (ids and scores) to the same query issued alone -- with different K and reply orders
inside one pass, with tied distances cut by K (pairs of equal rows) and with rows out of label order (deletes)."""
import threading

import numpy as np
import pytest
import torch

from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu
F32 = V.VecSimType_FLOAT32


def _hammer(idx, qs, ks, orders, want, n_threads, reps):
    errors, barrier = [], threading.Barrier(n_threads)

    def worker(t):
        try:
            barrier.wait()
            for _ in range(reps):
                for i in range(t, len(qs), n_threads):
                    rep = idx.topk_query(qs[i], ks[i], order=orders[i])
                    assert rep.code == V.VecSim_QueryReply_OK
                    ids, sc = rep.results()
                    wi, ws = want[i]
                    if orders[i] == V.BY_ID:
                        o = np.argsort(wi, kind="stable")
                        wi, ws = wi[o], ws[o]
                    assert ids.tolist() == wi.tolist() and sc.tolist() == ws.tolist(), (i, ks[i])
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    [x.start() for x in th]
    [x.join() for x in th]
    return errors


@pytest.mark.parametrize("dim,metric,n", [(128, V.VecSimMetric_Cosine, 700_000), (256, V.VecSimMetric_L2, 560_000)])
def test_forty_callers_share_wide_passes_and_get_their_serial_answers(dim, metric, n):
    lib = V.load()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(dim)
    x = torch.rand((n, dim), device=dev, generator=gen) * 2 - 1
    x[1::2] = x[0:n - (n % 2):2][: x[1::2].shape[0]]          # pairs of equal rows: every distance is tied once
    idx = V.VecSimIndex(F32, dim, metric)
    torch.cuda.synchronize()
    idx.add_device_rows(x.data_ptr(), n, 1)
    try:
        rng = np.random.default_rng(dim + 1)
        for lab in rng.choice(n, 200, replace=False):          # rows no longer in label order
            idx.delete_vector(int(lab) + 1)
        nq = 120
        qs = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
        ks = [(1, 9, 10, 33, 100, 10)[i % 6] for i in range(nq)]
        orders = [V.BY_ID if i % 5 == 0 else V.BY_SCORE for i in range(nq)]
        lib.RSGPU_SetTuning(b"coalesce", 0)
        want = [idx.topk_query(q, k).results() for q, k in zip(qs, ks)]
        lib.RSGPU_SetTuning(b"coalesce", 1)
        V.coalesce_stats(reset=True)
        errors = _hammer(idx, qs, ks, orders, want, 40, 4)
        assert not errors, errors[:3]
        st = V.coalesce_stats()
        assert st["queries"] == nq * 4
        assert st["wide_passes"] > 0 and st["wide_queries"] > 16 * st["wide_passes"], st
        # the knob: sixteen per pass again
        lib.RSGPU_SetTuning(b"coalesce_wide", 0)
        V.coalesce_stats(reset=True)
        errors = _hammer(idx, qs, ks, orders, want, 40, 1)
        assert not errors, errors[:3]
        st = V.coalesce_stats()
        assert st["wide_passes"] == 0 and st["mq_passes"] > 0, st
    finally:
        lib.RSGPU_SetTuning(b"coalesce", 1)
        lib.RSGPU_SetTuning(b"coalesce_wide", 1)
        idx.free()


@pytest.mark.parametrize("vtype,metric", [(V.VecSimType_FLOAT16, V.VecSimMetric_Cosine), (V.VecSimType_BFLOAT16, V.VecSimMetric_IP),
                                          (V.VecSimType_FLOAT16, V.VecSimMetric_IP)])
def test_sixteen_bit_ip_and_cosine_indexes_take_wide_passes_too(vtype, metric):
    """round 5: the FLOAT16 / BFLOAT16 IP / cosine batched route (BASELINE configs[2]) re-scores its survivors exactly, so
    wide_pass_capable admits these indexes: forty callers share matrix-core passes and every reply is bit-identical to the
    query issued alone"""
    import oracle as O
    lib = V.load()
    n, dim = 1_500_000, 256                                 # (rows of 512 bytes: the multi-query scan's smallest shape)
    idx = V.VecSimIndex(vtype, dim, metric)
    idx.add_philox_rows(31, 0, n, 1)
    t = O.F16 if vtype == V.VecSimType_FLOAT16 else O.BF16
    try:
        rng = np.random.default_rng(3)
        for lab in rng.choice(n, 100, replace=False):
            idx.delete_vector(int(lab) + 1)
        nq = 96
        qs = O.philox_rows(31, 1 << 40, nq, dim, t)
        if t == O.BF16:
            qs = (qs.astype(np.uint32) << 16).view(np.float32)
        ks = [(1, 10, 33, 100)[i % 4] for i in range(nq)]
        orders = [V.BY_SCORE] * nq
        lib.RSGPU_SetTuning(b"coalesce", 0)
        want = [idx.topk_query(q, k).results() for q, k in zip(qs, ks)]
        lib.RSGPU_SetTuning(b"coalesce", 1)
        # (a 16-bit index switches to the wide pass only when MORE than sixteen calls are queued at once -- a matter of timing on a
        # corpus this small: several rounds of 64 callers, every reply checked in every round)
        wide = 0
        for attempt in range(6):
            V.coalesce_stats(reset=True)
            errors = _hammer(idx, qs, ks, orders, want, 64, 3)
            assert not errors, errors[:3]
            wide += V.coalesce_stats()["wide_passes"]
            if wide:
                break
        assert wide > 0, V.coalesce_stats()
        # ... and the batch entry point itself: bit-identical to the single queries
        ids, sc, cnt = idx.topk_batch(qs, 100)
        for i in range(0, nq, 7):
            wi, ws = idx.topk_query(qs[i], 100).results()
            assert ids[i].tolist() == wi.tolist() and np.array_equal(sc[i], ws), i
    finally:
        lib.RSGPU_SetTuning(b"coalesce", 1)
        idx.free()
