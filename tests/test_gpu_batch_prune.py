"""GPU: the small kernels between the matrix-core launches of a batched pass (round 6) -- batch_select_kernel with its list in
registers or re-read from memory (knob batch_select_regs), threshold selects that prune the list to the new bound (knob
batch_prune; never the L2 lists, whose keys are upper bounds), compact_cand_kernel's parallel concatenation, batch_rescore_kernel
dealing a pruned list out one candidate per group.  Every combination of the knobs: replies BIT-IDENTICAL to
VecSimIndex_TopKQuery (and so to each other), also where a query's list is longer than the registers hold (20 000 copies of one
row: the in-place, chunked form of the pruning) and where it overflows (the host redoes the query).
Reference: B x VecSimIndex_TopKQuery, src/iterators/hybrid_reader.c:374."""
import numpy as np
import pytest

from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu
F32, F16, IP, COS, L2 = V.VecSimType_FLOAT32, V.VecSimType_FLOAT16, V.VecSimMetric_IP, V.VecSimMetric_Cosine, V.VecSimMetric_L2
KNOBS = [(1, 1), (1, 0), (0, 1), (0, 0)]  # (batch_prune, batch_select_regs)


@pytest.fixture
def lib():
    lb = V.load()
    yield lb
    lb.RSGPU_SetTuning(b"batch_prune", 1)
    lb.RSGPU_SetTuning(b"batch_select_regs", 1)


def batched(lib, g, queries, k, prune, regs):
    assert lib.RSGPU_SetTuning(b"batch_prune", prune) == 0 and lib.RSGPU_SetTuning(b"batch_select_regs", regs) == 0
    lib.RSGPU_ResetProfile()
    lib.RSGPU_SetProfiling(1)
    out = g.topk_batch(queries, k)
    lib.RSGPU_SetProfiling(0)
    launches, _, _ = V.scan_profile()
    return out, launches


def same_as_single(g, queries, k, out):
    ids, sc, cnt = out
    for i, q in enumerate(queries):
        wi, ws = g.topk_query(q, k).results()
        assert cnt[i] == len(wi), i
        assert ids[i][: cnt[i]].tolist() == wi.tolist(), i
        assert np.array_equal(sc[i][: cnt[i]], ws, equal_nan=True), i


@pytest.mark.parametrize("vtype,metric,dim,n", [(F32, COS, 768, 530_001), (F32, IP, 256, 700_000), (F16, IP, 768, 540_000),
                                                (F32, L2, 256, 600_000)])
@pytest.mark.parametrize("k", [10, 100])
def test_every_knob_combination_is_bit_identical_to_single_queries(lib, vtype, metric, dim, n, k):
    g = V.VecSimIndex(vtype, dim, metric)
    assert g.add_philox_rows(dim + k, 0, n, 1) == n
    queries = np.random.default_rng(dim + 7 * k).uniform(-1, 1, (300, dim)).astype(np.float16 if vtype == F16 else np.float32)
    first = None
    for prune, regs in KNOBS:
        out, launches = batched(lib, g, queries, k, prune, regs)
        assert launches == 2, "the batched path was not taken (%d profiled launches)" % launches
        if first is None:
            same_as_single(g, queries, k, out)
            first = out
        else:
            assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out, first)), (prune, regs)
    g.free()


@pytest.mark.parametrize("vtype,metric", [(F32, COS), (F16, IP)])
@pytest.mark.parametrize("copies,spread", [(3_000, True), (40_000, False)])
def test_long_lists_and_lists_that_overflow(lib, vtype, metric, copies, spread):
    """`copies` identical rows that are every nearby query's nearest: each of them is inside every bound, so a query's list holds
    them all.  3 000 of them spread through the corpus: lists thirty times the usual length, no overflow (one batched launch).
    40 000 appended at the end: above the list's room (32 Ki) -- the select flags the query and the host redoes it on the
    single-query path.  Ties are broken by row.  (The path lists above the 16 Ki entries the registers hold take is the one
    batch_select_regs = 0 forces: the test above.)"""
    dim, n, k = 256, 560_000, 10
    npdt = np.float16 if vtype == F16 else np.float32
    rng = np.random.default_rng(copies)
    v = rng.uniform(-1, 1, dim).astype(npdt)
    g = V.VecSimIndex(vtype, dim, metric)
    if spread:
        mat = rng.uniform(-1, 1, (n, dim)).astype(npdt)
        where = np.arange(0, n, n // copies)[:copies]
        mat[where] = v
        g.add_bulk(mat, first_label=1)
        answer = set((where + 1).tolist())
    else:
        assert g.add_philox_rows(31, 0, n, 1) == n
        g.add_bulk(np.tile(v, (copies, 1)), first_label=n + 1)
        answer = set(range(n + 1, n + copies + 1))
    # queries near v (the copies win) and a few unrelated ones
    queries = np.concatenate([(v.astype(np.float32) + rng.normal(0, 0.05, (24, dim))).astype(npdt),
                              rng.uniform(-1, 1, (8, dim)).astype(npdt)])
    first = None
    for prune, regs in KNOBS:
        out, launches = batched(lib, g, queries, k, prune, regs)
        assert (launches == 1) if spread else (launches > 1), launches
        if first is None:
            same_as_single(g, queries, k, out)
            first = out
            assert set(out[0][0][:k].tolist()) <= answer  # (the copies are the answer)
        else:
            assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out, first)), (prune, regs)
    g.free()
