"""The exchange over RCCL, in C (round 4; redisearch_amd/csrc/shard_comm.cpp, exchange_kernels.hip): ncclAllGather of the
per-shard top-k + a merge kernel on every rank, reachable (a) rank per process through RSGPU_ShardComm_* and (b) behind a
sharded VecSim handle with the "shard_exchange" knob.  A one-GPU box can only form a ONE-rank communicator, so the
communicator tests run with world = 1 (the collective degenerates, every other step is the N-rank code); the merge kernel
itself is held to the host merge on many-rank inputs -- ties, padding, NaN, fewer candidates than k -- fed directly."""
import ctypes as C

import numpy as np
import pytest
import torch

from redisearch_amd import sharded as SH
from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu
F32 = V.VecSimType_FLOAT32


def _index(n, dim, metric, seed=3):
    idx = V.VecSimIndex(F32, dim, metric)
    assert idx.add_philox_rows(seed, 0, n, 1) == n
    return idx


@pytest.mark.parametrize("world,k", [(2, 10), (8, 10), (8, 100), (5, 1), (64, 100), (3, 1000)])
def test_merge_kernel_equals_the_host_merge_on_many_rank_inputs(world, k):
    lib = V.load()
    rng = np.random.default_rng(world * 1000 + k)
    n = world * k
    scores = rng.integers(0, 40, n).astype(np.float32) / 8.0            # many ties
    scores[rng.integers(0, n, max(n // 50, 1))] = np.nan               # NaN distances rank last
    labels = rng.permutation(10 * n)[:n].astype(np.uint64) + 1
    pad = rng.random(n) < 0.3                                          # shards with fewer than k rows pad their slots
    labels[pad] = np.uint64(0xFFFFFFFFFFFFFFFF)
    hs, hl = np.zeros(k, np.float64), np.zeros(k, np.uint64)
    ds, dl = np.zeros(k, np.float64), np.zeros(k, np.uint64)
    m = lib.RSGPU_MergeTopKHost(scores.ctypes.data_as(C.c_void_p), labels.ctypes.data_as(C.c_void_p), n, k,
                                hs.ctypes.data_as(C.c_void_p), hl.ctypes.data_as(C.c_void_p))
    g = lib.RSGPU_MergeTopKDevice(0, scores.ctypes.data_as(C.c_void_p), labels.ctypes.data_as(C.c_void_p), n, k,
                                  ds.ctypes.data_as(C.c_void_p), dl.ctypes.data_as(C.c_void_p))
    assert g == m, V.last_error()
    assert dl[:g].tolist() == hl[:m].tolist()
    assert np.array_equal(ds[:g], hs[:m], equal_nan=True)


@pytest.mark.parametrize("metric", [V.VecSimMetric_Cosine, V.VecSimMetric_L2])
def test_one_rank_communicator_returns_the_index_answers(metric):
    idx = _index(60_000, 64, metric)
    try:
        dev = torch.device("cuda", 0)
        qs = np.random.default_rng(5).uniform(-1, 1, (12, 64)).astype(np.float32)
        for k in (1, 10, 100):
            sc = SH.ShardComm(idx, k, dev)
            try:
                assert sc.world == 1
                for q in qs:
                    labels, scores = sc.query(q)
                    wi, ws = idx.topk_query(q, k).results()
                    assert labels.tolist() == wi.tolist()
                    assert scores.tolist() == ws.astype(np.float32).astype(np.float64).tolist()
                n, ns = sc.stats()
                assert n == len(qs) and ns > 0
            finally:
                sc.free()
    finally:
        idx.free()


_SHARDED_RCCL = r"""
import ctypes as C, numpy as np
from redisearch_amd import vecsim as V
lib = V.load()
F32 = V.VecSimType_FLOAT32
sh = V.ShardedIndex(F32, 48, V.VecSimMetric_L2, 1)
plain = V.VecSimIndex(F32, 48, V.VecSimMetric_L2)
rows = np.random.default_rng(2).uniform(-1, 1, (3000, 48)).astype(np.float32)
for i, r in enumerate(rows):
    sh.add_vector(r, i + 1)
plain.add_bulk(rows)
qs = np.random.default_rng(6).uniform(-1, 1, (9, 48)).astype(np.float32)
lib.RSGPU_SetTuning(b"shard_exchange", 1)
for q in qs:
    for order in (V.BY_SCORE, V.BY_ID):
        gi, gs = sh.topk_query(q, 10, order=order).results()
        wi, ws = plain.topk_query(q, 10, order=order).results()
        assert gi.tolist() == wi.tolist() and gs.tolist() == ws.tolist()
st = (C.c_uint64 * 3)()
lib.RSGPU_ShardedIndex_GetRcclStats(sh.ptr, st, 0)
assert st[0] == 18 and st[2] == 1, list(st)
print("SHARDED_RCCL_OK")
"""


def test_sharded_index_with_the_rccl_exchange():
    """a one-shard RSGPU_ShardedIndex with shard_exchange = 1: every top-k goes through the (one-rank) communicator its
    worker created.  In a child process with a deadline: a collective that never completes must fail this test, not hang
    the suite."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        p = subprocess.run([sys.executable, "-c", _SHARDED_RCCL], cwd=root, capture_output=True, text=True, timeout=75)
    except subprocess.TimeoutExpired:
        pytest.fail("the in-process RCCL exchange did not finish within 75 s")
    assert p.returncode == 0 and "SHARDED_RCCL_OK" in p.stdout, (p.stdout[-400:], p.stderr[-1200:])


def test_two_shards_on_one_device_cannot_form_a_communicator():
    """RCCL wants one device per rank: the knob fails loudly (a NULL reply + RSGPU_LastError), the host merge still works"""
    lib = V.load()
    lib.RSGPU_SetTuning(b"shards", 2)
    try:
        sh = V.VecSimIndex(F32, 32, V.VecSimMetric_L2)
    finally:
        lib.RSGPU_SetTuning(b"shards", 0)
    try:
        assert sh.add_philox_rows(3, 0, 5_000, 1) == 5_000
        q = np.random.default_rng(7).uniform(-1, 1, 32).astype(np.float32)
        want = sh.topk_query(q, 5).results()
        lib.RSGPU_SetTuning(b"shard_exchange", 1)
        rep = lib.VecSimIndex_TopKQuery(sh.ptr, q.ctypes.data_as(C.c_void_p), 5, None, V.BY_SCORE)
        assert not rep and "one device per rank" in V.last_error()
        lib.RSGPU_SetTuning(b"shard_exchange", 0)
        got = sh.topk_query(q, 5).results()
        assert got[0].tolist() == want[0].tolist()
    finally:
        lib.RSGPU_SetTuning(b"shard_exchange", 0)
        sh.free()
