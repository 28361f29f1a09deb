"""GPU, >= 2 DEVICES: the multi-device forms of SURVEY.md 8(e) on distinct physical devices.  Nothing here runs on a one-GPU
box (every test skips itself); on the 8-GPU node the whole file runs unattended, so that the first hardware run of the
scaling bench is NOT also the first execution of
  * `ncclCommInitAll` over distinct devices and the grouped all-gather of the in-process exchange (shard_comm.cpp),
  * the per-shard `hipSetDevice` workers and the peer-visible pinned reply memory of sharded_index.cpp,
  * `bench.py` under `torch.distributed.run` with one rank per device (RSGPU_ShardComm_*: ncclAllGather + merge kernel).
Reference shape: per-shard top-K -> coordinator heap of K (reference src/module.c:3541-3547).  Every RCCL step runs in a child
process with a deadline: a collective that never completes fails the test instead of hanging the suite."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_DEV = torch.cuda.device_count() if torch.cuda.is_available() else 0
needs_two = pytest.mark.skipif(N_DEV < 2, reason="needs >= 2 visible devices (%d here)" % N_DEV)


def _child(code, timeout, env=None):
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.update(env or {})
    try:
        p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=e)
    except subprocess.TimeoutExpired:
        pytest.fail("child did not finish within %d s" % timeout)
    return p


# (i) ShardedIndex over DISTINCT devices: RCCL exchange == host merge == unsharded == oracle; deletes; k > rows per shard
_SHARDED = r"""
import ctypes as C, math, sys
import numpy as np, torch
import oracle as O
from redisearch_amd import vecsim as V
lib = V.load()
import os
DRY = os.environ.get("RSGPU_TEST_ONE_DEVICE_DRY_RUN") == "1"    # every shard on device 0, host merge only (see the last test)
nd = torch.cuda.device_count()
assert nd >= 2 or DRY
F32 = V.VecSimType_FLOAT32
for shards in sorted({2, 3 if DRY else nd}):
    for metric, om in ((V.VecSimMetric_L2, O.L2), (V.VecSimMetric_Cosine, O.COSINE)):
        n, dim, seed = 24_000, 64, 31
        per = n // shards
        s = V.ShardedIndex(F32, dim, metric, shards, devices=[0] * shards if DRY else list(range(shards)))
        assert DRY or sorted(s.shard_device(i) for i in range(shards)) == list(range(shards))   # one device per shard
        for i in range(shards):
            cnt = per if i < shards - 1 else n - per * (shards - 1)
            assert s.shard(i).add_philox_rows(seed, i * per, cnt, 1 + i * per) == cnt
        one = V.VecSimIndex(F32, dim, metric)
        one.add_philox_rows(seed, 0, n, 1)
        o = O.FlatIndex(O.F32, dim, om)
        o.add_bulk(O.philox_rows(seed, 0, n, dim), 1)
        qs = [O.philox_rows(seed, n + qi, 1, dim)[0] for qi in range(5)]

        def check(tag):
            for q in qs:
                for k in (1, 10, 100, per + 7):         # the last: more than one shard's share
                    answers = {}
                    for ex in ((0, 0) if DRY else (0, 1)):   # host merge, then ncclCommInitAll + grouped all-gather + merge kernel
                        lib.RSGPU_SetTuning(b"shard_exchange", ex)
                        answers[ex] = s.topk_query(q, k).results()
                    lib.RSGPU_SetTuning(b"shard_exchange", 0)
                    answers.setdefault(1, answers[0])
                    ui, us = one.topk_query(q, k).results()
                    oi, os_ = o.topk(q, k)
                    assert answers[0][0].tolist() == answers[1][0].tolist() == ui.tolist(), (tag, shards, k)
                    if k <= 100:      # (twelve thousand deep, near-ties may order differently in the oracle's arithmetic)
                        assert ui.tolist() == oi.tolist(), (tag, shards, k)
                    if k <= 100:   # (k beyond a shard's rows: ids as above; tests/test_gpu_sharded.py holds that case to ids as well)
                        assert answers[0][1].tolist() == us.tolist(), (tag, shards, k)
                        # the exchange carries fp32 scores (k * 16 B per rank): equal after the same narrowing
                        assert answers[1][1].astype(np.float32).tolist() == us.astype(np.float32).tolist(), (tag, shards, k)
                    assert np.allclose(answers[0][1], us, rtol=1e-6, atol=1e-6) and np.allclose(answers[1][1], us, rtol=1e-6, atol=1e-6)
                    if k <= 100:
                        assert np.allclose(us, os_, rtol=1e-5, atol=1e-4)
        check("pristine")
        st = (C.c_uint64 * 3)()
        lib.RSGPU_ShardedIndex_GetRcclStats(s.ptr, st, 0)
        assert DRY or (st[2] == shards and st[0] > 0), list(st)   # the communicator spans one rank per shard
        # deletes on every shard, re-adds under new labels
        rng = np.random.default_rng(shards)
        dead = rng.choice(np.arange(1, n + 1), 400, replace=False)
        for lab in dead.tolist():
            assert s.delete_vector(lab) == 1 and one.delete_vector(lab) == 1 and o.delete(lab) == 1
        fresh = rng.uniform(-1, 1, (60, dim)).astype(np.float32)
        for j, v in enumerate(fresh):
            s.add_vector(v, n + 1 + j); one.add_vector(v, n + 1 + j); o.add(v, n + 1 + j)
        check("after deletes")
        nq = s.normalized_query(qs[0])
        assert math.isnan(s.get_distance_from_unsafe(int(dead[0]), nq))
        s.free(); one.free()
print("MULTIDEVICE_SHARDED_OK")
"""


@needs_two
def test_sharded_index_over_distinct_devices_rccl_equals_host_merge_equals_unsharded_equals_oracle():
    p = _child(_SHARDED, 600)
    assert p.returncode == 0 and "MULTIDEVICE_SHARDED_OK" in p.stdout, (p.stdout[-600:], p.stderr[-2000:])


# (iii) FLOAT64 rows: 64-bit orderable keys over the exchange (a double does not fit the fp32 score slot)
_F64 = r"""
import ctypes as C
import numpy as np, torch
import oracle as O
from redisearch_amd import vecsim as V
lib = V.load()
import os
DRY = os.environ.get("RSGPU_TEST_ONE_DEVICE_DRY_RUN") == "1"
nd = torch.cuda.device_count()
shards = 3 if DRY else min(nd, 4)
dim, n = 24, 9_000
rng = np.random.default_rng(12)
x = rng.uniform(-1, 1, (n, dim))
x[1::2] = x[0::2] * (1 + 1e-12)        # pairs that differ below fp32 resolution: only 64-bit keys order them
s = V.ShardedIndex(V.VecSimType_FLOAT64, dim, V.VecSimMetric_L2, shards, devices=[0] * shards if DRY else list(range(shards)))
one = V.VecSimIndex(V.VecSimType_FLOAT64, dim, V.VecSimMetric_L2)
o = O.FlatIndex(O.F64, dim, O.L2)
for i in range(n):
    s.add_vector(x[i], i + 1); one.add_vector(x[i], i + 1)
o.add_bulk(x, 1)
for qi in range(6):
    q = rng.uniform(-1, 1, dim)
    for k in (1, 10, 64):
        lib.RSGPU_SetTuning(b"shard_exchange", 0 if DRY else 1)
        gi, gs = s.topk_query(q, k).results()
        lib.RSGPU_SetTuning(b"shard_exchange", 0)
        hi, hs = s.topk_query(q, k).results()
        ui, us = one.topk_query(q, k).results()
        oi, os_ = o.topk(q, k)
        assert gi.tolist() == hi.tolist() == ui.tolist() == oi.tolist(), (qi, k)
        assert gs.tolist() == hs.tolist() == us.tolist(), (qi, k)      # doubles, bit for bit
print("MULTIDEVICE_F64_OK")
"""


@needs_two
def test_float64_keys_over_the_rccl_exchange_on_distinct_devices():
    p = _child(_F64, 300)
    assert p.returncode == 0 and "MULTIDEVICE_F64_OK" in p.stdout, (p.stdout[-600:], p.stderr[-2000:])


def _bench_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, stdout[-800:]
    assert len(lines[0]) < 8192
    return json.loads(lines[0])


# (ii) the driver's multi-GPU command, small: one rank per device under torch.distributed.run
@needs_two
@pytest.mark.parametrize("n", sorted({2, N_DEV}) if N_DEV >= 2 else [2])
def test_bench_under_torch_distributed_run_one_rank_per_device(n):
    port = 29500 + (os.getpid() % 400) + n
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", str(n), "--rows", "200000", "--steps", "5", "--warmup", "2"]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    except subprocess.TimeoutExpired:
        pytest.fail("bench.py --gpus %d under torch.distributed.run did not finish within 900 s" % n)
    assert p.returncode == 0, (p.stdout[-600:], p.stderr[-3000:])
    rec = _bench_line(p.stdout)
    assert rec["n_gpus"] == n and rec["steps"] == 5 and rec["scaling"] == "weak"
    assert rec["config"]["verify"]["ok"] is True, rec["config"]["verify"]
    assert rec["config"]["corpus_rows_total"] == 200000 * n
    assert rec["collective"]["ranks"] == n and rec["collective"]["payload_bytes_per_rank"] == 160
    assert rec["value"] == pytest.approx(n * rec["config"]["global_qps_on_sharded_corpus"], rel=1e-3)   # (rounded in the line)
    assert 0 < rec["roofline"]["frac_min_over_devices"] <= rec["roofline"]["frac_max_over_devices"] < 1.05


# ... and the one-process form (`python bench.py --gpus N`): N device shards behind the plain VecSim handle, RCCL exchange timed
@needs_two
def test_bench_in_one_process_over_distinct_devices():
    n = min(N_DEV, 8)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        p = subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--rows", "200000", "--steps", "5", "--warmup", "2", "--no-extras"],
                           cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    except subprocess.TimeoutExpired:
        pytest.fail("bench.py --gpus %d (one process) did not finish within 900 s" % n)
    assert p.returncode == 0, (p.stdout[-600:], p.stderr[-3000:])
    rec = _bench_line(p.stdout)
    assert rec["n_gpus"] == n and rec["config"]["verify"]["ok"] is True
    assert rec["collective"]["ranks"] == n and rec["collective"]["timed_exchange"] == "rccl", rec["collective"]


@pytest.mark.skipif(N_DEV >= 2, reason="the real thing runs above")
def test_dry_run_of_the_multi_device_scripts_on_one_device():
    """One-GPU boxes: the very scripts of the tests above with every shard on device 0 and the host merge in both arms -- what a
    one-device box can prove about them (API use, oracle calls, the delete / re-add choreography, FLOAT64 tie pairs); the RCCL
    arm itself needs one device per rank."""
    for code, tag in ((_SHARDED, "MULTIDEVICE_SHARDED_OK"), (_F64, "MULTIDEVICE_F64_OK")):
        p = _child(code, 600, {"RSGPU_TEST_ONE_DEVICE_DRY_RUN": "1"})
        assert p.returncode == 0 and tag in p.stdout, (p.stdout[-600:], p.stderr[-2000:])


def test_this_file_arms_itself():
    """On a one-GPU box this is the only test of the file that runs: it pins the skip condition to the device count, so a box
    with two devices cannot silently skip the rest."""
    skipped = N_DEV < 2
    assert needs_two.args[0] is skipped
    if not skipped:
        assert N_DEV >= 2
