"""FLOAT32 batched queries on the matrix cores WITHOUT a stored shadow (round 4, gemm_qs_f32_kernel): 10M x 768 fp32 FLAT,
256 queries per corpus pass through RSGPU_FlatIndex_TopKBatch on a plain index.  Per metric and kernel shape (knob
gemm_qs_f32: 1 = eight waves x 32 queries, 2 = four waves x 64): HIP-event time of the whole device pipeline per pass,
HBM GB/s of the fp32 rows, MFMA TFLOP/s, and a bit-identity spot check against VecSimIndex_TopKQuery."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redisearch_amd import vecsim as V  # noqa: E402

rows = int(os.environ.get("ROWS", 10_000_000))
dim, batch = int(os.environ.get("DIM", 768)), 256
k = int(os.environ.get("K", 100))
reps = int(os.environ.get("REPS", 8))
torch.cuda.set_device(0)
lib = V.load()
for kv in filter(None, os.environ.get("TUNING", "").split(",")):   # TUNING="gemm_qs_f8=1": any engine knob (read at index creation too)
    key, val = kv.split("=")
    assert lib.RSGPU_SetTuning(key.encode(), int(val)) == 0, kv
out = {"rows": rows, "dim": dim, "k": k, "tuning": os.environ.get("TUNING", "")}
for mname, metric in (("cosine", V.VecSimMetric_Cosine), ("l2", V.VecSimMetric_L2)):
    if os.environ.get("METRICS") and mname not in os.environ["METRICS"]:
        continue
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, dim, metric)
    idx.reserve(rows)
    idx.add_philox_rows(47, 0, rows, 1)
    qs = np.random.default_rng(48).uniform(-1, 1, (4, batch, dim)).astype(np.float32)
    for shape in [int(x) for x in os.environ.get("SHAPES", "1,2").split(",")]:
        lib.RSGPU_SetTuning(b"gemm_qs_f32", shape)
        t0 = time.perf_counter()
        idx.topk_batch(qs[0], k)  # warm-up (allocations; L2: the half norms)
        first = time.perf_counter() - t0
        lib.RSGPU_ResetProfile()
        lib.RSGPU_SetProfiling(1)
        before = V.coalesce_stats()["mq_passes"]
        t0 = time.perf_counter()
        for i in range(reps):
            ids, sc, cnt = idx.topk_batch(qs[(i + 1) % 4], k)
        el = time.perf_counter() - t0
        lib.RSGPU_SetProfiling(0)
        launches, ms, by = V.scan_profile()
        mq = V.coalesce_stats()["mq_passes"] - before
        dev_ms = ms / max(launches, 1)
        same = True
        for i in (0, 85, 170, 255):
            si, ss = idx.topk_query(qs[reps % 4][i], k).results()
            same &= si.tolist() == ids[i].tolist() and ss.tolist() == sc[i].tolist()
        flops = 2.0 * batch * dim * rows
        out["%s_shape%d" % (mname, shape)] = {
            "device_ms_per_pass": dev_ms, "qps_device": batch / dev_ms * 1e3, "qps_wall": reps * batch / el, "launches": launches,
            "mq_scan_passes": mq, "hbm_gbs_of_fp32_rows": rows * dim * 4 / dev_ms / 1e6, "hbm_frac": rows * dim * 4 / dev_ms / 1e6 / 8000,
            "mfma_tflops": flops / dev_ms / 1e9, "first_call_s": first, "bit_identical_to_single_queries": bool(same)}
        print(mname, shape, json.dumps(out["%s_shape%d" % (mname, shape)]), flush=True)
    lib.RSGPU_SetTuning(b"gemm_qs_f32", 2)
    idx.free()
    lib.RSGPU_ReleaseWorkspaces()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/batch_f32%s.json" % os.environ.get("OUT_TAG", ""), "w"), indent=1)
