"""Sweeps the scan kernel's launch parameters on the BASELINE config (10M x 768 fp32) and prints the
HIP-event kernel time / achieved GB/s per variant. GPU only; writes gpurun_out/tune_scan.json."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redisearch_amd import vecsim as V  # noqa: E402

rows = int(os.environ.get("ROWS", 10_000_000))
dim = int(os.environ.get("DIM", 768))
lib = V.load()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
idx = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_Cosine)
idx.reserve(rows)
gen = torch.Generator(device=dev)
gen.manual_seed(47)
done = 0
while done < rows:
    m = min(1_000_000, rows - done)
    t = torch.rand((m, dim), device=dev, generator=gen).mul_(2).sub_(1)
    torch.cuda.synchronize()
    idx.add_device_rows(t.data_ptr(), m, done + 1)
    done += m
    del t
qs = np.random.default_rng(48).uniform(-1, 1, (64, dim)).astype(np.float32)


def run(n=24):
    lib.RSGPU_ResetProfile()
    lib.RSGPU_SetProfiling(1)
    for i in range(n):
        r = lib.VecSimIndex_TopKQuery(idx.ptr, qs[i % 64].ctypes.data_as(C.c_void_p), 10, None, 0)
        lib.VecSimQueryReply_Free(r)
    lib.RSGPU_SetProfiling(0)
    k, ms, by = V.scan_profile()
    return ms / k, by / k / (ms / k / 1e3) / 1e9


res = []
run(8)
for nt in (1, 0):
    for u in (4, 2, 8):
        for bpc in (8, 4, 6, 12, 16, 2):
            lib.RSGPU_SetTuning(b"nontemporal", nt)
            lib.RSGPU_SetTuning(b"rows_per_group", u)
            lib.RSGPU_SetTuning(b"blocks_per_cu", bpc)
            run(4)
            ms, gbs = run()
            res.append(dict(nt=nt, rows_per_group=u, blocks_per_cu=bpc, kernel_ms=ms, gbs=gbs))
            print(res[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/tune_scan.json", "w"), indent=1)
best = max(res, key=lambda r: r["gbs"])
print("BEST", best)
