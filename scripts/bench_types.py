"""Scan-kernel throughput for every FLAT element type and metric (4M x 768, top-10, 100 queries each) through
VecSimIndex_TopKQuery, HIP-event time of the scan kernel -> algorithmic GB/s.  Writes gpurun_out/types_bench.json."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redisearch_amd import vecsim as V  # noqa: E402

rows, dim, k = int(os.environ.get("ROWS", 4_000_000)), 768, 10
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = V.load()
TYPES = [("FLOAT32", V.VecSimType_FLOAT32, torch.float32, 4), ("FLOAT64", V.VecSimType_FLOAT64, torch.float64, 8),
         ("FLOAT16", V.VecSimType_FLOAT16, torch.float16, 2), ("BFLOAT16", V.VecSimType_BFLOAT16, torch.bfloat16, 2),
         ("INT8", V.VecSimType_INT8, torch.int8, 1), ("UINT8", V.VecSimType_UINT8, torch.uint8, 1)]
METRICS = [("L2", V.VecSimMetric_L2), ("IP", V.VecSimMetric_IP), ("COSINE", V.VecSimMetric_Cosine)]
only = os.environ.get("TYPES_ONLY")
if only:
    TYPES = [t for t in TYPES if t[0] in only.split(",")]
rpgs = [int(x) for x in os.environ.get("RPG", "0").split(",")]   # rows_per_group knob values to sweep (0 = default)
out = []
for tname, vt, tdt, esz in TYPES:
    gen = torch.Generator(device=dev)
    gen.manual_seed(47)
    if tdt in (torch.int8, torch.uint8):
        lo, hi = (-127, 128) if tdt == torch.int8 else (0, 256)
        x = torch.randint(lo, hi, (rows, dim), device=dev, generator=gen, dtype=torch.int32).to(tdt)
        qs = np.random.default_rng(48).integers(lo, hi, (100, dim)).astype(np.int8 if tdt == torch.int8 else np.uint8)
    else:
        x = (torch.rand((rows, dim), device=dev, generator=gen, dtype=torch.float32) * 2 - 1).to(tdt)
        qs = np.random.default_rng(48).uniform(-1, 1, (100, dim)).astype(np.float64 if tdt == torch.float64 else np.float32)
    for mname, m, rpg in [(a, b, r) for a, b in METRICS for r in rpgs]:
        lib.RSGPU_SetTuning(b"rows_per_group", rpg)
        idx = V.VecSimIndex(vt, dim, m)
        torch.cuda.synchronize()
        idx.add_device_rows(x.data_ptr(), rows, 1)
        for q in qs[:5]:
            idx.topk_query(q, k)
        lib.RSGPU_ResetProfile()
        lib.RSGPU_SetProfiling(1)
        t0 = time.perf_counter()
        for q in qs:
            idx.topk_query(q, k)
        el = time.perf_counter() - t0
        lib.RSGPU_SetProfiling(0)
        launches, ms, by = V.scan_profile()
        r = {"type": tname, "metric": mname, "rows_per_group": rpg, "rows": rows, "dim": dim, "qps": len(qs) / el, "scan_kernel_ms": ms / launches,
             "scan_gbs": by / launches / (ms / launches) / 1e6, "frac_of_8TBs": by / launches / (ms / launches) / 1e6 / 8000}
        out.append(r)
        print(json.dumps(r), flush=True)
        del idx
    del x
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/types_bench.json", "w"), indent=1)
