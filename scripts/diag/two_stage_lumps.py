#!/usr/bin/env python3
"""Where do the occasional 30-100 ms lumps in a two-stage query stream come from?  Runs ~15 s of back-to-back two-stage
queries (and, as a control, ~5 s of plain fp32 scans) and prints every query slower than 4x the median with its wall-clock
offset: a lump that recurs with a fixed period and hits both streams alike is something outside the library (a monitor
polling the GPU), not a path inside it."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from redisearch_amd import vecsim as V  # noqa: E402

lib = V.load()
rows, dim, k = 10_000_000, 768, 10
lib.RSGPU_SetTuning(b"shadow8", 1)
idx = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_Cosine)
lib.RSGPU_SetTuning(b"shadow8", 0)
idx.reserve(rows)
idx.add_philox_rows(47, 0, rows, 1)
s = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2)
s.add_philox_rows(47, 1 << 40, 256, 1)
qs = s.read_rows(0, 256)
s.free()
out = {}
for name, two, secs in (("two_stage", 1, 15.0), ("fp32_scan", 0, 6.0)):
    lib.RSGPU_SetTuning(b"two_stage", two)
    for i in range(10):
        idx.topk_query(qs[i], k)
    V.two_stage_stats(reset=True)
    lat, at = [], []
    t0 = time.perf_counter()
    i = 0
    while time.perf_counter() - t0 < secs:
        q = qs[i % 256]
        a = time.perf_counter()
        rep = lib.VecSimIndex_TopKQuery(idx.ptr, q.ctypes.data_as(C.c_void_p), k, None, V.BY_SCORE)
        lib.VecSimQueryReply_Free(rep)
        b = time.perf_counter()
        lat.append(b - a)
        at.append(a - t0)
        i += 1
    lat = np.array(lat)
    med = float(np.median(lat))
    lumps = [(round(at[j], 3), round(lat[j] * 1e3, 2)) for j in np.flatnonzero(lat > 4 * med)]
    out[name] = {"queries": len(lat), "median_ms": med * 1e3, "p99_ms": float(np.percentile(lat, 99) * 1e3), "max_ms": float(lat.max() * 1e3),
                 "qps": len(lat) / (at[-1] + lat[-1]), "lumps_at_s_ms": lumps[:40], "n_lumps": len(lumps),
                 "two_stage_stats": V.two_stage_stats() if two else None}
print(json.dumps(out, indent=1))
