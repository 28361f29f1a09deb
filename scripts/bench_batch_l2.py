#!/usr/bin/env python3
"""Batched 10 M x 768 fp16 top-100, 256 queries per corpus pass on the matrix cores: the L2 form (round 3) next to the IP
form on the same rows.  One JSON object on stdout.  GPU only."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (device runtime first)
import bench as B
from redisearch_amd import vecsim as V

rows = int(os.environ.get("ROWS", 10_000_000))
dim, k, batch, reps = 768, 100, 256, 10
lib = V.load()
out = {"rows": rows, "dim": dim, "k": k, "batch": batch}
for name, metric in (("ip", V.VecSimMetric_IP), ("l2", V.VecSimMetric_L2)):
    idx = V.VecSimIndex(V.VecSimType_FLOAT16, dim, metric)
    idx.reserve(rows)
    idx.add_philox_rows(B.SEED, 0, rows, 1)
    qs = B.philox_host_rows(V, B.QUERY_BASE, batch * 4, dim, V.VecSimType_FLOAT16).reshape(4, batch, dim)
    t0 = time.perf_counter()
    idx.topk_batch(qs[0], k)
    first = time.perf_counter() - t0
    lib.RSGPU_ResetProfile()
    lib.RSGPU_SetProfiling(1)
    t0 = time.perf_counter()
    for i in range(reps):
        ids, sc, cnt = idx.topk_batch(qs[(i + 1) % 4], k)
    el = time.perf_counter() - t0
    lib.RSGPU_SetProfiling(0)
    launches, ms, _ = V.scan_profile()
    same = True
    for i in (0, 85, 170, 255):
        si, ss = idx.topk_query(qs[reps % 4][i], k).results()
        same &= si.tolist() == ids[i].tolist() and ss.tolist() == sc[i].tolist()
    d = ms / max(launches, 1)
    out[name] = {"device_ms_per_pass": d, "launches": launches, "qps_device": batch / d * 1e3, "qps_wall": reps * batch / el,
                 "first_call_s": first, "bit_identical_to_single_queries": bool(same),
                 "tflops": 2.0 * batch * dim * rows / d / 1e9}
    idx.free()
print(json.dumps(out))
