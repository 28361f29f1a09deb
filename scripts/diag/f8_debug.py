import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from redisearch_amd import vecsim as V
lib = V.load()
lib.RSGPU_SetTuning(b"gemm_qs_f8", int(os.environ.get("F8", 1)))
dim, n, k = int(os.environ.get("DIM", 256)), int(os.environ.get("N", 700000)), 10
g = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_Cosine)
g.add_philox_rows(5, 0, n, 1)
q = np.random.default_rng(1).uniform(-1, 1, (300, dim)).astype(np.float32)
want = [g.topk_query(v, k).results() for v in q[:8]]
print("singles done", flush=True)
ids, sc, cnt = g.topk_batch(q, k)
print("batch done", flush=True)
for i, (wi, ws) in enumerate(want):
    print(i, ids[i].tolist() == wi.tolist(), sc[i].tolist() == ws.tolist(), len(set(ids[i].tolist()) & set(wi.tolist())), cnt[i])
