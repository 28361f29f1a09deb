"""BASELINE configs[4] through RSGPU_HybridQuery in a loop (the oracle encodes the posting lists; for rocprofv3 --kernel-trace: per-kernel times and the gaps
between them).  Prints the wall time per query."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import oracle as O  # noqa: E402
from redisearch_amd import search as S  # noqa: E402
from redisearch_amd import vecsim as V  # noqa: E402

for kv in os.environ.get("RSGPU_TUNING", "").split(","):   # e.g. RSGPU_TUNING=hybrid_one_pass=0
    if kv:
        key, val = kv.split("=")
        assert V.load().RSGPU_SetTuning(key.encode(), int(val)) == 0, kv
n_docs, n_vec, dim = int(os.environ.get("N_DOCS", 50_000_000)), int(os.environ.get("N_VEC", 5_000_000)), 768
reps = int(os.environ.get("REPS", 200))
rng = np.random.default_rng(49)
lists = []
for r in (2, 4):
    docs = np.flatnonzero(rng.random(n_docs + 1) < 0.2 / r).astype(np.uint64)
    docs = docs[docs > 0]
    ii = O.InvertedIndex(O.C_FREQS_ONLY)
    ii.add_many(docs, np.minimum(1 + rng.geometric(0.5, docs.size), 255).astype(np.uint32))
    lists.append(ii)
doc_len = (50 + rng.poisson(150, n_docs + 1)).astype(np.uint32)
table = S.DocTable(doc_len, np.ones(n_docs + 1, np.float32))
avg = float(doc_len[1:].mean())
idx = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2)
idx.reserve(n_vec)
idx.add_philox_rows(B.SEED, 0, n_vec, 1)
q = O.philox_rows(B.SEED, B.QUERY_BASE, 1, dim)[0]
g = [S.Postings.from_flat(l.flatten()) for l in lists]
idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists]
bidf = [S.calculate_idf_bm25(n_docs, l.unique_docs) for l in lists]


hq = S.HybridQuery(g, table, "BM25STD", idf, bidf, [1.0, 1.0], n_docs, avg, top_n=10, index=idx, q=q, k=10)


def fused():
    hq.run()            # the bare C call: the argument block is prepared once, as a C caller's would be
    return hq.results()


r0 = fused()
for _ in range(10):
    fused()
walls = []
for _ in range(reps):
    t0 = time.perf_counter()
    hq.run()
    walls.append((time.perf_counter() - t0) * 1e3)
r = hq.results()
assert r["top"][0].tolist() == r0["top"][0].tolist() and r["knn"][0].tolist() == r0["knn"][0].tolist()
w = np.array(walls)
print("HYBRID_FUSED hits %d wall_ms min %.4f p50 %.4f mean %.4f" % (r["n_hits"], w.min(), np.percentile(w, 50), w.mean()), flush=True)
