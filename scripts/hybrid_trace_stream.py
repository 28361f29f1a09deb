"""Phase clocks of the two-launch hybrid query on the distinct-query stream (bench.py's configs[4] inputs): per tile of the
tile kernel (split: tiles that hold vectors / the rest, first round / later), and of the two branches of the reduce kernel
relative to the tile kernel's end.  Knobs via RSGPU_TUNING=key=value,..."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench as B  # noqa: E402
from redisearch_amd import search as S  # noqa: E402
from redisearch_amd import vecsim as V  # noqa: E402

torch.cuda.set_device(0)
lib = V.load()
for kv in os.environ.get("RSGPU_TUNING", "").split(","):
    if kv:
        key, val = kv.split("=")
        assert lib.RSGPU_SetTuning(key.encode(), int(val)) == 0, kv
n_docs, n_vec, dim, n_a, n_b = 50_000_000, 5_000_000, 768, 2, 2
rng = np.random.default_rng(149)
doc_len = (50 + rng.poisson(150, n_docs + 1)).astype(np.uint32)
table = S.DocTable(doc_len, np.ones(n_docs + 1, np.float32))
avg = float(doc_len[1:].mean())
idx = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2)
idx.reserve(n_vec)
idx.add_philox_rows(B.SEED, 0, n_vec, 1)
raws = [B._term_list(rng, n_docs, n_docs * 0.2 / r) for r in [2] * n_a + [4] * n_b]
lists = [S.Postings.from_flat(B.encode_freqs_only(d, f)) for d, f, _, _ in raws]
qv = B.philox_host_rows(V, B.QUERY_BASE + 100, 4, dim)
hqs = []
for qi, (i, j) in enumerate([(0, 2), (1, 3), (0, 3), (1, 2)]):
    dfs = [raws[i][0].size, raws[j][0].size]
    hqs.append(S.HybridQuery([lists[i], lists[j]], table, "BM25STD", [S.calculate_idf(n_docs, d) for d in dfs],
                             [S.calculate_idf_bm25(n_docs, d) for d in dfs], [1.0, 1.0], n_docs, avg, top_n=10, index=idx, q=qv[qi], k=10))
for hq in hqs * 3:
    hq.run()
lib.RSGPU_SetTuning(b"hybrid_trace", 1)
names = ["window ends", "window staged", "probe done", "hits compacted", "scored", "ranked + written", "distances", "end"]
for rep in range(2):
    for qi, hq in enumerate(hqs):
        hq.run()
        t = S.hybrid_trace().astype(np.int64)
        if os.environ.get("TRACE_NPY") and rep == 1:
            np.save(os.path.join("gpurun_out", os.environ["TRACE_NPY"] + "_q%d.npy" % qi), t)
        red, t = t[-2:], t[:-2]
        t0 = t[:, 0].min()
        d = np.diff(t, axis=1) / 100.0
        start, end = (t[:, 0] - t0) / 100.0, (t[:, 8] - t0) / 100.0
        has_vec = d[:, 6] > 1.0                      # tiles whose distance phase did something
        rec = {"rep": rep, "query": qi, "tiles": int(t.shape[0]), "kernel_span_us": float(end.max()),
               "tile_start_us": {q: float(np.percentile(start, q)) for q in (50, 75, 90, 100)},
               "tiles_with_vectors": int(has_vec.sum()),
               "vector_tiles": {"duration_p50": float(np.percentile((end - start)[has_vec], 50)), "end_p95": float(np.percentile(end[has_vec], 95)),
                                "end_max": float(end[has_vec].max()), "phases_mean": {n: round(float(d[has_vec, i].mean()), 2) for i, n in enumerate(names)}},
               "other_tiles": {"duration_p50": float(np.percentile((end - start)[~has_vec], 50)), "end_max": float(end[~has_vec].max()),
                               "first_round_duration": float((end - start)[~has_vec & (start < 5)].mean()),
                               "later_duration": float((end - start)[~has_vec & (start >= 5)].mean()) if (~has_vec & (start >= 5)).any() else None,
                               "phases_mean": {n: round(float(d[~has_vec, i].mean()), 2) for i, n in enumerate(names)}},
               "reduce_kernel_us_after_tile_kernel_end": {
                   br: {n: round(float((red[b, p] - t0) / 100.0 - end.max()), 2) for p, n in
                        enumerate(["start", "firsts loaded", "bound", "passing tiles listed", "entries collected", "ranked + written"])}
                   for b, br in enumerate(["scores", "knn"])}}
        print(json.dumps(rec), flush=True)
lib.RSGPU_SetTuning(b"hybrid_trace", 0)
