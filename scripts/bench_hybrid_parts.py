#!/usr/bin/env python3
"""BASELINE configs[4] through the two-launch hybrid query with one branch at a time: the device time of the tile / reduce
kernels for (both branches, score only, KNN only) and with 2 / 3 lists -- where the tile kernel's time goes."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import bench as B  # noqa: E402
from redisearch_amd import search as S  # noqa: E402
from redisearch_amd import vecsim as V  # noqa: E402


def main():
    lib = V.load()
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        assert lib.RSGPU_SetTuning(k.encode(), int(v)) == 0, kv
    n_docs, n_vec, dim = 50_000_000, 5_000_000, 768
    rng = np.random.default_rng(49)
    raw = []
    for r in (2, 4):
        docs = np.flatnonzero(rng.random(n_docs + 1) < 0.2 / r).astype(np.uint64)
        docs = docs[docs > 0]
        freqs = np.minimum(1 + rng.geometric(0.5, docs.size), 255).astype(np.uint32)
        raw.append((docs, freqs))
    doc_len = (50 + rng.poisson(150, n_docs + 1)).astype(np.uint32)
    table = S.DocTable(doc_len, np.ones(n_docs + 1, np.float32))
    avg = float(doc_len[1:].mean())
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2)
    idx.reserve(n_vec)
    idx.add_philox_rows(B.SEED, 0, n_vec, 1)
    q = B.philox_host_rows(V, B.QUERY_BASE, 1, dim)[0]
    g = [S.Postings.from_flat(B.encode_freqs_only(d, f)) for d, f in raw]
    idf = [S.calculate_idf(n_docs, d.size) for d, _ in raw]
    forms = {
        "both": S.HybridQuery(g, table, "BM25STD", idf, idf, [1.0, 1.0], n_docs, avg, top_n=10, index=idx, q=q, k=10),
        "score_only": S.HybridQuery(g, table, "BM25STD", idf, idf, [1.0, 1.0], n_docs, avg, top_n=10),
        "docscore_only": S.HybridQuery(g, table, "DOCSCORE", idf, idf, [1.0, 1.0], n_docs, avg, top_n=10),
        "knn_only": S.HybridQuery(g, index=idx, q=q, k=10),
        "knn_k1": S.HybridQuery(g, index=idx, q=q, k=1),
        "one_list_score": S.HybridQuery(g[1:], table, "DOCSCORE", idf[1:], idf[1:], [1.0], n_docs, avg, top_n=10),
    }
    import gc
    gc.disable()
    for rep in range(2):
        for name, hq in forms.items():
            for _ in range(3):
                hq.run()
            walls = []
            for _ in range(40):
                t0 = time.perf_counter()
                hq.run()
                walls.append((time.perf_counter() - t0) * 1e3)
            tile, red = [], []
            lib.RSGPU_SetProfiling(1)
            for _ in range(5):
                hq.run()
                p = S.profile()
                tile.append(p["intersect_ms"])
                red.append(p["topn_ms"])
            lib.RSGPU_SetProfiling(0)
            print(json.dumps({"rep": rep, "form": name, "path": S.hybrid_path(), "hits": int(hq.results()["n_hits"]),
                              "wall_p50_ms": float(np.percentile(walls, 50)), "tile_ms": min(tile), "reduce_ms": min(red)}), flush=True)
    # where a tile's time goes: the phase clock of every tile (100 MHz ticks)
    lib.RSGPU_SetTuning(b"hybrid_trace", 1)
    for name in ("both", "score_only", "knn_only"):
        hq = forms[name]
        for _ in range(3):
            hq.run()
        t = S.hybrid_trace().astype(np.int64)
        t0 = t[:, 0].min()
        names = ["window ends", "window staged", "probe done", "hits compacted", "scored", "ranked + written", "distances", "end"]
        d = np.diff(t, axis=1) / 100.0      # us
        start = (t[:, 0] - t0) / 100.0
        end = (t[:, 8] - t0) / 100.0
        rec = {"form": name, "tiles": int(t.shape[0]), "kernel_span_us": float(end.max()),
               "tile_start_us": {q: float(np.percentile(start, q)) for q in (0, 25, 50, 75, 90, 100)},
               "tile_duration_us": {q: float(np.percentile(end - start, q)) for q in (5, 50, 95)},
               "phase_mean_us": {n: float(d[:, i].mean()) for i, n in enumerate(names)},
               "phase_p95_us": {n: float(np.percentile(d[:, i], 95)) for i, n in enumerate(names)},
               "first_round_tiles_mean_duration_us": float((end - start)[start < 5].mean()),
               "late_tiles_mean_duration_us": float((end - start)[start >= 5].mean()) if (start >= 5).any() else None}
        print(json.dumps(rec), flush=True)
    lib.RSGPU_SetTuning(b"hybrid_trace", 0)


if __name__ == "__main__":
    main()
