#!/usr/bin/env python3
"""BASELINE configs[4] (bench.py's hybrid sub-record) with engine knobs A/B'd INSIDE one process: the corpus, posting lists
and the query are built once, every knob setting is timed in turn, twice (box-to-box spread is larger than most effects:
only same-process comparisons count).  usage: bench_hybrid_ab.py key=v1,v2 [key2=v1,v2 ...]  -> gpurun_out/hybrid_ab.json"""
import itertools
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (before the engine: see docs/DESIGN_NOTES.md 9)
import bench as B  # noqa: E402
from redisearch_amd import search as S  # noqa: E402
from redisearch_amd import vecsim as V  # noqa: E402


def main():
    knobs = []
    for a in sys.argv[1:]:
        k, vs = a.split("=")
        knobs.append((k, [int(x) for x in vs.split(",")]))
    lib = V.load()
    n_docs, n_vec, dim = 50_000_000, int(os.environ.get("N_VEC", 5_000_000)), 768   # N_VEC=50000000: every document has a vector
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
    enc = [B.encode_freqs_only(d, f) for d, f in raw]
    idf = [S.calculate_idf(n_docs, d.size) for d, _ in raw]
    bidf = [S.calculate_idf_bm25(n_docs, d.size) for d, _ in raw]
    g = [S.Postings.from_flat(e) for e in enc]
    lib.RSGPU_SetTuning(b"cache_decoded", 0)
    g_cold = [S.Postings.from_flat(e) for e in enc]
    lib.RSGPU_SetTuning(b"cache_decoded", 1)
    hq = S.HybridQuery(g, table, "BM25STD", idf, bidf, [1.0, 1.0], n_docs, avg, top_n=10, index=idx, q=q, k=10)
    hq_cold = S.HybridQuery(g_cold, table, "BM25STD", idf, bidf, [1.0, 1.0], n_docs, avg, top_n=10, index=idx, q=q, k=10)
    hq.run()
    ref = hq.results()
    out = []
    import gc
    gc.disable()
    for rep in range(2):
        for combo in itertools.product(*[vs for _, vs in knobs]):
            for (k, _), v in zip(knobs, combo):
                assert lib.RSGPU_SetTuning(k.encode(), v) == 0, k
            rec = {"rep": rep, "knobs": {k: v for (k, _), v in zip(knobs, combo)}}
            for name, h, cache in (("warm", hq, 1), ("cold", hq_cold, 0)):
                lib.RSGPU_SetTuning(b"cache_decoded", cache)
                for _ in range(3):
                    h.run()
                r = h.results()
                same = (r["n_hits"] == ref["n_hits"] and r["top"][0].tolist() == ref["top"][0].tolist() and
                        r["top"][1].tolist() == ref["top"][1].tolist() and r["knn"][0].tolist() == ref["knn"][0].tolist() and
                        r["knn"][1].tolist() == ref["knn"][1].tolist())
                walls = []
                for _ in range(60):
                    t0 = time.perf_counter()
                    h.run()
                    walls.append((time.perf_counter() - t0) * 1e3)
                lib.RSGPU_SetProfiling(1)
                h.run()
                prof = S.profile()
                lib.RSGPU_SetProfiling(0)
                rec[name] = {"p50_ms": float(np.percentile(walls, 50)), "p95_ms": float(np.percentile(walls, 95)), "min_ms": min(walls),
                             "same_answers": bool(same),
                             "stage_device_ms": {k_: prof.get(k_) for k_ in ("decode_ms", "intersect_ms", "score_ms", "topn_ms", "knn_ms")}}
            lib.RSGPU_SetTuning(b"cache_decoded", 1)
            out.append(rec)
            print(json.dumps(rec), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "hybrid_ab.json"), "w") as f:
        json.dump({"encoded_bytes": sum(x.num_bytes for x in g), "hits": int(ref["n_hits"]), "runs": out}, f, indent=1)


if __name__ == "__main__":
    main()
