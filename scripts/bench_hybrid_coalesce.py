#!/usr/bin/env python3
"""BASELINE configs[4] from n pthread callers (RediSearch's WORKERS n; examples/concurrent_hybrid_callers.c) with the hybrid
coalescer off / on at several depths and both block orders: QPS, latency percentiles, the coalescer's counters.  One JSON line per
(setting, threads); OUT=<name> also writes them to gpurun_out/<name>."""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import bench as B  # noqa: E402
from redisearch_amd import search as S  # noqa: E402
from redisearch_amd import vecsim as V  # noqa: E402


def main():
    lib = V.load()
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
    qs = B.philox_host_rows(V, B.QUERY_BASE, 16, dim)
    g = [S.Postings.from_flat(B.encode_freqs_only(d, f)) for d, f in raw]
    idf = [S.calculate_idf(n_docs, d.size) for d, _ in raw]
    hqs = [S.HybridQuery(g, table, "BM25STD", idf, idf, [1.0, 1.0], n_docs, avg, top_n=10, index=idx, q=qs[t], k=10) for t in range(16)]
    lib.RSGPU_SetTuning(b"hybrid_coalesce", 0)
    serial = []
    for hq in hqs:
        hq.run()
        serial.append(hq.results())
    import gc
    gc.disable()
    libdir = os.path.join(ROOT, "redisearch_amd", "lib")
    so = os.path.join(tempfile.mkdtemp(prefix="rs_hcallers_"), "libhybrid_callers.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "concurrent_hybrid_callers.c"),
                           "-L" + libdir, "-lVectorSimilarity", "-Wl,-rpath," + libdir, "-lpthread", "-o", so])
    cl = C.CDLL(so)
    cl.rs_hybrid_callers_run.restype = C.c_long
    cl.rs_hybrid_callers_run.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_double)]
    blocks = (C.c_void_p * len(hqs))(*[C.addressof(h.args) for h in hqs])
    settings = [(0, 0, 0)] + [(1, d, i) for d in (1, 2, 3) for i in (1, 0)]
    if os.environ.get("SETTINGS"):
        settings = [tuple(int(x) for x in s.split(",")) for s in os.environ["SETTINGS"].split(";")]
    lines = []
    for on, depth, inter in settings:
        lib.RSGPU_SetTuning(b"hybrid_coalesce", on)
        if on:
            lib.RSGPU_SetTuning(b"hybrid_coalesce_depth", depth)
            lib.RSGPU_SetTuning(b"hybrid_coalesce_interleave", inter)
        for threads in (1, 2, 4, 8, 16):
            cap = 100000
            lat = np.zeros((threads, cap), np.uint64)
            counts = np.zeros(threads, np.uint64)
            el = C.c_double(0)
            S.hybrid_coalesce_stats(reset=True)
            total = cl.rs_hybrid_callers_run(blocks, len(hqs), threads, float(os.environ.get("SECONDS", "0.6")), lat.ctypes.data_as(C.c_void_p), cap,
                                             counts.ctypes.data_as(C.c_void_p), C.byref(el))
            st = S.hybrid_coalesce_stats()
            allv = np.concatenate([lat[t, :int(min(counts[t], cap))] for t in range(threads)]).astype(np.float64) / 1e6
            same = all(h.results()["top"][0].tolist() == s["top"][0].tolist() and h.results()["knn"][0].tolist() == s["knn"][0].tolist() and
                       h.results()["knn"][1].tolist() == s["knn"][1].tolist() and h.results()["top"][1].tolist() == s["top"][1].tolist()
                       for h, s in zip(hqs, serial))
            line = {"coalesce": on, "depth": depth, "interleave": inter, "threads": threads, "queries": int(total), "qps": total / el.value,
                    "p50_ms": float(np.percentile(allv, 50)), "p95_ms": float(np.percentile(allv, 95)), "stats": st,
                    "queries_per_grid": (st["grid_queries"] / st["grids"]) if st["grids"] else None, "answers_identical_to_serial": bool(same)}
            lines.append(line)
            print(json.dumps(line), flush=True)
    if os.environ.get("OUT"):
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(lines, open(os.path.join(ROOT, "gpurun_out", os.environ["OUT"]), "w"), indent=1)


if __name__ == "__main__":
    main()
