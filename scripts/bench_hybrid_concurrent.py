#!/usr/bin/env python3
"""BASELINE configs[4] from n caller threads at once (RediSearch's WORKERS n): every thread owns an argument block and a query
vector and calls RSGPU_HybridQuery in a loop (ctypes releases the GIL); QPS, latency percentiles, answers compared with each
thread's serial answers.  Both forms of the call (hybrid_tiles = 1 / 0) in one process."""
import json
import os
import sys
import threading
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
    import gc
    gc.disable()
    # pthread callers through the plain C ABI (examples/concurrent_hybrid_callers.c): what a C module's worker pool pays -- no
    # interpreter between two calls
    import ctypes as C
    import subprocess
    import tempfile
    libdir = os.path.join(ROOT, "redisearch_amd", "lib")
    so = os.path.join(tempfile.mkdtemp(prefix="rs_hcallers_"), "libhybrid_callers.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "concurrent_hybrid_callers.c"),
                           "-L" + libdir, "-lVectorSimilarity", "-Wl,-rpath," + libdir, "-lpthread", "-o", so])
    cl = C.CDLL(so)
    cl.rs_hybrid_callers_run.restype = C.c_long
    cl.rs_hybrid_callers_run.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_double)]
    blocks = (C.c_void_p * len(hqs))(*[C.addressof(h.args) for h in hqs])
    for threads in (1, 2, 4, 8, 16):
        cap = 100000
        lat = np.zeros((threads, cap), np.uint64)
        counts = np.zeros(threads, np.uint64)
        el = C.c_double(0)
        total = cl.rs_hybrid_callers_run(blocks, len(hqs), threads, 1.0, lat.ctypes.data_as(C.c_void_p), cap, counts.ctypes.data_as(C.c_void_p), C.byref(el))
        allv = np.concatenate([lat[t, :int(min(counts[t], cap))] for t in range(threads)]).astype(np.float64) / 1e6
        print(json.dumps({"driver": "C pthreads", "threads": threads, "queries": int(total), "qps": total / el.value,
                          "p50_ms": float(np.percentile(allv, 50)), "p95_ms": float(np.percentile(allv, 95))}), flush=True)
    if os.environ.get("C_ONLY") == "1":
        return
    for tiles in (1, 0, 1):
        lib.RSGPU_SetTuning(b"hybrid_tiles", tiles)
        serial = []
        for hq in hqs:
            hq.run()
            serial.append(hq.results())
        for threads in (1, 2, 4, 8, 16):
            lat = [[] for _ in range(threads)]
            ok = [True] * threads
            stop = time.perf_counter() + 1.0
            go = threading.Barrier(threads)

            def work(t):
                hq = hqs[t]
                go.wait()
                while True:
                    t0 = time.perf_counter()
                    if t0 >= stop:
                        break
                    hq.run()
                    lat[t].append(time.perf_counter() - t0)
                r = hq.results()
                ok[t] = (r["n_hits"] == serial[t]["n_hits"] and all(r[k][j].tolist() == serial[t][k][j].tolist() for k in ("top", "knn") for j in (0, 1)))
            th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
            t_start = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            el = time.perf_counter() - t_start
            all_lat = np.concatenate([np.array(x) for x in lat]) * 1e3
            print(json.dumps({"hybrid_tiles": tiles, "threads": threads, "queries": int(all_lat.size), "qps": all_lat.size / el,
                              "p50_ms": float(np.percentile(all_lat, 50)), "p95_ms": float(np.percentile(all_lat, 95)),
                              "answers_identical_to_serial": bool(all(ok))}), flush=True)
    lib.RSGPU_SetTuning(b"hybrid_tiles", 1)


if __name__ == "__main__":
    main()
