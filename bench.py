"""bench.py -- BASELINE.json's headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE configs[1] -- 10M x 768 fp32 FLAT, COSINE, top-10, single-query
stream through the VecSim C ABI (VecSimIndex_TopKQuery -> reply), corpus resident in HBM, synthetic
U(-1,1) data.  A "step" is one KNN query = one pass over the rank's corpus shard.
N > 1: the corpus is row-sharded, 10M rows per GPU (weak scaling: 80M rows at N=8, BASELINE
configs[3]); every query runs on all shards, per-shard top-10 (fp32 score, u64 label) are exchanged
with an RCCL all-gather over xGMI and merged.  `value` counts 10M-row shard scans per second over all
ranks (N x the global QPS on the sharded corpus), so that perfect weak scaling reads N x the 1-GPU
value; the global QPS is in config.

Adds `roofline` (scan kernel, HIP events on its own stream inside the timed region) and
`cpu_baseline` (the CPU oracle's FLAT scan on a bounded sample, rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per GPU (default: the BASELINE config)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", choices=["cosine", "l2", "ip"], default="cosine",
                    help="cosine = BASELINE configs[1] (default); l2 = the metric configs[3] names for the 8-GPU corpus")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=200_000)
    ap.add_argument("--no-two-stage-extra", action="store_true", help="skip the extra two-stage measurement (N=1)")
    ap.add_argument("--tuning", action="append", default=[], help="engine knob key=value (A/B experiments only)")
    ap.add_argument("--cpu-config0", action="store_true",
                    help="no GPU: time BASELINE configs[0] (100k x 128 fp32 L2 top-10, single query) on the host cores with "
                         "the cpu_baseline port and print one JSON line")
    return ap.parse_args()


def cpu_baseline(dim, k, sample_rows, full_rows, budget_s=15.0, metric="cosine"):
    """The oracle's FLAT scan (scalar max-heap, one thread = how one RediSearch worker runs one FLAT
    query) timed on a bounded sample of the same workload, scaled linearly in rows to the full size."""
    import oracle as O
    import subprocess
    lib = O.lib
    native = os.path.join(ROOT, "oracle", "_build", "liboracle_native.so")
    try:  # prefer a -march=native build made on THIS host
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        nat = C.CDLL(native)
        for name in ("oflat_new", "oflat_add_bulk", "oflat_topk_heap", "oflat_free"):
            getattr(nat, name).restype = getattr(lib, name).restype
            getattr(nat, name).argtypes = getattr(lib, name).argtypes
        lib = nat
        flavour = "march=native"
    except Exception:
        flavour = "portable(avx512/avx2 clones)"
    rng = np.random.default_rng(47)
    data = rng.uniform(-1, 1, (sample_rows, dim)).astype(np.float32)
    h = lib.oflat_new(O.F32, dim, {"cosine": O.COSINE, "l2": O.L2, "ip": O.IP}[metric], 0, 1024)
    lib.oflat_add_bulk(h, data.ctypes.data_as(C.c_void_p), sample_rows, 1)
    qs = np.random.default_rng(48).uniform(-1, 1, (64, dim)).astype(np.float32)
    ids = np.zeros(k, np.uint64)
    sc = np.zeros(k, np.float64)
    n, t0 = 0, time.perf_counter()
    while True:
        q = qs[n % len(qs)]
        lib.oflat_topk_heap(h, q.ctypes.data_as(C.c_void_p), k, ids.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p))
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 2000:
            break
    # (b) eight concurrent queries on eight threads (how a WORKERS 8 deployment runs FLAT queries);
    # ctypes releases the GIL inside the C scan
    import threading
    counts = [0] * 8
    stop_at = time.perf_counter() + min(6.0, 0.4 * budget_s)

    def worker(t):
        ids_t, sc_t = np.zeros(k, np.uint64), np.zeros(k, np.float64)
        i = t
        while time.perf_counter() < stop_at:
            q = qs[i % len(qs)]
            lib.oflat_topk_heap(h, q.ctypes.data_as(C.c_void_p), k, ids_t.ctypes.data_as(C.c_void_p), sc_t.ctypes.data_as(C.c_void_p))
            counts[t] += 1
            i += 8

    t8 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    el8 = time.perf_counter() - t8
    qps8 = sum(counts) / el8 * sample_rows / full_rows
    lib.oflat_free(h)
    qps_sample = n / el
    return {"value": qps_sample * sample_rows / full_rows, "unit": "queries/s", "cores": 1, "kind": "port",
            "eight_threads": {"value": qps8, "cores": 8, "note": "8 concurrent queries, same sample, scaled by rows"},
            "sample": "%d queries over a %d x %d fp32 %s sample in %.1f s (oracle/flat_oracle.c oflat_topk_heap, %s), "
                      "scaled by rows to %d" % (n, sample_rows, dim, metric, el, flavour, full_rows)}


def cpu_config0(n=100_000, dim=128, k=10, nq=1000):
    """BASELINE configs[0] -- the reference's own CPU-runnable case -- on host cores with the same port the
    cpu_baseline leg uses (warm-up 3, 1000 queries, p50 / p95 / QPS).  No GPU involved."""
    import platform
    import subprocess
    import oracle as O
    lib, flavour = O.lib, "portable(avx512/avx2 clones)"
    try:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
        nat = C.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle_native.so"))
        for name in ("oflat_new", "oflat_add_bulk", "oflat_topk_heap", "oflat_free"):
            getattr(nat, name).restype = getattr(lib, name).restype
            getattr(nat, name).argtypes = getattr(lib, name).argtypes
        lib, flavour = nat, "march=native"
    except Exception:
        pass
    data = np.random.default_rng(47).uniform(-1, 1, (n, dim)).astype(np.float32)
    qs = np.random.default_rng(48).uniform(-1, 1, (nq, dim)).astype(np.float32)
    h = lib.oflat_new(O.F32, dim, O.L2, 0, 1024)
    lib.oflat_add_bulk(h, data.ctypes.data_as(C.c_void_p), n, 1)
    ids, sc = np.zeros(k, np.uint64), np.zeros(k, np.float64)
    lat = np.zeros(nq)
    run = lambda q: lib.oflat_topk_heap(h, q.ctypes.data_as(C.c_void_p), k, ids.ctypes.data_as(C.c_void_p),
                                        sc.ctypes.data_as(C.c_void_p))
    for i in range(3):
        run(qs[i])
    t0 = time.perf_counter()
    for i in range(nq):
        s = time.perf_counter()
        run(qs[i])
        lat[i] = time.perf_counter() - s
    el = time.perf_counter() - t0
    lib.oflat_free(h)
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    return {"config": "BASELINE configs[0]: %dx%d fp32 FLAT L2 top-%d, single query, CPU" % (n, dim, k),
            "kind": "port", "sample": "oracle/flat_oracle.c oflat_topk_heap, %s, 1 thread, %d queries" % (flavour, nq),
            "cores": 1, "value": nq / el, "unit": "queries/s", "p50_ms": float(np.percentile(lat, 50) * 1e3),
            "p95_ms": float(np.percentile(lat, 95) * 1e3), "gb_per_s": n * dim * 4 * nq / el / 1e9,
            "host": {"cpu": cpu, "nproc": os.cpu_count(), "machine": platform.machine()}}


def main():
    a = parse()
    if a.cpu_config0:
        print(json.dumps(cpu_config0()), flush=True)
        return
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, "launch with --nproc-per-node == --gpus"
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # RSGPU_BENCH_FORCE_DIST=1 (test hook): create the RCCL process group and run the real all-gather even with one
    # rank (under torch.distributed.run --nproc-per-node 1), so the collective code path is exercised on a 1-GPU box
    force_dist = os.environ.get("RSGPU_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from redisearch_amd import vecsim as V
    lib = V.load()
    for kv in a.tuning:
        key, val = kv.split("=")
        assert lib.RSGPU_SetTuning(key.encode(), int(val)) == 0, kv
    two_stage = any(kv.replace(" ", "") in ("shadow16=1", "shadow8=1") for kv in a.tuning)
    shadow8 = any(kv.replace(" ", "") == "shadow8=1" for kv in a.tuning)
    # N=1: the index also carries the int8 shadow of the opt-in two-stage exact scan; it is switched OFF for the
    # timed headline loop (plain fp32 scan) and measured separately afterwards as an extra config entry
    extra_two_stage = world == 1 and not two_stage and not a.no_two_stage_extra and a.metric == "cosine"
    if extra_two_stage:
        lib.RSGPU_SetTuning(b"shadow8", 1)
        lib.RSGPU_SetTuning(b"two_stage", 0)

    # ---- corpus: rows_per_gpu x dim fp32 generated in HBM, shard r holds labels r*rows+1 .. (r+1)*rows
    rows, dim, k = a.rows, a.dim, a.k
    index = V.VecSimIndex(V.VecSimType_FLOAT32, dim, {"cosine": V.VecSimMetric_Cosine, "l2": V.VecSimMetric_L2,
                                                      "ip": V.VecSimMetric_IP}[a.metric])
    index.reserve(rows)
    gen = torch.Generator(device=dev)
    gen.manual_seed(47 + rank)
    chunk = 1_000_000
    done = 0
    while done < rows:
        m = min(chunk, rows - done)
        t = torch.rand((m, dim), device=dev, dtype=torch.float32, generator=gen).mul_(2.0).sub_(1.0)
        torch.cuda.synchronize()
        index.add_device_rows(t.data_ptr(), m, rank * rows + done + 1)
        done += m
        del t
    torch.cuda.empty_cache()
    assert index.index_size() == rows
    queries = np.random.default_rng(48).uniform(-1, 1, (1000, dim)).astype(np.float32)

    # RSGPU_BENCH_FORCE_SHARDED=1 (test hook): run the sharded code path -- device-side per-shard top-k, packing,
    # merge -- on a single rank too (the all-gather itself needs >= 2 ranks)
    force_sharded = os.environ.get("RSGPU_BENCH_FORCE_SHARDED") == "1" or force_dist
    if world > 1 or force_sharded:
        from redisearch_amd.sharded import ShardedTopK
        loc_s = torch.empty(k, device=dev, dtype=torch.float32)
        loc_l = torch.empty(k, device=dev, dtype=torch.int64)

        def local_topk(q, kk):
            index.topk_device(q, kk, loc_s.data_ptr(), loc_l.data_ptr())
            return loc_s, loc_l

        sharded = ShardedTopK(local_topk, k, dev)

    def one_query(i):
        q = queries[i % len(queries)]
        if world == 1 and not force_sharded:
            rep = lib.VecSimIndex_TopKQuery(index.ptr, q.ctypes.data_as(C.c_void_p), k, None, V.BY_SCORE)
            n = lib.VecSimQueryReply_Len(rep)
            lib.VecSimQueryReply_Free(rep)
            return n
        labels, _ = sharded.query(q)   # per-shard top-k -> RCCL all-gather -> merge
        return len(labels)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        assert one_query(i) == k
    lib.RSGPU_ResetProfile()
    lib.RSGPU_SetProfiling(1)
    lat = np.zeros(a.steps)
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        s = time.perf_counter()
        one_query(a.warmup + i)
        lat[i] = time.perf_counter() - s
    barrier()
    elapsed = time.perf_counter() - t0
    lib.RSGPU_SetProfiling(0)
    launches, kern_ms, kern_bytes = V.scan_profile()
    extra = None
    if extra_two_stage:  # same index, same queries, two-stage exact scan switched on; results must be identical
      try:
        lib.RSGPU_SetTuning(b"two_stage", 1)
        for i in range(5):
            one_query(i)
        q = queries[7]
        rep = index.topk_query(q, k)
        ids2, sc2 = rep.results()
        lib.RSGPU_SetTuning(b"two_stage", 0)
        ids1, sc1 = index.topk_query(q, k).results()
        lib.RSGPU_SetTuning(b"two_stage", 1)
        same = ids1.tolist() == ids2.tolist() and sc1.tolist() == sc2.tolist()
        torch.cuda.synchronize()
        n2 = min(a.steps, 200)
        t2 = time.perf_counter()
        for i in range(n2):
            one_query(a.warmup + i)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t2
        lib.RSGPU_SetTuning(b"two_stage", 0)
        extra = {"qps": n2 / el2, "ms_per_query": el2 / n2 * 1e3, "queries": n2, "bit_identical_to_fp32_scan": bool(same),
                 "what": "opt-in: scan of an int8 shadow with per-row scales (7.72 GB) + error-bounded filter + fp32 "
                         "re-scoring of the survivors; NOT the headline value"}
      except Exception as e:  # the extra must never cost the headline line
        lib.RSGPU_SetTuning(b"two_stage", 0)
        extra = {"error": repr(e)}

    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        global_qps = a.steps / elapsed
        avg_kernel_s = (kern_ms / 1e3) / max(launches, 1)
        achieved = (kern_bytes / max(launches, 1)) / avg_kernel_s / 1e9 if launches else 0.0
        out = {
            "metric": "KNN queries/sec + p50 latency, 10M\u00d7768 fp32 FLAT top-10, 1/2/4/8 GPU",  # BASELINE.json's metric
            "value": global_qps * world,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%dx%d fp32 FLAT %s top-%d per GPU, single-query stream via VecSimIndex_TopKQuery"
                            % (rows, dim, a.metric.upper(), k),
                "value_definition": "10M-row shard scans per second summed over the GPUs (= n_gpus x the global QPS on the "
                                    "row-sharded corpus; at n_gpus=1 it IS the QPS); p50/p95 latency below",
                "rows_per_gpu": rows, "dim": dim, "k": k, "metric": a.metric.upper(),
                "corpus_rows_total": rows * world,
                "parallelism": "row-sharded x%d, RCCL all-gather of per-shard top-k + merge" % world if world > 1 else "single GPU",
                "global_qps_on_sharded_corpus": global_qps,
                "p50_ms": float(np.percentile(lat, 50) * 1e3), "p95_ms": float(np.percentile(lat, 95) * 1e3),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": ("scan_kernel<i8,IPS,G=16,ITERS=3,U=4> over the int8 shadow (two-stage exact scan)" if two_stage and shadow8
                           else "scan_kernel<f16,IP,G=32,ITERS=3,U=4> over the fp16 shadow (two-stage exact scan)" if two_stage
                           else "scan_kernel<f32,%s,G=64,ITERS=3,U=8> (FLAT scan)" % ("L2" if a.metric == "l2" else "IP")), "launches": int(launches),
                "avg_kernel_ms": avg_kernel_s * 1e3, "algorithmic_bytes_per_launch": kern_bytes / max(launches, 1),
            },
        }
        if extra is not None:
            out["config"]["two_stage_exact_scan_extra"] = extra
        if two_stage:  # opt-in experiment (--tuning shadow16=1), never the default line
            out["config"]["two_stage_fp16_shadow"] = ("scan of an fp16 shadow + error-bounded filter + fp32 re-scoring of the "
                                                      "survivors: results bit-identical to the fp32 scan; roofline bytes = shadow bytes")
        # HBM traffic of the scan kernel from the committed PMC pass (separate rocprofv3 --pmc runs of
        # this same command, corrected as MI355X_MICROARCH.md prescribes); bench.py cannot read PMCs itself
        pmc = os.path.join(ROOT, "profiles", "r01_scan_pmc_hbm_traffic.json")
        if os.path.exists(pmc):
            p = json.load(open(pmc))
            if p.get("rows") == rows and p.get("dim") == dim and not two_stage:
                out["roofline"]["traffic"] = p["traffic_bytes_per_launch"]
                out["roofline"]["traffic_source"] = "profiles/r01_scan_pmc_hbm_traffic.json (FETCH_SIZE x2 + WRITE_SIZE, KB->B)"
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(dim, k, min(a.cpu_sample_rows, rows), rows, metric=a.metric)
            except Exception as e:  # the baseline leg must never cost the measured line
                out["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
