"""bench.py -- BASELINE.json's headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE configs[1] -- 10M x 768 fp32 FLAT, COSINE, top-10, single-query stream through
the VecSim C ABI (VecSimIndex_TopKQuery -> reply), corpus resident in HBM.  A "step" is one KNN query = one pass over
the corpus.  The corpus is SYNTHETIC and KEYED: element (i, j) = Philox4x32-10(seed; i, j) mapped to
[-1, 1), generated in place in HBM by RSGPU_FlatIndex_AddPhiloxRows -- any host can regenerate any row, and the
cpu_baseline leg does: after the timed loop the rows one timed query returned are regenerated on the host and re-scored
in fp64 (`config.verify`).  tests/test_gpu_fullsize.py checks whole queries of this very corpus against the CPU oracle.

Every input of the timed regions comes from the product or from numpy: the corpus and the queries from the library's
Philox kernel (queries read back through a scratch index), the hybrid extra's posting lists from encode_freqs_only
below.  oracle/ is imported in the cpu_baseline leg only -- the CPU timing, the verification of the timed answers
(`config.verify`, `cpu_baseline.gpu_answers_checked_...`) and the oracle check of the hybrid extra -- never with
--no-cpu-baseline (tests/test_bench_contract_cpu.py::test_bench_imports_the_oracle_only_in_its_cpu_leg).

N > 1 (weak scaling, BASELINE configs[3] at N=8: 80M rows): the corpus is row-sharded, 10M rows per GPU, every query
runs on all shards and the per-shard top-10 lists are merged by (score, label).
  * launched by torch.distributed.run (WORLD_SIZE set; how the driver runs it): one rank per GPU, per-shard top-k
    exchanged with ONE RCCL all-gather over xGMI, merged in C (RSGPU_MergeTopKPacked);
  * launched as a plain `python bench.py --gpus N` (WORLD_SIZE unset): ONE process drives the N devices through
    the ordinary VecSimIndex handle over N device shards (RSGPU_SetTuning("shards", N): a worker thread per device, host
    K-way merge behind VecSimIndex_TopKQuery) -- the in-process form a Redis module would use.
`value` = shard scans per second summed over the GPUs = N x the global QPS on the sharded corpus (at N=1 it is the
QPS); perfect weak scaling reads N x the 1-GPU value.  The global QPS and p50/p95 latency are in `config`.

The JSON line also carries `roofline` (scan kernel: HIP events on its own stream inside the timed region; kernel name
reported by the library; PMC traffic from the committed rocprofv3 pass), `cpu_baseline` (the CPU oracle's FLAT scan on
the host cores -- on the FULL 10M-row corpus when the host has the memory, else on a bounded sample, and it says which)
and, at N=1, sub-records measured after the headline loop: `config.two_stage_exact_scan_extra`,
`config.batched_mfma` (BASELINE configs[2]) and `config.hybrid` (BASELINE configs[4]).
"""
import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

import numpy as np

# RCCL prints its version banner to stdout when a communicator comes up; this script's stdout is the driver's one JSON line:
# ask the library to route the banner to stderr (opt-in since round 5 -- the swap is process-wide, csrc/shard_comm.cpp)
os.environ.setdefault("RSGPU_QUIET_RCCL_BANNER", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0    # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak (same guide); never the 2:1-sparsity figure
SEED = 47
QUERY_BASE = 1 << 40     # queries are corpus-generator rows far beyond any corpus row


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
    ap.add_argument("--cpu-baseline-mode", choices=["auto", "full", "sample"], default="auto")
    ap.add_argument("--no-extras", action="store_true", help="skip the N=1 sub-records (two-stage, batched MFMA, hybrid)")
    ap.add_argument("--no-two-stage-extra", action="store_true")
    ap.add_argument("--no-callers-extra", action="store_true")
    ap.add_argument("--no-batched-extra", action="store_true")
    ap.add_argument("--no-hybrid-extra", action="store_true")
    ap.add_argument("--replicas", action="store_true", help="N>1, single process: N full replicas instead of N shards "
                                                            "(throughput of concurrent callers; reported separately)")
    ap.add_argument("--tuning", action="append", default=[], help="engine knob key=value (A/B experiments only)")
    ap.add_argument("--preflight", action="store_true",
                    help="only the multi-GPU pre-flight: a small index sharded over EVERY visible device (hipSetDevice, VMM, pinned "
                         "host merge, then the RCCL exchange across all of them), answers held to a one-device index; prints one "
                         "JSON line and fails loudly with the device index")
    ap.add_argument("--cpu-config0", action="store_true",
                    help="no GPU: time BASELINE configs[0] (100k x 128 fp32 L2 top-10, single query) on the host cores with "
                         "the cpu_baseline port and print one JSON line")
    return ap.parse_args()


# ---- CPU baseline (oracle; rank 0, N=1 only) -------------------------------------------------------------------------
def _oracle_lib():
    """The oracle's scan library: a -march=native build made on THIS host when possible."""
    import subprocess
    import oracle as O
    lib, flavour = O.lib, "portable(avx512/avx2 clones)"
    try:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        nat = C.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle_native.so"))
        for name in ("oflat_new", "oflat_add_bulk", "oflat_topk_heap", "oflat_free"):
            getattr(nat, name).restype = getattr(lib, name).restype
            getattr(nat, name).argtypes = getattr(lib, name).argtypes
        lib, flavour = nat, "march=native"
    except Exception:
        pass
    return O, lib, flavour


def scan_source_hash():
    """sha256 (first 16 hex digits) of the single-query scan kernel's source: scan_kernels.hip + scan_ops.hpp"""
    import hashlib
    h = hashlib.sha256()
    for f in ("scan_kernels.hip", "scan_ops.hpp"):
        h.update(open(os.path.join(ROOT, "redisearch_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def _mem_available_gb():
    try:
        for l in open("/proc/meminfo"):
            if l.startswith("MemAvailable"):
                return int(l.split()[1]) / 1e6
    except Exception:
        pass
    return 0.0


def _cpus():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(dim, k, rows, sample_rows, metric, mode, gpu_check=None):
    """The oracle's FLAT scan (scalar K-bounded max-heap over every row, one thread = how one RediSearch worker runs
    one FLAT query; plus 8 concurrent queries on 8 threads = a WORKERS 8 deployment), on the same keyed corpus.
    full: all `rows` rows regenerated on the host (warm-up 3, >= 30 queries); sample: `sample_rows` rows,
    scaled linearly in rows.  gpu_check (query index -> (ids, scores)) lets the full mode double as an oracle check of
    the timed GPU answers."""
    import threading
    O, lib, flavour = _oracle_lib()
    om = {"cosine": O.COSINE, "l2": O.L2, "ip": O.IP}[metric]
    need_gb = rows * dim * 4 * 2.2 / 1e9
    full = mode == "full" or (mode == "auto" and _mem_available_gb() > need_gb + 16 and _cpus() >= 16)
    n_rows = rows if full else min(sample_rows, rows)
    t0 = time.perf_counter()
    if full:  # probe the generator's speed first: a slow host falls back to the sample
        O.philox_rows(SEED, 0, 100_000, dim)
        est = (time.perf_counter() - t0) * rows / 100_000
        if mode == "auto" and est > 90:
            full, n_rows = False, min(sample_rows, rows)
    data = O.philox_rows(SEED, 0, n_rows, dim)
    t_gen = time.perf_counter() - t0
    h = lib.oflat_new(O.F32, dim, om, 0, 1024)
    lib.oflat_add_bulk(h, data.ctypes.data_as(C.c_void_p), n_rows, 1)
    del data
    t_build = time.perf_counter() - t0
    qs = O.philox_rows(SEED, QUERY_BASE, 64, dim)
    ids, sc = np.zeros(k, np.uint64), np.zeros(k, np.float64)

    def run(q, i_out=ids, s_out=sc):
        return lib.oflat_topk_heap(h, q.ctypes.data_as(C.c_void_p), k, i_out.ctypes.data_as(C.c_void_p),
                                   s_out.ctypes.data_as(C.c_void_p))
    checked = None
    if full and gpu_check:
        checked = {"queries": 0, "ids_identical": True, "max_abs_score_diff": 0.0}
        for qi, (gi, gs) in gpu_check.items():
            m = run(qs[qi % 64])
            checked["queries"] += 1
            checked["ids_identical"] &= ids[:m].tolist() == list(gi)
            checked["max_abs_score_diff"] = max(checked["max_abs_score_diff"], float(np.max(np.abs(sc[:m] - np.asarray(gs)))))
    for i in range(3):
        run(qs[i])
    budget = 25.0 if full else 12.0
    lat = []
    t1 = time.perf_counter()
    while True:
        s = time.perf_counter()
        run(qs[(3 + len(lat)) % 64])
        lat.append(time.perf_counter() - s)
        el = time.perf_counter() - t1
        if (len(lat) >= 30 and el >= 0.5 * budget) or el >= budget or len(lat) >= 2000:
            break
    n1 = len(lat)
    counts = [0] * 8
    stop_at = time.perf_counter() + (8.0 if full else 5.0)

    def worker(t):
        ids_t, sc_t = np.zeros(k, np.uint64), np.zeros(k, np.float64)
        i = t
        while time.perf_counter() < stop_at or counts[t] == 0:
            run(qs[i % 64], ids_t, sc_t)
            counts[t] += 1
            i += 8
    t8 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [x.start() for x in th]
    [x.join() for x in th]
    el8 = time.perf_counter() - t8
    lib.oflat_free(h)
    scale = n_rows / rows
    out = {"value": n1 / el * scale, "unit": "queries/s", "cores": 1, "kind": "port",
           "p50_ms": float(np.percentile(lat, 50) * 1e3 / scale),
           "eight_threads": {"value": sum(counts) / el8 * scale, "cores": 8, "note": "8 concurrent queries on 8 threads"},
           "host_cpus": _cpus(),
           "sample": ("FULL corpus: %d queries (after 3 warm-up) over all %d x %d fp32 %s rows regenerated on the host "
                      "(keyed Philox corpus, %.0f s to generate + load), oracle/flat_oracle.c oflat_topk_heap, %s, 1 thread"
                      % (n1, rows, dim, metric, t_build, flavour)) if full else
                     ("%d queries over a %d x %d fp32 %s SAMPLE of the keyed corpus in %.1f s (oracle/flat_oracle.c "
                      "oflat_topk_heap, %s), scaled by rows to %d (host has %.0f GB free, %d cpus: full mode needs %.0f GB)"
                      % (n1, n_rows, dim, metric, el, flavour, rows, _mem_available_gb(), _cpus(), need_gb))}
    if checked is not None:
        out["gpu_answers_checked_against_oracle_on_full_corpus"] = checked
    return out


def cpu_config0(n=100_000, dim=128, k=10, nq=1000):
    """BASELINE configs[0] -- the reference's own CPU-runnable case -- on host cores with the same port the
    cpu_baseline leg uses (warm-up 3, 1000 queries, p50 / p95 / QPS).  No GPU involved."""
    import platform
    O, lib, flavour = _oracle_lib()
    data = O.philox_rows(SEED, 0, n, dim)
    qs = O.philox_rows(SEED, QUERY_BASE, nq, dim)
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


# ---- N=1 sub-records ---------------------------------------------------------------------------------------------------
def philox_host_rows(V, first_index, n, dim, vtype=None):
    """Rows first_index .. first_index+n of the keyed Philox corpus as a host array, produced by the PRODUCT's own generator
    (corpus_kernels.hip through a scratch L2 index: no normalisation) -- bench inputs never come from oracle/."""
    vtype = V.VecSimType_FLOAT32 if vtype is None else vtype
    s = V.VecSimIndex(vtype, dim, V.VecSimMetric_L2)
    try:
        assert s.add_philox_rows(SEED, first_index, n, 1) == n
        return s.read_rows(0, n)
    finally:
        s.free()


def encode_freqs_only(docs, freqs, block_entries=100):
    """Synthetic posting list in the reference's FreqsOnly block format (inverted_index/src/codec/freqs_only.rs: qint2
    [delta, freq]; qint/src/lib.rs: one control byte, 2 bits per field = byte length - 1, little-endian minimal bytes; a
    block holds 100 entries, the delta of its first entry is 0).  Pure numpy: the bench's input generator.  Returns the
    upload dict of redisearch_amd.search.Postings.from_flat (tests/test_bench_contract_cpu.py holds it to the oracle's
    block writer byte for byte)."""
    docs = np.ascontiguousarray(docs, np.uint64)
    freqs = np.ascontiguousarray(freqs, np.uint64)
    n = docs.size
    idx = np.arange(n)
    delta = np.zeros(n, np.uint64)
    delta[1:] = docs[1:] - docs[:-1]
    delta[idx % block_entries == 0] = 0
    assert n == 0 or int(delta.max()) <= 0xFFFFFFFF

    def nbytes(v):
        return (1 + (v > 0xFF) + (v > 0xFFFF) + (v > 0xFFFFFF)).astype(np.int64)
    ld, lf = nbytes(delta), nbytes(freqs)
    rec = 1 + ld + lf
    off = np.cumsum(rec) - rec
    total = int(off[-1] + rec[-1]) if n else 0
    out = np.zeros(max(total, 1), np.uint8)
    out[off] = ((ld - 1) | ((lf - 1) << 2)).astype(np.uint8)
    for b in range(4):
        m = ld > b
        out[off[m] + 1 + b] = ((delta[m] >> np.uint64(8 * b)) & np.uint64(0xFF)).astype(np.uint8)
        m = lf > b
        out[off[m] + 1 + ld[m] + b] = ((freqs[m] >> np.uint64(8 * b)) & np.uint64(0xFF)).astype(np.uint8)
    starts = np.arange(0, n, block_entries)
    ends = np.minimum(starts + block_entries, n)
    byte_off = np.concatenate([off[starts], [total]]).astype(np.uint64) if n else np.zeros(1, np.uint64)
    return dict(first=docs[starts], last=docs[ends - 1] if n else docs[:0], num_entries=(ends - starts).astype(np.uint32),
                offset=byte_off, bytes=out[:total], codec=2)   # RSGPU_CODEC_FREQS_ONLY


def encode_full(docs, freqs, masks, off_bytes, block_entries=100):
    """The same list in the reference's Full format (inverted_index/src/codec/full.rs:117-190: qint4 [delta, freq, field
    mask, offsets length] followed by the record's offsets bytes -- the term positions as varint deltas), the default of
    FT.CREATE without NOOFFSETS / NOFIELDS.  off_bytes: the concatenated offsets bytes of all records, record r owning
    freqs[r] of them (every position delta here is below 128: one varint byte each).  Pure numpy, byte-identical to the
    oracle's block writer (tests/test_bench_contract_cpu.py)."""
    docs = np.ascontiguousarray(docs, np.uint64)
    n = docs.size
    idx = np.arange(n)
    delta = np.zeros(n, np.uint64)
    delta[1:] = docs[1:] - docs[:-1]
    delta[idx % block_entries == 0] = 0
    assert n == 0 or int(delta.max()) <= 0xFFFFFFFF
    osz = np.ascontiguousarray(freqs, np.uint64)          # one offsets byte per occurrence
    fields = [delta, np.ascontiguousarray(freqs, np.uint64), np.ascontiguousarray(masks, np.uint64), osz]
    assert int(osz.sum()) == len(off_bytes)

    def nbytes(v):
        return (1 + (v > 0xFF) + (v > 0xFFFF) + (v > 0xFFFFFF)).astype(np.int64)
    lens = [nbytes(f) for f in fields]
    head = 1 + lens[0] + lens[1] + lens[2] + lens[3]
    rec = head + osz.astype(np.int64)
    off = np.cumsum(rec) - rec
    total = int(off[-1] + rec[-1]) if n else 0
    out = np.zeros(max(total, 1), np.uint8)
    ctrl = np.zeros(n, np.int64)
    for i in range(4):
        ctrl |= (lens[i] - 1) << (2 * i)
    out[off] = ctrl.astype(np.uint8)
    at = off + 1
    for i in range(4):
        for b in range(4):
            m = lens[i] > b
            out[at[m] + b] = ((fields[i][m] >> np.uint64(8 * b)) & np.uint64(0xFF)).astype(np.uint8)
        at = at + lens[i]
    if len(off_bytes):
        first_tail = np.cumsum(osz.astype(np.int64)) - osz.astype(np.int64)
        pos = np.repeat(at - first_tail, osz.astype(np.int64)) + np.arange(len(off_bytes))
        out[pos] = off_bytes
    starts = np.arange(0, n, block_entries)
    ends = np.minimum(starts + block_entries, n)
    byte_off = np.concatenate([off[starts], [total]]).astype(np.uint64) if n else np.zeros(1, np.uint64)
    return dict(first=docs[starts], last=docs[ends - 1] if n else docs[:0], num_entries=(ends - starts).astype(np.uint32),
                offset=byte_off, bytes=out[:total], codec=0)   # RSGPU_CODEC_FULL


def verify_answers(index, queries, k, metric, dim, total_rows, which=(0, 1, 2), answers=None, shards=1):
    """The returned neighbours of a few queries against first principles, on the host: every returned row is regenerated from
    the corpus key (oracle.philox_rows: the CPU twin of corpus_kernels.hip), its distance recomputed in fp64 (<= 1e-4 from the
    reply), the reply is sorted, and the K-th beats regenerated probe rows -- 2 048 per query, drawn from EVERY shard's row range
    when the corpus is row-sharded (shard s holds rows [s * total / shards, (s + 1) * total / shards)), so a shard whose winners
    got lost in the exchange shows.  answers: {query index: (labels, scores)} when the caller produced them (the multi-rank
    form: every rank takes part in the collective query, rank 0 checks); else index.topk_query.  Returns (worst error, answers)."""
    import oracle as O
    worst, got = 0.0, {}
    for qi in which:
        q = queries[qi].astype(np.float64)
        ids, sc = answers[qi] if answers is not None else index.topk_query(queries[qi], k).results()
        ids, sc = np.asarray(ids), np.asarray(sc)
        got[qi] = (ids.tolist(), sc.tolist())
        rows = np.stack([O.philox_rows(SEED, int(l) - 1, 1, dim)[0] for l in ids]).astype(np.float64)
        per = total_rows // shards
        n_probe = max(min(2048 // shards, per), 1)
        starts = [sh * per + (12345 + 4096 * qi) % max(per - n_probe, 1) for sh in range(shards)]
        probe = np.concatenate([O.philox_rows(SEED, p0, n_probe, dim) for p0 in starts]).astype(np.float64)
        probe_labels = np.concatenate([np.arange(p0 + 1, p0 + 1 + n_probe) for p0 in starts])

        def dist(x):
            if metric == "l2":
                return np.sum((x - q) ** 2, axis=1)
            dot = x @ q
            if metric == "cosine":
                dot = dot / (np.linalg.norm(x, axis=1) * np.linalg.norm(q))
            return 1.0 - dot
        d = dist(rows)
        worst = max(worst, float(np.max(np.abs(d - sc))))
        assert np.all(np.diff(sc) >= 0), "reply not sorted"
        others = dist(probe)
        labels = set(int(l) for l in ids)
        for lab, dj in zip(probe_labels.tolist(), others):
            assert lab in labels or dj >= sc[-1] - 1e-4, ("a regenerated row beats the returned K-th", qi, lab, dj, sc[-1])
    return worst, got


class no_gc:
    """Timed regions run with Python's cyclic collector off (and a collection right before): with torch and numpy loaded a
    generation-2 pass takes tens of milliseconds, and one landing inside a 1.4 ms query shows up as a 30-100 ms `max_ms`
    that has nothing to do with the library (the lump the round-2 driver run caught in the two-stage extra; none in 11 024
    queries of scripts/diag/two_stage_lumps.py, a process without torch).  .collections = passes that ran regardless."""
    def __enter__(self):
        gc.collect()
        self._was = gc.isenabled()
        gc.disable()
        self._before = sum(g["collections"] for g in gc.get_stats())
        self.collections = 0
        return self

    def __exit__(self, *exc):
        self.collections = sum(g["collections"] for g in gc.get_stats()) - self._before
        if self._was:
            gc.enable()
        return False


def _lat_summary(lat_s):
    lat = np.asarray(lat_s, np.float64) * 1e3
    return {"p50_ms": float(np.percentile(lat, 50)), "p95_ms": float(np.percentile(lat, 95)), "max_ms": float(lat.max())}


def extra_two_stage(lib, V, index, queries, k, steps, warmup):
    """The opt-in two-stage exact scan on the headline index (int8 shadow).  Every query that leaves the two-stage path
    runs the plain fp32 scan (exact, four times the bytes): the library counts each way out (RSGPU_GetTwoStageStats) and
    the record carries the counters next to per-query latencies -- a silent fallback shows up as `fallbacks` > 0 and as a
    p95 near the fp32 scan's latency."""
    import torch
    lib.RSGPU_SetTuning(b"two_stage", 1)
    try:
        for i in range(8):      # warm-up: workspaces at their final size, clocks up
            index.topk_query(queries[i], k)
        ids2, sc2 = index.topk_query(queries[7], k).results()
        lib.RSGPU_SetTuning(b"two_stage", 0)
        ids1, sc1 = index.topk_query(queries[7], k).results()
        lib.RSGPU_SetTuning(b"two_stage", 1)
        same = ids1.tolist() == ids2.tolist() and sc1.tolist() == sc2.tolist()
        torch.cuda.synchronize()
        n2 = max(min(steps, 1000), 200)   # (never fewer than 200 queries: 0.3 s)
        lat = np.zeros(n2)
        V.two_stage_stats(reset=True)
        with no_gc() as g:
            t2 = time.perf_counter()
            for i in range(n2):
                q = queries[(warmup + i) % len(queries)]
                s = time.perf_counter()
                rep = lib.VecSimIndex_TopKQuery(index.ptr, q.ctypes.data_as(C.c_void_p), k, None, V.BY_SCORE)
                lib.VecSimQueryReply_Free(rep)
                lat[i] = time.perf_counter() - s
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t2
        st = V.two_stage_stats()
        out = {"qps": n2 / el2, "ms_per_query": el2 / n2 * 1e3, "queries": n2, "bit_identical_to_fp32_scan": bool(same),
               "fallbacks": st["fallbacks"], "two_stage_stats": st, "python_gc_passes_inside_the_timed_loop": g.collections,
               "what": "opt-in: scan of an int8 shadow with per-row scales (7.72 GB) + error-bounded filter + fp32 "
                       "re-scoring of the survivors; NOT the headline value"}
        out.update(_lat_summary(lat))
        return out
    finally:
        lib.RSGPU_SetTuning(b"two_stage", 0)


def _callers_lib(V):
    """examples/concurrent_callers.c (pthread callers through the plain C ABI) as a ctypes library, built on the spot."""
    import subprocess
    import tempfile
    libdir = os.path.join(ROOT, "redisearch_amd", "lib")
    out = os.path.join(tempfile.mkdtemp(prefix="rs_callers_"), "libconcurrent_callers.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "concurrent_callers.c"), "-L" + libdir, "-lVectorSimilarity",
                           "-Wl,-rpath," + libdir, "-lpthread", "-o", out])
    cl = C.CDLL(out)
    cl.rs_callers_run.restype = C.c_long
    cl.rs_callers_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_double,
                                  C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    return cl


def _hybrid_callers_lib():
    """examples/concurrent_hybrid_callers.c (pthread callers of RSGPU_HybridQuery through the plain C ABI), built on the spot."""
    import subprocess
    import tempfile
    libdir = os.path.join(ROOT, "redisearch_amd", "lib")
    out = os.path.join(tempfile.mkdtemp(prefix="rs_hcallers_"), "libhybrid_callers.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "concurrent_hybrid_callers.c"), "-L" + libdir, "-lVectorSimilarity",
                           "-Wl,-rpath," + libdir, "-lpthread", "-o", out])
    cl = C.CDLL(out)
    cl.rs_hybrid_callers_run.restype = C.c_long
    cl.rs_hybrid_callers_run.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_double)]
    return cl


def run_callers(cl, index, queries, k, threads, seconds, keep_answers=False):
    nq, cap = len(queries), 20000
    lat = np.zeros((threads, cap), np.uint64)
    counts = np.zeros(threads, np.uint64)
    ids = np.zeros((nq, k), np.uint64) if keep_answers else None
    sc = np.zeros((nq, k), np.float64) if keep_answers else None
    el = C.c_double(0)
    q = np.ascontiguousarray(queries)
    total = cl.rs_callers_run(index.ptr, q.ctypes.data_as(C.c_void_p), q.strides[0], nq, k, threads, seconds,
                              lat.ctypes.data_as(C.c_void_p), cap, counts.ctypes.data_as(C.c_void_p),
                              ids.ctypes.data_as(C.c_void_p) if keep_answers else None,
                              sc.ctypes.data_as(C.c_void_p) if keep_answers else None, C.byref(el))
    assert total > 0, "a caller thread failed"
    lats = np.concatenate([lat[t, :int(counts[t])] for t in range(threads)]).astype(np.float64) / 1e9
    return total, el.value, lats, ids, sc


def extra_concurrent_callers(lib, V, index, queries, k, rows, dim, single_qps):
    """WORKERS-n deployments (reference src/util/workers.c:58,104): n threads, each issuing VecSimIndex_TopKQuery calls
    back to back on the SAME plain fp32 index (no shadow in use).  The library's coalescer lets calls that arrive while
    a corpus pass is in flight share the next pass (scan_mq_kernels.hip); the replies must be bit-identical to serial
    ones -- checked here on every query of the run."""
    cl = _callers_lib(V)
    nq = 64
    qs = queries[:nq]
    lib.RSGPU_SetTuning(b"coalesce", 0)
    serial = [index.topk_query(q, k).results() for q in qs]
    total0, el0, lat0, _, _ = run_callers(cl, index, qs, k, 8, 1.5)
    lib.RSGPU_SetTuning(b"coalesce", 1)
    out = {"workload": "%dx%d fp32 plain index (no shadow), n caller threads through VecSimIndex_TopKQuery "
                       "(examples/concurrent_callers.c, pthreads)" % (rows, dim),
           "single_stream_qps": single_qps,
           "eight_threads_without_coalescer": dict(qps=total0 / el0, **_lat_summary(lat0))}
    for threads in (1, 2, 4, 8, 16, 32, 64):
        V.coalesce_stats(reset=True)
        lib.RSGPU_ResetProfile()
        lib.RSGPU_SetProfiling(1)
        total, el, lat, ids, sc = run_callers(cl, index, qs, k, threads, 2.0, keep_answers=True)
        lib.RSGPU_SetProfiling(0)
        st = V.coalesce_stats()
        same = all(ids[i].tolist() == serial[i][0].tolist() and sc[i].tolist() == serial[i][1].tolist() for i in range(nq))
        rec = dict(qps=total / el, x_single_stream=total / el / single_qps, queries=int(total),
                   queries_per_pass=st["queries"] / max(st["passes"], 1), passes=st["passes"],
                   multi_query_passes=st["mq_passes"], bit_identical_to_serial=bool(same), **_lat_summary(lat))
        if st["wide_passes"]:   # more than sixteen callers: one matrix-core filter pass + exact re-scoring for all of them
            launches_w, ms_w, _ = V.scan_profile()
            rec["wide_passes"] = st["wide_passes"]
            rec["queries_per_wide_pass"] = st["wide_queries"] / st["wide_passes"]
            rec["wide_pass_device_ms"] = ms_w / max(launches_w, 1)
            rec["wide_pass_hbm_frac"] = rows * dim * 4 / max(ms_w / max(launches_w, 1), 1e-9) / 1e6 / HBM_PEAK_GBS
        if st["mq_passes"]:
            ms = st["mq_device_ns"] / st["mq_passes"] / 1e6
            rec["multi_query_scan_ms"] = ms
            rec["multi_query_scan_hbm_gbs"] = rows * dim * 4 / ms / 1e6
            rec["multi_query_scan_hbm_frac"] = rows * dim * 4 / ms / 1e6 / HBM_PEAK_GBS
            rec["linger_us_per_pass"] = st["linger_ns"] / 1e3 / max(st["passes"], 1)
        out["%d_threads" % threads] = rec
    out["kernel"] = V.last_mq_scan_kernel()
    # the same callers with the opt-in two-stage exact scan switched on (the index carries the int8 shadow): concurrent
    # K <= 16 calls share multi-query passes over the SHADOW (scan_mq_i8_kernel), survivors re-scored from the fp32 rows --
    # the replies must still be the plain index's, bit for bit
    lib.RSGPU_SetTuning(b"two_stage", 1)
    try:
        sh = {}
        for threads in (1, 8, 16):
            V.coalesce_stats(reset=True)
            V.two_stage_stats(reset=True)
            total, el, lat, ids, sc = run_callers(cl, index, qs, k, threads, 2.0, keep_answers=True)
            st, ts = V.coalesce_stats(), V.two_stage_stats()
            same = all(ids[i].tolist() == serial[i][0].tolist() and sc[i].tolist() == serial[i][1].tolist() for i in range(nq))
            sh["%d_threads" % threads] = dict(qps=total / el, queries=int(total), queries_per_pass=st["queries"] / max(st["passes"], 1),
                                              multi_query_passes=st["mq_passes"], redone_on_the_single_path=st["mq_redo"],
                                              shadow_scan_ms=st["mq_device_ns"] / max(st["mq_passes"], 1) / 1e6,
                                              two_stage_fallbacks=ts["fallbacks"], bit_identical_to_plain_serial=bool(same),
                                              **_lat_summary(lat))
        sh["what"] = "opt-in int8 shadow (+25 % HBM): the two-stage exact scan, coalesced; NOT the headline mode"
        out["with_int8_shadow_two_stage"] = sh
    except Exception as e:  # the extra must never take the record down
        out["with_int8_shadow_two_stage"] = {"error": str(e)[:200]}
    finally:
        lib.RSGPU_SetTuning(b"two_stage", 0)
    out["qps"] = out["8_threads"]["qps"]
    out["x_single_stream"] = out["8_threads"]["x_single_stream"]
    out["p50_ms"] = out["8_threads"]["p50_ms"]
    out["bit_identical_to_serial"] = all(out["%d_threads" % t]["bit_identical_to_serial"] for t in (1, 2, 4, 8, 16, 32, 64))
    out["wide_pass_kernel"] = ("more than 16 queued callers: gemm_qs_f32_kernel (fp32 tiles by DMA, rounded to bf16 on the way to the "
                               "matrix cores, 256 queries stationary) + exact re-scoring from the fp32 rows; no stored shadow")
    return out


def _batched_wall(index, qs, k, ids_of_block, block, calls=3):
    """Wall time per 256-query pass when a caller hands RSGPU_FlatIndex_TopKBatch SEVERAL passes per call (here 4 x 256): the two
    pipeline slots of batch_query.cpp overlap pass b's D2H + host reply building with pass b+1's device work, which one pass
    per call cannot.  Also checks that the block-of-4 call returns, for the queries of `block`, the ids the one-pass call gave."""
    nb, batch, dim = qs.shape
    flat = np.ascontiguousarray(qs.reshape(nb * batch, dim))
    ids4, _, _ = index.topk_batch(flat, k)   # warm (second slot's allocations)
    t0 = time.perf_counter()
    for _ in range(calls):
        ids4, _, _ = index.topk_batch(flat, k)
    el = time.perf_counter() - t0
    same = bool(np.array_equal(ids4[block * batch:(block + 1) * batch], ids_of_block))
    return el / (calls * nb) * 1e3, same


def extra_batched_f32(lib, V, index, rows, dim, want_shadow8_back=False):
    """The rest of K5: 256 queries per corpus pass on a PLAIN fp32 cosine index (RediSearch's default type; an index of its own --
    the headline index carries the two-stage extra's opt-in shadow) through RSGPU_FlatIndex_TopKBatch.  Default route since round
    6: the fp32 rows are quantised to int8 on their way to the int8 matrix cores (gemm_qs_h8r_kernel<.., SRC_F8>; nothing stored);
    beside it the bf16-in-flight form (knob gemm_qs_f8 = 0: gemm_qs_f32_kernel).  Survivors are re-scored exactly: replies must be
    bit-identical to VecSimIndex_TopKQuery."""
    k, batch, reps = 100, 256, 8
    del index
    # a PLAIN index: the headline index was created with the shadow8 knob set (the two-stage extra's shadow) and the knob is read at
    # creation -- left on, this index would carry a stored shadow too and never take the in-flight route (rounds 6's first records
    # measured the bf16 form twice that way)
    lib.RSGPU_SetTuning(b"shadow8", 0)
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_Cosine)
    if want_shadow8_back:
        lib.RSGPU_SetTuning(b"shadow8", 1)
    try:
        idx.reserve(rows)
        idx.add_philox_rows(SEED, 0, rows, 1)
        qs = philox_host_rows(V, QUERY_BASE + 5000, batch * 4, dim).reshape(4, batch, dim)
        idx.topk_batch(qs[0], k)  # allocations, the index-wide quantisation statistics
        V.coalesce_stats(reset=True)
        lib.RSGPU_ResetProfile()
        lib.RSGPU_SetProfiling(1)
        t0 = time.perf_counter()
        for i in range(reps):
            ids, sc, cnt = idx.topk_batch(qs[(i + 1) % 4], k)
        el = time.perf_counter() - t0
        lib.RSGPU_SetProfiling(0)
        launches, ms, _ = V.scan_profile()
        route = int(lib.RSGPU_LastBatchRoute())   # 6: fp32 rows quantised to int8 in flight (rsgpu_ext.h)
        mq = V.coalesce_stats()["mq_passes"]
        dev_ms = ms / max(launches, 1)
        same = True
        for i in (0, 85, 170, 255):
            si, ss = idx.topk_query(qs[reps % 4][i], k).results()
            same &= si.tolist() == ids[i].tolist() and ss.tolist() == sc[i].tolist()
        wall_ms, same4 = _batched_wall(idx, qs, k, ids, reps % 4)
        same &= same4
        bf16_ms, bf16_same, bf16_route = None, None, None
        try:   # the bf16-in-flight form over the same index, for the record (rounds 4-5's route)
            lib.RSGPU_SetTuning(b"gemm_qs_f8", 0)
            idx.topk_batch(qs[0], k)
            lib.RSGPU_ResetProfile()
            lib.RSGPU_SetProfiling(1)
            for i in range(5):
                ids0, sc0, _ = idx.topk_batch(qs[(i + 1) % 4], k)
            lib.RSGPU_SetProfiling(0)
            l0, ms0, _ = V.scan_profile()
            bf16_route = int(lib.RSGPU_LastBatchRoute())   # 2: fp32 rows rounded to bf16 in flight
            bf16_ms = ms0 / max(l0, 1)
            lib.RSGPU_SetTuning(b"gemm_qs_f8", 1)
            idsr, scr, _ = idx.topk_batch(qs[5 % 4], k)
            bf16_same = bool(np.array_equal(ids0, idsr) and np.array_equal(sc0, scr))
        finally:
            lib.RSGPU_SetTuning(b"gemm_qs_f8", 1)
        flops = 2.0 * batch * dim * rows
        return {"workload": "%dx%d fp32 FLAT COSINE top-%d on a plain index (no shadow), batch=%d queries per corpus pass "
                            "(RSGPU_FlatIndex_TopKBatch)" % (rows, dim, k, batch),
                "device_ms_per_pass": dev_ms, "wall_ms_per_pass": wall_ms, "wall_over_device": wall_ms / dev_ms,
                "qps_device": batch / dev_ms * 1e3, "qps_wall": batch / wall_ms * 1e3,
                "qps_wall_one_pass_per_call": reps * batch / el,
                "matrix_core_passes": int(launches), "exact_multi_query_scan_passes_instead": int(mq),
                "hbm_gbs": rows * dim * 4 / dev_ms / 1e6, "hbm_frac": rows * dim * 4 / dev_ms / 1e6 / HBM_PEAK_GBS,
                "mfma_tflops": flops / dev_ms / 1e9, "mfma_frac": flops / dev_ms / 1e9 / MFMA_PEAK_TFLOPS,
                "int8_tops": flops / dev_ms / 1e9, "int8_frac_of_5000_TOPS": flops / dev_ms / 1e9 / 5000.0,
                "kernel": "gemm_qs_h8r_kernel<.., SRC_F8> (round 6: fp32 rows quantised to int8 IN FLIGHT -- four v_fma_f32 + three "
                          "v_perm_b32 per chunk, once per workgroup, register-staged ring -- v_mfma_i32_32x32x32_i8 against 256 "
                          "register-stationary int8 queries; thresholds widened by the Cauchy-Schwarz band of the measured quantisation "
                          "errors) + progressive thresholds + batch_rescore_kernel (the single-query scan's arithmetic) + per-query "
                          "select; HIP events around the whole device pipeline of a pass",
                "bf16_in_flight_pass_device_ms": bf16_ms, "bf16_in_flight_pass_same_replies": bf16_same,
                "route": route, "route_is_int8_in_flight": route == 6, "bf16_in_flight_route": bf16_route,
                "bit_identical_to_single_queries": bool(same),
                "algorithmic_bytes_per_pass": rows * dim * 4}
    finally:
        idx.free()


def extra_batched(lib, V, rows, dim):
    """BASELINE configs[2]: rows x dim fp16 FLAT IP top-100, 256 queries per corpus pass on the matrix cores."""
    k, batch, reps = 100, 256, 10
    idx = V.VecSimIndex(V.VecSimType_FLOAT16, dim, V.VecSimMetric_IP)
    try:
        idx.reserve(rows)
        idx.add_philox_rows(SEED, 0, rows, 1)
        qs = philox_host_rows(V, QUERY_BASE, batch * 4, dim, V.VecSimType_FLOAT16).reshape(4, batch, dim)
        idx.topk_batch(qs[0], k)  # allocations
        lib.RSGPU_ResetProfile()
        lib.RSGPU_SetProfiling(1)
        t0 = time.perf_counter()
        for i in range(reps):
            ids, sc, cnt = idx.topk_batch(qs[(i + 1) % 4], k)
        el = time.perf_counter() - t0
        lib.RSGPU_SetProfiling(0)
        launches, ms, _ = V.scan_profile()
        route = int(lib.RSGPU_LastBatchRoute())   # 5: fp16 rows quantised to int8 in flight (rsgpu_ext.h)
        dev_ms = ms / max(launches, 1)
        # parity: 4 of the 256 queries against the single-query path of the same index -- BIT-IDENTICAL since round 5 (the
        # survivors of the matrix-core passes are re-scored with the single-query scan's arithmetic, as on every other route)
        ok, worst = True, 0.0
        for i in (0, 85, 170, 255):
            si, ss = idx.topk_query(qs[reps % 4][i], k).results()
            ok &= si.tolist() == ids[i].tolist() and ss.tolist() == sc[i].tolist()
            worst = max(worst, float(np.max(np.abs(np.sort(sc[i]) - np.sort(ss)))))
        wall_ms, same4 = _batched_wall(idx, qs, k, ids, reps % 4)
        ok &= same4
        flops = 2.0 * batch * dim * rows
        # the fp16 MFMA pass over the same index (knob gemm_qs_h8 = 0: rounds 1-5's route), for the record
        f16_ms, f16_same = None, None
        try:
            lib.RSGPU_SetTuning(b"gemm_qs_h8", 0)
            idx.topk_batch(qs[0], k)
            lib.RSGPU_ResetProfile()
            lib.RSGPU_SetProfiling(1)
            for i in range(5):
                ids0, sc0, _ = idx.topk_batch(qs[(i + 1) % 4], k)
            lib.RSGPU_SetProfiling(0)
            l0, ms0, _ = V.scan_profile()
            f16_ms = ms0 / max(l0, 1)
            idsr, scr, _ = idx.topk_batch(qs[5 % 4], k)     # (the default route again on the block the loop ended with)
            lib.RSGPU_SetTuning(b"gemm_qs_h8", 5)
            idsr, scr, _ = idx.topk_batch(qs[5 % 4], k)
            f16_same = bool(np.array_equal(ids0, idsr) and np.array_equal(sc0, scr))
        finally:
            lib.RSGPU_SetTuning(b"gemm_qs_h8", 5)
        # opt-in int8 shadow of the same corpus (RSGPU_SetTuning("shadow8") before VecSimIndex_New): the filter passes run
        # on the int8 matrix cores over half the bytes, survivors are re-scored from the fp16 rows -- the replies must be
        # BIT-IDENTICAL to single queries on the fp16 index
        i8 = None
        try:
            idx.free()
            lib.RSGPU_SetTuning(b"shadow8", 1)
            lib.RSGPU_SetTuning(b"two_stage", 1)   # (query-time switch of every shadow; the headline loop keeps it off)
            idx = V.VecSimIndex(V.VecSimType_FLOAT16, dim, V.VecSimMetric_IP)
            lib.RSGPU_SetTuning(b"shadow8", 0)
            idx.reserve(rows)
            idx.add_philox_rows(SEED, 0, rows, 1)
            t0 = time.perf_counter()
            idx.topk_batch(qs[0], k)  # builds the shadow
            build_s = time.perf_counter() - t0
            lib.RSGPU_ResetProfile()
            lib.RSGPU_SetProfiling(1)
            t0 = time.perf_counter()
            for i in range(reps):
                ids8, sc8, cnt8 = idx.topk_batch(qs[(i + 1) % 4], k)
            el8 = time.perf_counter() - t0
            lib.RSGPU_SetProfiling(0)
            l8, ms8, by8 = V.scan_profile()
            d8 = ms8 / max(l8, 1)
            same = True
            for i in (0, 85, 170, 255):
                si, ss = idx.topk_query(qs[reps % 4][i], k).results()
                same &= si.tolist() == ids8[i].tolist() and ss.tolist() == sc8[i].tolist()
            i8 = {"device_ms_per_pass": d8, "qps_device": batch / d8 * 1e3, "qps_wall": reps * batch / el8,
                  "shadow_bytes_read_per_pass": by8 / max(l8, 1), "hbm_gbs_of_shadow_bytes": by8 / max(l8, 1) / d8 / 1e6,
                  "int8_tops": flops / d8 / 1e9, "int8_frac_of_5000_TOPS": flops / d8 / 1e9 / 5000.0,
                  "first_call_incl_shadow_build_s": build_s,
                  "bit_identical_to_single_queries": bool(same),
                  "note": "opt-in (+50 % HBM for an fp16 index); exact: Cauchy-Schwarz band from the actual quantisation-error norms"}
        except Exception as e:  # the extra must never take the headline down
            i8 = {"error": str(e)[:200]}
        finally:
            lib.RSGPU_SetTuning(b"two_stage", 0)
        payload = {"kind": "batched", "rows": rows, "dim": dim, "k": k, "queries": qs[reps % 4].copy(),
                   "ids": ids.copy(), "scores": sc.copy(), "which": (0, 85, 170, 255)}
        return {"workload": "%dx%d fp16 FLAT IP top-%d, batch=%d queries per corpus pass (RSGPU_FlatIndex_TopKBatch)" % (rows, dim, k, batch),
                "int8_shadow_extra": i8, "route": route, "route_is_int8_in_flight": route == 5,
                "device_ms_per_pass": dev_ms, "wall_ms_per_pass": wall_ms, "wall_over_device": wall_ms / dev_ms,
                "qps_device": batch / dev_ms * 1e3, "qps_wall": batch / wall_ms * 1e3, "qps_wall_one_pass_per_call": reps * batch / el,
                "wall_definition": "4 passes (1024 queries) per RSGPU_FlatIndex_TopKBatch call: the call's two pipeline slots overlap "
                                   "host reply building with the next pass; device_ms_per_pass = HIP events around one pass alone",
                "hbm_gbs": rows * dim * 2 / dev_ms / 1e6, "hbm_frac": rows * dim * 2 / dev_ms / 1e6 / HBM_PEAK_GBS,
                "mfma_tflops": flops / dev_ms / 1e9, "mfma_frac": flops / dev_ms / 1e9 / MFMA_PEAK_TFLOPS,
                "kernel": "gemm_qs_h8r_kernel (round 6: fp16 rows quantised to int8 IN FLIGHT -- once per workgroup, register-staged ring, "
                          "v_mfma_i32_32x32x32_i8 against register-stationary int8 queries; nothing stored next to the index; thresholds "
                          "widened by the Cauchy-Schwarz band of the actual quantisation-error maxima) + exact re-scoring of the survivors "
                          "(batch_rescore_kernel: the single-query scan's arithmetic) + per-query select; HIP events around the whole "
                          "device pipeline of a pass",
                "fp16_mfma_pass_device_ms": f16_ms, "fp16_mfma_pass_same_replies": f16_same,
                "int8_tops": flops / dev_ms / 1e9, "int8_frac_of_5000_TOPS": flops / dev_ms / 1e9 / 5000.0,
                "bit_identical_to_single_queries": bool(ok),
                "parity": {"ok": bool(ok and worst == 0.0), "vs": "single-query path, 4 of 256 queries: ids and scores identical "
                                                                   "(replaced by the CPU-oracle check in the cpu_baseline leg)",
                           "max_abs_dist_diff": worst}}, payload
    finally:
        idx.free()


def _term_list(rng, n_docs, df):
    """A term's postings: membership as an independent Bernoulli(df / n_docs) draw per document -- sampled through its gaps
    (geometric), which is the same process as hashing every (term, doc) pair at a hundredth of the draws --
    frequency 1 + Geometric(0.5) capped at 255, one field-mask byte and one position delta (< 128) per occurrence."""
    gaps = rng.geometric(df / n_docs, int(df * 1.02) + 4096)
    docs = np.cumsum(gaps, dtype=np.uint64)
    docs = docs[docs <= n_docs]
    freqs = np.minimum(1 + rng.geometric(0.5, docs.size), 255).astype(np.uint32)
    masks = (1 << rng.integers(0, 8, docs.size)).astype(np.uint32)
    offs = rng.integers(1, 128, int(freqs.sum())).astype(np.uint8)
    return docs, freqs, masks, offs


def _hybrid_stream(lib, S, enc, raws, table, idx, qvecs, n_docs, n_vec, avg, dim, n_a, cycles=3, decoded_bpp=8, modes=("warm", "cold"),
                   concurrent_threads=()):
    """A STREAM of distinct queries: n_a x n_b term pairs over independent lists and a query vector of its
    own per query, issued round-robin so that consecutive queries share neither a list nor candidate rows.  One cycle touches
    every list's decoded arrays (240 MB at 4 + 4 lists) and 16 x 77 MB of rows -- several times the 256 MiB Infinity Cache --
    so every query reads its postings, doc-table lines and rows from HBM.  enc: the uploaded-format dicts of the lists (the
    first n_a are the rank-2 terms).  Returns (record, per-pair results)."""
    n_b = len(enc) - n_a
    pairs = [(i % n_a, n_a + (i + i // n_a) % n_b) for i in range(n_a * n_b)]
    enc_bytes = [len(e["bytes"]) for e in enc]
    out, answers = {}, None
    for mode in modes:
        lib.RSGPU_SetTuning(b"cache_decoded", 1 if mode == "warm" else 0)
        lists = [S.Postings.from_flat(e) for e in enc]
        try:
            def make_queries():
                qq = []
                for qi, (i, j) in enumerate(pairs):
                    dfs = [raws[i][0].size, raws[j][0].size]
                    form = os.environ.get("RSGPU_BENCH_HYBRID_FORM", "both")   # diagnostics (scripts/gpu_prof_hybrid.sh): one branch only
                    qq.append(S.HybridQuery([lists[i], lists[j]], table if form != "knn" else None, "BM25STD", [S.calculate_idf(n_docs, d) for d in dfs],
                                            [S.calculate_idf_bm25(n_docs, d) for d in dfs], [1.0, 1.0], n_docs, avg, top_n=10 if form != "knn" else 0,
                                            index=idx if form != "score" else None, q=qvecs[qi], k=10 if form != "score" else 0))
                return qq
            hqs = make_queries()
            res = []
            for hq in hqs:   # untimed first cycle: allocations, and (warm) every list's one decode
                hq.run()
                res.append(hq.results())
            path = S.hybrid_path()
            walls = []
            with no_gc():
                for _ in range(cycles):
                    for hq in hqs:
                        t0 = time.perf_counter()
                        hq.run()
                        walls.append((time.perf_counter() - t0) * 1e3)
            lib.RSGPU_SetProfiling(1)   # per-stage device times (HIP events; a sync per stage: not the wall figure)
            prof = []
            for hq in hqs:
                hq.run()
                prof.append(S.profile())
            lib.RSGPU_SetProfiling(0)
            same = all(hq.results()["top"][0].tolist() == r["top"][0].tolist() and hq.results()["knn"][0].tolist() == r["knn"][0].tolist()
                       for hq, r in zip(hqs, res))
            tile = float(np.mean([x.get("intersect_ms") or 0.0 for x in prof]))
            red = float(np.mean([x.get("topn_ms") or 0.0 for x in prof]))
            dec = float(np.mean([x.get("decode_ms") or 0.0 for x in prof]))
            rec = {"wall_ms_p50": float(np.percentile(walls, 50)), "wall_ms_p95": float(np.percentile(walls, 95)), "wall_ms_min": min(walls),
                   "queries_timed": len(walls), "qps": 1e3 / float(np.percentile(walls, 50)),
                   "path": "two_launches" if path == 1 else ("general_tile_kernel" if path == 2 else "staged"),
                   "device_ms": {"tile_kernel": tile, "reduce_kernel": red, "decode": dec} if path in (1, 2) else
                                {k_: float(np.mean([x.get(k_) or 0.0 for x in prof])) for k_ in ("intersect_ms", "score_ms", "topn_ms", "knn_ms")},
                   "same_answers_every_cycle": bool(same)}
            if mode == "warm" and concurrent_threads:
                # the reference runs its hybrid iterators on worker threads (src/util/workers.c:58,104): the same stream from
                # T threads at once -- every thread its own argument blocks and device scratch, the lists / table / index shared
                # (round 6: pthread callers through the plain C ABI -- examples/concurrent_hybrid_callers.c, what a C module's worker
                # pool pays; rounds 3-5 drove this leg from Python threads, an interpreter between two calls)
                conc = {}
                try:
                    hcl = _hybrid_callers_lib()
                    sets = [make_queries() for _ in range(4)]         # 64 argument blocks: thread t cycles over t, t + n, ...
                    flat = [hq for st in sets for hq in st]
                    for hq in flat:                                   # untimed: allocations
                        hq.run()
                    blocks = (C.c_void_p * len(flat))(*[C.addressof(h.args) for h in flat])
                    for nt in concurrent_threads:
                        cap = 60000
                        lat = np.zeros((nt, cap), np.uint64)
                        counts = np.zeros(nt, np.uint64)
                        el = C.c_double(0)
                        S.hybrid_coalesce_stats(reset=True)
                        total = hcl.rs_hybrid_callers_run(blocks, len(flat), nt, 0.5, lat.ctypes.data_as(C.c_void_p), cap,
                                                          counts.ctypes.data_as(C.c_void_p), C.byref(el))
                        st = S.hybrid_coalesce_stats()
                        allv = np.concatenate([lat[t, :int(min(counts[t], cap))] for t in range(nt)]).astype(np.float64) / 1e6
                        ok = total > 0
                        for q_, hq in enumerate(flat):
                            r, want = hq.results(), res[q_ % len(res)]
                            ok = ok and r["n_hits"] == want["n_hits"] and r["top"][0].tolist() == want["top"][0].tolist() and \
                                r["top"][1].tolist() == want["top"][1].tolist() and r["knn"][0].tolist() == want["knn"][0].tolist()
                        conc["%d_threads" % nt] = {"qps": total / el.value, "queries": int(total), "p50_ms": float(np.percentile(allv, 50)),
                                                   "p95_ms": float(np.percentile(allv, 95)), "same_answers_as_serial": bool(ok),
                                                   "queries_per_shared_grid": (st["grid_queries"] / st["grids"]) if st["grids"] else None,
                                                   "queries_launched_alone": st["alone"]}
                    del sets, flat
                except Exception as e:   # the extra must never take the headline down
                    conc["error"] = repr(e)[:200]
                rec["concurrent_callers"] = conc
            if mode == "warm":
                answers = res
                # algorithmic bytes of the tile kernel, per query: 4 B per decoded posting of both lists + 12 B per hit
                # (frequency in the second list, doc length, doc score) + a row per hit that has a vector
                hits = np.array([r["n_hits"] for r in res], np.float64)
                n_cand = []
                for (i, j) in pairs[:4]:   # exact candidate counts of four pairs (the staged intersection, read back)
                    h = S.intersect([lists[i], lists[j]])
                    gi, _ = h.read()
                    n_cand.append(int(np.searchsorted(gi, n_vec, side="right")))
                    h.free()
                cand = float(np.mean(n_cand))
                postings = float(np.mean([raws[i][0].size + raws[j][0].size for i, j in pairs]))
                alg = postings * 4 + float(hits.mean()) * 12 + cand * dim * 4
                rec.update({"hits_mean": float(hits.mean()), "candidates_with_vector_mean_of_4_pairs": cand,
                            "tile_kernel_algorithmic_bytes": alg})
                if path in (1, 2) and tile > 0:
                    rec["tile_kernel_gbs"] = alg / tile / 1e6
                    rec["tile_kernel_hbm_frac"] = alg / tile / 1e6 / HBM_PEAK_GBS
                    rec["tile_plus_reduce_hbm_frac"] = alg / (tile + red) / 1e6 / HBM_PEAK_GBS
            else:
                eb = float(np.mean([enc_bytes[i] + enc_bytes[j] for i, j in pairs]))
                rec["encoded_bytes_per_query"] = eb
                if dec > 0:
                    rec["decode_gbs_of_encoded_bytes"] = eb / dec / 1e6
                    rec["decode_hbm_frac_of_encoded_bytes"] = eb / dec / 1e6 / HBM_PEAK_GBS
                    # what the decode kernel moves: the encoded bytes read + the decoded arrays written (decoded_bpp bytes per
                    # posting: doc id + frequency; the Full codec also mask, offsets position, offsets length)
                    wr = float(np.mean([raws[i][0].size + raws[j][0].size for i, j in pairs])) * decoded_bpp
                    rec["decoded_bytes_written_per_query"] = wr
                    rec["decode_gbs_read_plus_written"] = (eb + wr) / dec / 1e6
                    rec["decode_hbm_frac_read_plus_written"] = (eb + wr) / dec / 1e6 / HBM_PEAK_GBS
                if path == 1 and dec + tile > 0:
                    rec["decode_plus_intersect_gbs_of_encoded_bytes"] = eb / (dec + tile) / 1e6
                rec["same_answers_as_warm"] = all(a["top"][0].tolist() == b_["top"][0].tolist() and a["knn"][0].tolist() == b_["knn"][0].tolist()
                                                  and a["n_hits"] == b_["n_hits"] for a, b_ in zip(res, answers))
            out[mode] = rec
        finally:
            lib.RSGPU_SetTuning(b"cache_decoded", 1)
            for x in lists:
                x.free()
    out["pairs"] = len(pairs)
    out["lists"] = {"rank_2_terms": n_a, "rank_4_terms": n_b, "postings": [int(r[0].size) for r in raws], "encoded_bytes": enc_bytes}
    return out, answers, pairs


def _hybrid_general_shapes(lib, S, enc_fo, enc_full, raws, table, idx, qvecs, n_docs, avg, n_a, cycles=3):
    """The query shapes round 3's two-launch form handed back to the staged pipeline, through the general tile kernel (round 4:
    hybrid_tree_tile_kernel, RSGPU_HybridQueryPath == 2) and -- knob hybrid_tree_tiles = 0 -- staged, same process, same
    queries: the hit list wanted (a third launch packs it), a slop-dependent scorer over Full-codec lists (per-hit
    IndexResult_MinOffsetDelta from the term offsets), a (b|c) (a union child), a phrase window (max_slop over the offsets).
    Eight queries per shape, round-robin over distinct lists, decode-cache warm."""
    n_b = len(enc_fo) - n_a
    fo = [S.Postings.from_flat(e) for e in enc_fo]
    fu = [S.Postings.from_flat(e) for e in enc_full]
    out = {}
    try:
        def sc(ix):
            dfs = [raws[t][0].size for t in ix]
            return dict(table=table, scorer=None, idf=[S.calculate_idf(n_docs, d) for d in dfs],
                        bm25_idf=[S.calculate_idf_bm25(n_docs, d) for d in dfs], weight=[1.0] * len(ix), num_docs=n_docs, avg_doc_len=avg,
                        top_n=10)
        shapes = {}
        pairs = [(i % n_a, n_a + (i + i // n_a) % n_b) for i in range(8)]
        mk = []
        for qi, (i, j) in enumerate(pairs):
            a = sc([i, j]); a["scorer"] = "BM25STD"
            mk.append(lambda a=a, i=i, j=j, qi=qi: S.HybridQuery([fo[i], fo[j]], index=idx, q=qvecs[qi], k=10, want_hits=True, **a))
        shapes["hit_list_wanted_freqs_only_bm25std_knn"] = mk
        mk = []
        for qi, (i, j) in enumerate(pairs):
            a = sc([i, j]); a["scorer"] = "TFIDF.DOCNORM"   # (the bench table has no max-frequency column: TFIDF proper would score 0)
            mk.append(lambda a=a, i=i, j=j, qi=qi: S.HybridQuery([fu[i], fu[j]], index=idx, q=qvecs[qi], k=10, **a))
        shapes["tfidf_docnorm_over_full_codec_slop_from_offsets_knn"] = mk
        mk = []
        for qi, (i, j) in enumerate(pairs):
            j2 = n_a + (j - n_a + 1) % n_b
            a = sc([i, j, j2]); a["scorer"] = "BM25STD"
            mk.append(lambda a=a, i=i, j=j, j2=j2, qi=qi: S.HybridTreeQuery(S.OP_INTERSECT, [(S.OP_TERM, 1.0, [fo[i]]), (S.OP_UNION, 1.0, [fo[j], fo[j2]])],
                                                                           index=idx, q=qvecs[qi], k=10, **a))
        shapes["term_and_union_of_two_freqs_only_bm25std_knn"] = mk
        mk = []
        for qi, (i, j) in enumerate(pairs):
            a = sc([i, j]); a["scorer"] = "BM25STD"
            mk.append(lambda a=a, i=i, j=j, qi=qi: S.HybridTreeQuery(S.OP_INTERSECT, [(S.OP_TERM, 1.0, [fu[i]]), (S.OP_TERM, 1.0, [fu[j]])], max_slop=30,
                                                                     index=idx, q=qvecs[qi], k=10, **a))
        shapes["two_terms_max_slop_30_full_codec_bm25std_knn"] = mk
        mk = []
        for qi, (i, j) in enumerate(pairs):   # round 5: `a | b` -- a root union, one tile-kernel pass per child + one reduce
            a = sc([i, j]); a["scorer"] = "BM25STD"
            mk.append(lambda a=a, i=i, j=j, qi=qi: S.HybridTreeQuery(S.OP_UNION, [(S.OP_TERM, 1.0, [fo[i]]), (S.OP_TERM, 1.0, [fo[j]])],
                                                                     index=idx, q=qvecs[qi], k=10, **a))
        shapes["root_union_of_two_terms_freqs_only_bm25std_knn"] = mk
        mk = []
        for qi, (i, j) in enumerate(pairs):   # round 5: `a ((b c) | d)` -- a nested tree (RSGPU_HybridTreeNodesQuery), score folded node by node
            j2, j3 = n_a + (j - n_a + 1) % n_b, n_a + (j - n_a + 2) % n_b
            a = sc([i, j, j2, j3]); a["scorer"] = "BM25STD"
            tree = ("and", 1.0, [("t", 0), ("or", 1.0, [("and", 1.0, [("t", 1), ("t", 2)]), ("t", 3)])])
            mk.append(lambda a=a, i=i, j=j, j2=j2, j3=j3, qi=qi, tree=tree: S.HybridNodesQuery(tree, [fo[i], fo[j], fo[j2], fo[j3]], index=idx,
                                                                                              q=qvecs[qi], k=10, **a))
        shapes["nested_a_and_bc_or_d_freqs_only_bm25std_knn"] = mk
        mk = []
        for qi, (i, j) in enumerate(pairs):   # round 5: `(a|b) (c|d)` -- no term every hit holds: the smaller union drives, a pass per term
            i2, j2 = (i + 1) % n_a, n_a + (j - n_a + 1) % n_b
            a = sc([i, i2, j, j2]); a["scorer"] = "BM25STD"
            mk.append(lambda a=a, i=i, i2=i2, j=j, j2=j2, qi=qi: S.HybridTreeQuery(S.OP_INTERSECT, [(S.OP_UNION, 1.0, [fo[i], fo[i2]]),
                                                                                                   (S.OP_UNION, 1.0, [fo[j], fo[j2]])],
                                                                                  index=idx, q=qvecs[qi], k=10, **a))
        shapes["two_unions_of_two_freqs_only_bm25std_knn"] = mk
        mk = []
        for qi, (i, j) in enumerate(pairs):   # round 6: `a | b` WITH the hit list -- the passes' runs merged by doc id (7 M hits)
            a = sc([i, j]); a["scorer"] = "BM25STD"
            mk.append(lambda a=a, i=i, j=j, qi=qi: S.HybridTreeQuery(S.OP_UNION, [(S.OP_TERM, 1.0, [fo[i]]), (S.OP_TERM, 1.0, [fo[j]])],
                                                                     index=idx, q=qvecs[qi], k=10, want_hits=True, **a))
        shapes["root_union_of_two_terms_hit_list_wanted_freqs_only_bm25std_knn"] = mk
        mk = []
        for qi, (i, j) in enumerate(pairs):   # round 6: `a ((b (c|d)) | e)` -- a union below an intersection below a union: the match folded over the tree
            j2, j3, i2 = n_a + (j - n_a + 1) % n_b, n_a + (j - n_a + 2) % n_b, (i + 1) % n_a
            a = sc([i, j, j2, j3, i2]); a["scorer"] = "BM25STD"
            tree = ("and", 1.0, [("t", 0), ("or", 1.0, [("and", 1.0, [("t", 1), ("or", 1.0, [("t", 2), ("t", 3)])]), ("t", 4)])])
            mk.append(lambda a=a, i=i, j=j, j2=j2, j3=j3, i2=i2, qi=qi, tree=tree: S.HybridNodesQuery(tree, [fo[i], fo[j], fo[j2], fo[j3], fo[i2]], index=idx,
                                                                                                     q=qvecs[qi], k=10, **a))
        shapes["nested_union_below_intersection_below_union_freqs_only_bm25std_knn"] = mk
        mk = []
        for qi, (i, j) in enumerate(pairs):   # round 6: `a ((b c) | d)` over Full-codec lists, TFIDF.DOCNORM -- the per-hit slop through nested children
            j2, j3 = n_a + (j - n_a + 1) % n_b, n_a + (j - n_a + 2) % n_b
            a = sc([i, j, j2, j3]); a["scorer"] = "TFIDF.DOCNORM"
            tree = ("and", 1.0, [("t", 0), ("or", 1.0, [("and", 1.0, [("t", 1), ("t", 2)]), ("t", 3)])])
            mk.append(lambda a=a, i=i, j=j, j2=j2, j3=j3, qi=qi, tree=tree: S.HybridNodesQuery(tree, [fu[i], fu[j], fu[j2], fu[j3]], index=idx,
                                                                                              q=qvecs[qi], k=10, **a))
        shapes["nested_a_and_bc_or_d_full_codec_tfidf_docnorm_slop_knn"] = mk
        for name, makers in shapes.items():
            rec = {}
            answers = {}
            for mode, knob in (("general_kernel", 1), ("staged", 0)):
                lib.RSGPU_SetTuning(b"hybrid_tree_tiles", knob)
                try:
                    hqs = [m() for m in makers]
                    for hq in hqs:
                        hq.run()
                    path = S.hybrid_path()
                    answers[mode] = [hq.results() for hq in hqs]
                    walls = []
                    with no_gc():
                        for _ in range(cycles):
                            for hq in hqs:
                                t0 = time.perf_counter()
                                hq.run()
                                walls.append((time.perf_counter() - t0) * 1e3)
                    rec[mode] = {"wall_ms_p50": float(np.percentile(walls, 50)), "wall_ms_p95": float(np.percentile(walls, 95)), "path": path,
                                 "hits_mean": float(np.mean([r["n_hits"] for r in answers[mode]]))}
                    del hqs
                finally:
                    lib.RSGPU_SetTuning(b"hybrid_tree_tiles", 1)
            rec["same_answers"] = all(x["n_hits"] == y["n_hits"] and x["top"][0].tolist() == y["top"][0].tolist() and
                                      x["top"][1].tolist() == y["top"][1].tolist() and x["knn"][0].tolist() == y["knn"][0].tolist()
                                      for x, y in zip(answers["general_kernel"], answers["staged"]))
            out[name] = rec
    finally:
        for x in fo + fu:
            x.free()
    return out


def extra_hybrid(lib, V, n_docs=50_000_000, n_vec=5_000_000, dim=768, n_a=4, n_b=4):
    """BASELINE configs[4]: 2-term intersection over Zipf postings (50M docs) -> FLAT 5M x 768 ad-hoc KNN top-10 + BM25STD.
    The headline figure is a STREAM of distinct queries (16 term pairs over 4 + 4 independent lists, a query vector each) --
    FreqsOnly postings and, as SURVEY 8(d)'s second variant, the same lists in the Full codec (FT.CREATE's default) -- warm
    and cold; the earlier same-query-repeated figure is kept as `repeat_same_query`.
    Returns (record, payload): the payload holds what the cpu-baseline leg needs to hold the answers to the CPU oracle
    (check_hybrid_with_oracle); nothing here touches oracle/."""
    from redisearch_amd import search as S
    rng = np.random.default_rng(149)
    doc_len = (50 + rng.poisson(150, n_docs + 1)).astype(np.uint32)
    doc_score = np.ones(n_docs + 1, np.float32)
    avg = float(doc_len[1:].mean())
    table = S.DocTable(doc_len, doc_score)
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2)
    try:
        idx.reserve(n_vec)
        idx.add_philox_rows(SEED, 0, n_vec, 1)
        t0 = time.perf_counter()
        raws = [_term_list(rng, n_docs, n_docs * 0.2 / r) for r in [2] * n_a + [4] * n_b]
        enc_fo = [encode_freqs_only(d, f) for d, f, _, _ in raws]
        enc_full = [encode_full(d, f, m, o) for d, f, m, o in raws]
        gen_s = time.perf_counter() - t0
        qvecs = philox_host_rows(V, QUERY_BASE + 100, n_a * n_b, dim)
        fo, ans_fo, pairs = _hybrid_stream(lib, S, enc_fo, raws, table, idx, qvecs, n_docs, n_vec, avg, dim, n_a, concurrent_threads=(2, 4, 8, 16))
        full, ans_full, _ = _hybrid_stream(lib, S, enc_full, raws, table, idx, qvecs, n_docs, n_vec, avg, dim, n_a, decoded_bpp=20)
        codec_same = all(a["top"][0].tolist() == b_["top"][0].tolist() and a["top"][1].tolist() == b_["top"][1].tolist()
                         and a["knn"][0].tolist() == b_["knn"][0].tolist() and a["n_hits"] == b_["n_hits"] for a, b_ in zip(ans_fo, ans_full))
        rep, payload = _hybrid_repeat_same_query(lib, V, S, table, idx, doc_len, doc_score, avg, n_docs, n_vec, dim)
        try:
            general = _hybrid_general_shapes(lib, S, enc_fo, enc_full, raws, table, idx, qvecs, n_docs, avg, n_a)
        except Exception as e:                      # (an extra: the stream record above stands on its own)
            general = {"error": repr(e)}
        try:
            mutated, mut_payload = _hybrid_after_deletes(lib, V, S, enc_fo, raws, table, idx, qvecs, n_docs, n_vec, avg, dim, n_a, fo["warm"])
        except Exception as e:
            mutated, mut_payload = {"error": repr(e)}, None
        w = fo["warm"]
        rec = {"workload": "2-term intersect (Zipf ranks 2 and 4: df 0.1N / 0.05N, %d docs) -> FLAT %dx%d fp32 L2 ad-hoc KNN top-10 + BM25STD "
                           "top-10; a round-robin STREAM over %d distinct term pairs (%d + %d independent lists), a query vector per query"
                           % (n_docs, n_vec, dim, len(pairs), n_a, n_b),
               "wall_ms_per_query": w["wall_ms_p50"], "wall_ms_p95": w["wall_ms_p95"], "qps": w["qps"], "queries_timed": w["queries_timed"],
               "figure": "p50 of %d queries of the distinct-query stream, FreqsOnly postings, decode-cache warm (every list decoded "
                         "once, kept in HBM); consecutive queries share no list and no candidate row -- the cycle's working set is "
                         "several times the Infinity Cache" % w["queries_timed"],
               "path": w["path"],
               "stream_freqs_only": fo, "stream_full_codec": full,
               "general_tile_kernel_shapes": general,
               "after_deletes": mutated,
               "full_codec_answers_equal_freqs_only": bool(codec_same),
               "input_generation_s": gen_s,
               "repeat_same_query": rep,
               "parity": rep.get("parity")}
        # the cpu leg holds two queries of the stream (one per codec) to the CPU oracle, KNN included
        payload["stream"] = [dict(codec=c, raw=[raws[i], raws[j]], q=qvecs[qi], n_vec=n_vec, dim=dim, ans=ans[qi],
                                  idf=[S.calculate_idf(n_docs, raws[t][0].size) for t in (i, j)],
                                  bidf=[S.calculate_idf_bm25(n_docs, raws[t][0].size) for t in (i, j)])
                             for c, ans, qi in (("freqs_only", ans_fo, 0), ("full", ans_full, 5)) for (i, j) in [pairs[qi]]]
        payload["after_deletes"] = mut_payload
        return rec, payload
    finally:
        idx.free()


def _hybrid_after_deletes(lib, V, S, enc_fo, raws, table, idx, qvecs, n_docs, n_vec, avg, dim, n_a, pristine, frac=0.01):
    """VERDICT r04 next 1(b, c): the same stream over the index after it stopped being pristine -- 1 % of the vectors deleted at
    random (VecSimIndex_DeleteVector, one call each, as src/spec.c:3533-3541 issues them) and half as many documents re-added
    under NEW doc ids (an update is delete + a new id, src/indexer.c:179-190).  The label -> row table lives in HBM since round 5
    (csrc/label_table.hpp): the queries stay on the two-launch tile path.  Reported: the first delete's latency (rounds 1-4: a
    hash map of every row under the writer lock), the mean of the rest, the stream's p50 against the pristine index's.
    LAST leg of the hybrid extra: it mutates the index."""
    rng = np.random.default_rng(151)
    n_del = int(n_vec * frac)
    victims = rng.choice(n_vec, n_del, replace=False).astype(np.uint64) + 1
    mode0 = idx.label_table()
    t0 = time.perf_counter()
    assert idx.delete_vector(int(victims[0])) == 1
    first_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    for lab in victims[1:].tolist():
        idx.delete_vector(lab)
    rest_ms = (time.perf_counter() - t0) * 1e3 / max(n_del - 1, 1)
    n_add = n_del // 2
    new_labels = n_vec + 1 + 2 * np.arange(n_add, dtype=np.uint64)     # (every other new doc id: documents without a vector between)
    fresh_first = 1 << 33
    chunk = 4096
    t0 = time.perf_counter()
    for a0 in range(0, n_add, chunk):
        block = philox_host_rows(V, fresh_first + a0, min(chunk, n_add - a0), dim)
        for j in range(block.shape[0]):
            idx.add_vector(block[j], int(new_labels[a0 + j]))
    add_ms = (time.perf_counter() - t0) * 1e3 / max(n_add, 1)
    rec, answers, pairs = _hybrid_stream(lib, S, enc_fo, raws, table, idx, qvecs, n_docs, n_vec + 2 * n_add, avg, dim, n_a, modes=("warm",))
    w = rec["warm"]
    out = {"deleted": n_del, "re_added_under_new_doc_ids": n_add, "label_table_before": mode0, "label_table_after": idx.label_table(),
           "label_table_legend": "0 identity labels, 1 device table (csrc/label_table.hpp), 2 host hash maps",
           "first_delete_ms": first_ms, "delete_ms_mean_of_rest": rest_ms, "add_vector_ms_mean": add_ms,
           "stream_warm": w, "path": w["path"], "wall_ms_p50": w["wall_ms_p50"],
           "pristine_wall_ms_p50": pristine["wall_ms_p50"], "p50_over_pristine": w["wall_ms_p50"] / pristine["wall_ms_p50"]}
    i, j = pairs[0]
    payload = dict(raw=[raws[i], raws[j]], q=qvecs[0], n_vec=n_vec, dim=dim, ans=answers[0], victims=victims, new_labels=new_labels,
                   fresh_first=fresh_first, idf=[S.calculate_idf(n_docs, raws[t][0].size) for t in (i, j)],
                   bidf=[S.calculate_idf_bm25(n_docs, raws[t][0].size) for t in (i, j)])
    return out, payload


def _hybrid_repeat_same_query(lib, V, S, table, idx, doc_len, doc_score, avg, n_docs, n_vec, dim):
    """The round-1..3 figure of configs[4], kept for continuity: ONE term pair (ranks 2 and 4, membership by an independent
    hash, seed 49) and ONE query vector, the same query 40 times back to back -- its whole working set (30 MB of decoded
    postings, 77 MB of rows, the gather lines) stays in the 256 MiB Infinity Cache, so this is a cache-resident figure, not
    an HBM one.  Also: the staged pipeline behind the same call (A/B), the cold form, the parity payload."""
    rng = np.random.default_rng(49)
    raw = []
    for r in (2, 4):
        docs = np.flatnonzero(rng.random(n_docs + 1) < 0.2 / r).astype(np.uint64)
        docs = docs[docs > 0]
        freqs = np.minimum(1 + rng.geometric(0.5, docs.size), 255).astype(np.uint32)
        raw.append((docs, freqs))
    if True:
        q = philox_host_rows(V, QUERY_BASE, 1, dim)[0]
        g = [S.Postings.from_flat(encode_freqs_only(d, f)) for d, f in raw]
        idf = [S.calculate_idf(n_docs, d.size) for d, _ in raw]
        bidf = [S.calculate_idf_bm25(n_docs, d.size) for d, _ in raw]

        def pipeline():
            h = S.intersect(g)
            h.score(table, "BM25STD", idf, bidf, [1.0, 1.0], n_docs, avg, want_scores=False)
            ti, ts = h.topn(10)
            ki, kd = h.knn_rerank(idx, q, 10)
            return h, ti, ts, ki, kd
        h, ti, ts, ki, kd = pipeline()
        staged = []
        for _ in range(6):
            t0 = time.perf_counter()
            h2, *_ = pipeline()
            staged.append((time.perf_counter() - t0) * 1e3)
            h2.free()

        hq = S.HybridQuery(g, table, "BM25STD", idf, bidf, [1.0, 1.0], n_docs, avg, top_n=10, index=idx, q=q, k=10)

        def fused():
            hq.run()
            return hq.results()
        r = fused()
        path = S.hybrid_path()        # 1: two launches (hybrid_kernels.hip), 0: the staged pipeline
        walls = []
        for _ in range(40):           # the bare C call (argument block prepared once, as a C caller's is)
            t0 = time.perf_counter()
            hq.run()
            walls.append((time.perf_counter() - t0) * 1e3)
        # the same call through the staged pipeline (ten kernels on two streams), same process: the A/B of the two-launch form
        lib.RSGPU_SetTuning(b"hybrid_tiles", 0)
        try:
            hq.run()
            r_staged = hq.results()
            walls_staged = []
            for _ in range(40):
                t0 = time.perf_counter()
                hq.run()
                walls_staged.append((time.perf_counter() - t0) * 1e3)
            lib.RSGPU_SetProfiling(1)
            hq.run()
            prof_staged = S.profile()
            lib.RSGPU_SetProfiling(0)
        finally:
            lib.RSGPU_SetTuning(b"hybrid_tiles", 1)
        staged_same = all(r[k_][j].tolist() == r_staged[k_][j].tolist() for k_ in ("top", "knn") for j in (0, 1)) and r["n_hits"] == r_staged["n_hits"]
        fused_ok = (r["n_hits"] == len(h) and r["top"][0].tolist() == ti.tolist() and r["top"][1].tolist() == ts.tolist()
                    and r["knn"][0].tolist() == ki.tolist() and r["knn"][1].tolist() == kd.tolist())
        # COLD: the lists' decoded arrays are not kept (cache_decoded = 0): every query decodes both posting lists from
        # their wire-format bytes first -- SURVEY.md 8(d) defines the intersect stage on ENCODED bytes
        lib.RSGPU_SetTuning(b"cache_decoded", 0)
        try:
            g_cold = [S.Postings.from_flat(encode_freqs_only(d, f)) for d, f in raw]
            hq_cold = S.HybridQuery(g_cold, table, "BM25STD", idf, bidf, [1.0, 1.0], n_docs, avg, top_n=10, index=idx, q=q, k=10)
            hq_cold.run()
            rc = hq_cold.results()
            cold = []
            for _ in range(20):
                t0 = time.perf_counter()
                hq_cold.run()
                cold.append((time.perf_counter() - t0) * 1e3)
            lib.RSGPU_SetProfiling(1)
            hq_cold.run()
            prof_cold = S.profile()
            lib.RSGPU_SetProfiling(0)
            cold_ok = rc["top"][0].tolist() == ti.tolist() and rc["knn"][0].tolist() == ki.tolist()
        finally:
            lib.RSGPU_SetTuning(b"cache_decoded", 1)
        enc_bytes = sum(x.num_bytes for x in g)
        # per-stage device times (HIP events; adds a sync per stage, so not the wall figure)
        lib.RSGPU_SetProfiling(1)
        fused()
        prof = S.profile()
        lib.RSGPU_SetProfiling(0)
        gi, gf = h.read()
        adhoc = idx.adhoc_ctx(q)
        seam_ok = bool(np.array_equal(adhoc.get_exact_distances(ki), kd))   # the per-label ad-hoc seam gives the same distances
        n_cand = int(np.searchsorted(gi, n_vec, side="right"))
        # decode of both lists per query: its own stage in the two-launch form (events around the decode launches), the
        # difference of the intersect stages in the staged one
        dec_ms = (prof_cold.get("decode_ms") or 0.0) if path == 1 else max((prof_cold.get("intersect_ms") or 0) - (prof.get("intersect_ms") or 0), 0.0)
        n_ent = [x.num_entries for x in g]
        rec = {"workload": "2-term intersect (Zipf df 0.1N / 0.05N, %d docs) -> FLAT %dx%d fp32 L2 ad-hoc KNN top-10 + BM25STD top-10" % (n_docs, n_vec, dim),
               "wall_ms_per_query": float(np.percentile(walls, 50)), "qps": 1e3 / float(np.percentile(walls, 50)),
               "wall_ms_p95": float(np.percentile(walls, 95)), "wall_ms_min": min(walls), "queries_timed": len(walls),
               "figure": "p50 of %d back-to-back queries, decode-cache warm (posting lists decoded once, kept in HBM)" % len(walls),
               "cold": {"wall_ms_per_query": float(np.percentile(cold, 50)), "wall_ms_p95": float(np.percentile(cold, 95)),
                        "what": "cache_decoded = 0: every query decodes both lists from their encoded bytes (%d B) first (eight lanes per block, from the sync points the lists' first decode left behind)" % enc_bytes,
                        "decode_plus_intersect_device_ms": dec_ms + (prof_cold.get("intersect_ms") or 0) if path == 1 else prof_cold.get("intersect_ms"),
                        "decode_device_ms": dec_ms,
                        "decode_gbs_of_encoded_bytes": enc_bytes / max(dec_ms, 1e-6) / 1e6,
                        "decode_plus_intersect_gbs_of_encoded_bytes": enc_bytes / max((dec_ms + (prof_cold.get("intersect_ms") or 0)) if path == 1 else (prof_cold.get("intersect_ms") or 1e-9), 1e-9) / 1e6,
                        "same_answers": bool(cold_ok)},
               "encoded_posting_bytes": enc_bytes,
               "postings": n_ent, "hits": len(gi),
               "entry_point": ("RSGPU_HybridQuery, two launches: hybrid_tile_kernel (probe + scores + KNN distances + per-tile winners) and "
                               "hybrid_reduce_kernel, one stream sync" if path == 1 else
                               "RSGPU_HybridQuery (one call, two stream syncs; score/top-N and KNN branches on two streams)"),
               "path": "two_launches" if path == 1 else "staged",
               "staged_pipeline_same_process": {"wall_ms_per_query": float(np.percentile(walls_staged, 50)), "wall_ms_p95": float(np.percentile(walls_staged, 95)),
                                                "stage_device_ms": {k_: prof_staged.get(k_) for k_ in ("intersect_ms", "score_ms", "topn_ms", "knn_ms")},
                                                "bit_identical_to_the_two_launch_answers": bool(staged_same),
                                                "what": "the same call with hybrid_tiles = 0: intersection written out, score / top-N and KNN branches on two streams (ten kernels)"},
               "wall_ms_stage_by_stage_entry_points": min(staged[1:]),
               "candidates_with_vector": n_cand,
               "stage_device_ms": ({"tile_kernel_ms": prof.get("intersect_ms"), "reduce_kernel_ms": prof.get("topn_ms")} if path == 1 else
                                   {k_: prof.get(k_) for k_ in ("intersect_ms", "score_ms", "topn_ms", "knn_ms")}),
               "stage_gbs": ({"tile kernel: 4 B per DECODED posting of both lists + 12 B per hit (frequency, doc table) + dim*4 B per candidate with a vector":
                              (sum(n_ent) * 4 + len(gi) * 12 + n_cand * dim * 4) / max(prof.get("intersect_ms") or 1e-9, 1e-9) / 1e6,
                              "tile kernel, in encoded bytes of both lists (SURVEY 8d's unit)": enc_bytes / max(prof.get("intersect_ms") or 1e-9, 1e-9) / 1e6}
                             if path == 1 else
                             {"intersect, warm (4 B per DECODED posting of both lists)": sum(n_ent) * 4 / max(prof.get("intersect_ms") or 1e-9, 1e-9) / 1e6,
                              "intersect, warm, in encoded bytes of both lists (SURVEY 8d's unit)": enc_bytes / max(prof.get("intersect_ms") or 1e-9, 1e-9) / 1e6,
                              "score (20 B read + 20 B written per hit)": len(gi) * 40 / max(prof.get("score_ms") or 1e-9, 1e-9) / 1e6,
                              "knn gather (dim*4 B per candidate with a vector)": n_cand * dim * 4 / max(prof.get("knn_ms") or 1e-9, 1e-9) / 1e6}),
               "parity": {"ok": bool(fused_ok and seam_ok), "vs": "fused == stage-by-stage entry points; KNN distances equal the per-label ad-hoc "
                                                                 "seam's (the CPU-oracle check runs in the cpu_baseline leg)"}}
        payload = dict(raw=raw, doc_len=doc_len, doc_score=doc_score, idf=idf, bidf=bidf, n_docs=n_docs, avg=avg,
                       gi=gi, gf=gf, ti=ti, ts=ts, ok=bool(fused_ok and seam_ok))
        for x in g + g_cold:
            x.free()
        return rec, payload


def check_batched_with_oracle(p):
    """cpu_baseline leg: the batched pass's answers against the CPU oracle over the same 10M fp16 rows (regenerated on the host):
    the same bar as tests/test_gpu_fullsize.py -- the replies ARE the single-query scan's (re-scored exactly since round 5), so
    what separates them from the oracle's scalar fp32 loop is north_star's fp32 tolerance: ids identical except members that tie
    rank K within 1e-4 (relative to max(1, |d|)), distances within 1e-4."""
    import oracle as O
    rows, dim, k = p["rows"], p["dim"], p["k"]
    if _mem_available_gb() < rows * dim * 2 * 2.2 / 1e9 + 8:
        return {"ok": None, "skipped": "host memory", "cpu_oracle_ms": None}
    o = O.FlatIndex(O.F16, dim, O.IP)
    step = 2_500_000
    for a0 in range(0, rows, step):
        o.add_bulk(O.philox_rows(SEED, a0, min(step, rows - a0), dim, O.F16), a0 + 1)
    ok, worst, diff = True, 0.0, 0
    t0 = time.perf_counter()
    for qi in p["which"]:
        oi, os_ = o.topk(p["queries"][qi], k)
        gset, oset = set(p["ids"][qi].tolist()), set(oi.tolist())
        nqb = o.normalized_query(p["queries"][qi])
        for lab in gset ^ oset:
            ok &= abs(o.distance_from(int(lab), nqb) - os_[-1]) <= 1e-4 * max(1.0, abs(os_[-1]))
        diff = max(diff, len(gset ^ oset))
        worst = max(worst, float(np.max(np.abs(np.sort(p["scores"][qi]) - np.sort(os_)) / np.maximum(1.0, np.abs(np.sort(os_))))))
    t_ms = (time.perf_counter() - t0) * 1e3 / len(p["which"])
    return {"ok": bool(ok and diff <= 2 and worst <= 1e-4), "cpu_oracle_ms": t_ms,
            "vs": "CPU oracle on the full %dx%d fp16 corpus, %d of 256 queries: ids identical except true fp32 ties at rank K (<= 1e-4 "
                  "relative), distances within 1e-4 relative" % (rows, dim, len(p["which"])),
            "max_rel_dist_diff": worst, "max_symmetric_difference": diff}


def check_hybrid_with_oracle(p):
    """cpu_baseline leg: the hybrid extra's answers against the CPU oracle on the same inputs (the oracle's own block
    writer re-encodes the lists from the raw (doc, freq) arrays: an independent path to the same postings)."""
    import oracle as O
    lists = []
    for docs, freqs in p["raw"]:
        ii = O.InvertedIndex(O.C_FREQS_ONLY)
        ii.add_many(docs, freqs)
        lists.append(ii)
    t0 = time.perf_counter()
    oi, of, _ = O.intersect(lists)
    t_int = (time.perf_counter() - t0) * 1e3
    sel = oi.astype(np.int64)
    os_ = O.score_flat("BM25STD", of, p["doc_len"][sel], np.ones(len(sel)), p["doc_score"][sel], p["idf"], p["bidf"], [1.0, 1.0], 1.0,
                       p["n_docs"], p["avg"])
    order = np.lexsort((oi, -os_))[:10]
    ok = (p["gi"].tolist() == oi.tolist() and p["gf"].tolist() == of.tolist() and p["ti"].tolist() == oi[order].tolist()
          and bool(np.allclose(p["ts"], os_[order], rtol=1e-12, atol=0)))
    # two queries of the distinct-query stream (one per codec) against the oracle DIRECTLY, the two-launch KNN answer included:
    # oracle intersection -> BM25STD top-10 in the reference's order; candidates with a vector -> their rows regenerated on the
    # host -> the oracle's FLAT L2 top-10 over exactly those rows
    stream = []
    for sp in p.get("stream", []):
        ls = []
        for docs, freqs, _, _ in sp["raw"]:
            ii = O.InvertedIndex(O.C_FREQS_ONLY)
            ii.add_many(docs, freqs)
            ls.append(ii)
        si, sf, _ = O.intersect(ls)
        sel2 = si.astype(np.int64)
        idf, bidf = sp["idf"], sp["bidf"]
        ss = O.score_flat("BM25STD", sf, p["doc_len"][sel2], np.ones(len(sel2)), p["doc_score"][sel2], idf, bidf, [1.0, 1.0], 1.0,
                          p["n_docs"], p["avg"])
        o10 = np.lexsort((si, -ss))[:10]
        a = sp["ans"]
        top_ok = a["n_hits"] == len(si) and a["top"][0].tolist() == si[o10].tolist() and bool(np.allclose(a["top"][1], ss[o10], rtol=1e-12, atol=0))
        cand = si[si <= sp["n_vec"]]
        rows = np.concatenate([O.philox_rows(SEED, int(l) - 1, 1, sp["dim"]) for l in cand]) if len(cand) else np.zeros((0, sp["dim"]), np.float32)
        fo = O.FlatIndex(O.F32, sp["dim"], O.L2)
        fo.add_bulk(rows, 1)
        ki, kd = fo.topk(sp["q"], 10)
        knn_ids = cand[ki.astype(np.int64) - 1]
        knn_ok = a["knn"][0].tolist() == knn_ids.tolist() and bool(np.all(np.abs(a["knn"][1] - kd) <= 1e-4 + 1e-5 * np.abs(kd)))
        stream.append({"codec": sp["codec"], "top_n_ok": bool(top_ok), "knn_ok": bool(knn_ok), "hits": int(len(si)), "candidates": int(len(cand)),
                       "knn_max_abs_diff": float(np.max(np.abs(a["knn"][1] - kd))) if len(kd) == len(a["knn"][1]) else None})
        ok = ok and top_ok and knn_ok
    mut = None
    mp = p.get("after_deletes")
    if mp:   # the stream's first query over the MUTATED index: deleted labels have no vector, re-added doc ids have a fresh one
        ls = []
        for docs, freqs, _, _ in mp["raw"]:
            ii = O.InvertedIndex(O.C_FREQS_ONLY)
            ii.add_many(docs, freqs)
            ls.append(ii)
        si, sf, _ = O.intersect(ls)
        sel2 = si.astype(np.int64)
        ss = O.score_flat("BM25STD", sf, p["doc_len"][sel2], np.ones(len(sel2)), p["doc_score"][sel2], mp["idf"], mp["bidf"], [1.0, 1.0], 1.0,
                          p["n_docs"], p["avg"])
        o10 = np.lexsort((si, -ss))[:10]
        a = mp["ans"]
        top_ok = a["n_hits"] == len(si) and a["top"][0].tolist() == si[o10].tolist() and bool(np.allclose(a["top"][1], ss[o10], rtol=1e-12, atol=0))
        old_c = si[(si <= mp["n_vec"]) & ~np.isin(si, mp["victims"])]
        new_c = si[np.isin(si, mp["new_labels"])]
        rows_old = [O.philox_rows(SEED, int(l) - 1, 1, mp["dim"]) for l in old_c]
        rows_new = [O.philox_rows(SEED, mp["fresh_first"] + int((int(l) - mp["n_vec"] - 1) // 2), 1, mp["dim"]) for l in new_c]
        cand = np.concatenate([old_c, new_c])
        fo = O.FlatIndex(O.F32, mp["dim"], O.L2)
        if len(cand):
            fo.add_bulk(np.concatenate(rows_old + rows_new), 1)
        ki, kd = fo.topk(mp["q"], 10) if len(cand) else (np.zeros(0, np.uint64), np.zeros(0))
        order = np.lexsort((cand[ki.astype(np.int64) - 1], kd))        # (the oracle ranks by its own labels; equal distances by doc id)
        knn_ids = cand[ki.astype(np.int64) - 1][order]
        knn_ok = a["knn"][0].tolist() == knn_ids.tolist() and bool(np.all(np.abs(a["knn"][1] - kd[order]) <= 1e-4 + 1e-5 * np.abs(kd[order])))
        mut = {"top_n_ok": bool(top_ok), "knn_ok": bool(knn_ok), "hits": int(len(si)), "candidates": int(len(cand)),
               "candidates_re_added": int(len(new_c)), "candidates_dropped_by_deletes": int(np.sum((si <= mp["n_vec"]) & np.isin(si, mp["victims"])))}
        ok = ok and top_ok and knn_ok
    return {"ok": bool(ok and p["ok"]), "cpu_oracle_intersect_ms": t_int, "stream_queries_vs_oracle": stream, "after_deletes_query_vs_oracle": mut,
            "vs": "CPU oracle: intersection ids/freqs identical, BM25STD top-10 identical (scores rtol 1e-12), KNN distances equal "
                  "the per-label ad-hoc seam's; fused == stage-by-stage; two queries of the stream (FreqsOnly, Full): hit count, "
                  "BM25STD top-10 and the KNN top-10 (ids identical, distances within 1e-4 + 1e-5 |d|) against the oracle directly"}


def preflight(lib, V, n_devices, rows=100_000, dim=64, k=10, n_shards=None):
    """Every visible device once, before anything long runs on them: a FLAT index on each device alone (allocation, the
    Philox kernel, a scan, the select's pinned-memory write), then ONE handle sharded over all of them answering through the
    host merge and through the RCCL exchange (ncclCommInitAll + ncclAllGather + merge kernel), all against the same
    one-device answers.  Raises with the device index on the first failure."""
    import torch
    q = np.random.default_rng(1).uniform(-1, 1, (4, dim)).astype(np.float32)
    out = {"devices": n_devices, "per_device_ok": []}
    ref = None
    for d in range(n_devices):
        try:
            torch.cuda.set_device(d)
            idx = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2)
            assert idx.add_philox_rows(SEED, 0, rows, 1) == rows
            ans = [idx.topk_query(x, k).results() for x in q]
            idx.free()
            if ref is None:
                ref = ans
            assert all(a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist() for a, b in zip(ans, ref)), "answers differ from device 0's"
            out["per_device_ok"].append(d)
        except Exception as e:
            raise RuntimeError("pre-flight failed on device %d: %r" % (d, e))
    torch.cuda.set_device(0)
    n_shards = n_devices if n_shards is None else n_shards
    oversubscribed = n_shards > n_devices          # (RSGPU_BENCH_OVERSUBSCRIBE: several shards per device -- no communicator)
    if n_shards > 1:
        lib.RSGPU_SetTuning(b"shards", n_shards)
        try:
            sh = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2)
        finally:
            lib.RSGPU_SetTuning(b"shards", 0)
        try:
            assert sh.add_philox_rows(SEED, 0, rows, 1) == rows
            for name, knob in (("host_merge", 0),) + ((("rccl_exchange", 1),) if not oversubscribed else ()):
                lib.RSGPU_SetTuning(b"shard_exchange", knob)
                try:
                    ans = [sh.topk_query(x, k).results() for x in q]
                    assert all(a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist() for a, b in zip(ans, ref)), name
                    out[name] = "ok"
                except Exception as e:
                    raise RuntimeError("pre-flight failed in the %d-shard %s: %r / %s" % (n_shards, name, e, V.last_error()))
            if oversubscribed:
                out["rccl_exchange"] = "skipped: %d shards on %d device(s) (RCCL wants one device per rank)" % (n_shards, n_devices)
            st = (C.c_uint64 * 3)()
            lib.RSGPU_ShardedIndex_GetRcclStats(lib.RSGPU_ShardedIndex_FromHandle(sh.ptr), st, 0)
            out["rccl_ranks"] = int(st[2])
        finally:
            lib.RSGPU_SetTuning(b"shard_exchange", 0)
            sh.free()
    return out


def extra_collective_rccl(lib, V, index, queries, k, dev, steps=40):
    """The RCCL exchange of the multi-GPU path on whatever this run has -- at N = 1 a ONE-rank communicator: the local top-k
    goes through the same C code the N-rank run executes (H2D of the winners, ncclAllGather, merge kernel, pinned-memory
    result), so the line carries what the exchange costs next to a query and that it returns the plain answers."""
    from redisearch_amd.sharded import ShardComm
    sc = ShardComm(index, k, dev)
    try:
        same = True
        for i in range(3):
            labels, scores = sc.query(queries[i])
            wi, ws = index.topk_query(queries[i], k).results()
            same &= labels.tolist() == wi.tolist() and scores.tolist() == ws.tolist()
        sc.stats(reset=True)
        t0 = time.perf_counter()
        for i in range(steps):
            sc.query(queries[(3 + i) % len(queries)])
        el = time.perf_counter() - t0
        n, ns = sc.stats()
        return {"kind": sc.kind, "ranks": sc.world, "us_per_query": ns / max(n, 1) / 1e3, "queries": int(n), "qps": steps / el,
                "payload_bytes_per_rank": k * 16, "same_answers_as_VecSimIndex_TopKQuery": bool(same),
                "note": "one-rank communicator on a one-GPU run: the N-rank code path with a degenerate collective"}
    finally:
        sc.free()


# stdout carries ONE JSON line and nothing else: libraries that print there (RCCL's version banner at communicator creation,
# under torch.distributed as much as under RSGPU_ShardComm_Init) are sent to stderr for the duration of the run -- at the file
# descriptor, where they write -- and the line itself goes out on the real stdout at the end.
_REAL_STDOUT = None


def _quiet_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    sys.stdout.flush()
    if _REAL_STDOUT is not None:
        os.dup2(_REAL_STDOUT, 1)
    print(json.dumps(obj), flush=True)


# The ONE stdout line stays small enough for the driver's capture (round 5's 21.8 KB line fell out of it): the contract
# keys, `roofline`, `cpu_baseline`, `collective` and a `summary` of SCALARS for the sub-records.  The whole record (every
# ladder, every per-shape table, every note) goes to gpurun_out/bench_full.json and, behind a prefix, to stderr.
LINE_LIMIT = 8192
FULL_RECORD = os.path.join(ROOT, "gpurun_out", "bench_full.json")


def _g(d, *path, default=None):
    for p in path:
        if not isinstance(d, dict) or p not in d:
            return default
        d = d[p]
    return d


def _r(x, nd=4):
    if isinstance(x, float):
        return float("%.*g" % (nd + 2, x))
    return x


def _scalars(d):
    return {k: _r(v) for k, v in d.items() if v is not None and isinstance(v, (int, float, bool))}


def summarise_extras(cfg):
    """Scalars only, one flat object per sub-record; a sub-record that failed keeps its error text (cut)."""
    s = {}
    for name in ("two_stage_exact_scan_extra", "concurrent_callers", "collective_rccl", "collective_host_merge", "batched_mfma",
                 "batched_mfma_f32", "hybrid", "first_delete_on_the_headline_index"):
        e = cfg.get(name)
        if isinstance(e, dict) and "error" in e:
            s[name] = {"error": str(e["error"])[:160]}
    ts = cfg.get("two_stage_exact_scan_extra")
    if isinstance(ts, dict) and "qps" in ts:
        s["two_stage"] = _scalars({"qps": ts.get("qps"), "p50_ms": ts.get("p50_ms"), "fallbacks": ts.get("fallbacks"),
                                   "bit_identical": ts.get("bit_identical_to_fp32_scan")})
    cc = cfg.get("concurrent_callers")
    if isinstance(cc, dict) and "error" not in cc:
        rec = {}
        for key, val in cc.items():
            if key.endswith("_threads") and isinstance(val, dict) and "qps" in val:
                rec["qps_" + key.split("_")[0]] = _r(val["qps"])
        rec["bit_identical_to_serial"] = all(v.get("bit_identical_to_serial", True) for v in cc.values() if isinstance(v, dict))
        s["callers"] = rec
    for name, short in (("batched_mfma", "cfg3_f16"), ("batched_mfma_f32", "cfg3_f32")):
        b = cfg.get(name)
        if isinstance(b, dict) and "device_ms_per_pass" in b:
            s[short] = _scalars({"device_ms_per_pass": b.get("device_ms_per_pass"), "wall_ms_per_pass": b.get("wall_ms_per_pass"),
                                 "qps_device": b.get("qps_device"), "qps_wall": b.get("qps_wall"), "hbm_frac": b.get("hbm_frac"),
                                 "mfma_frac": b.get("mfma_frac"), "fp16_mfma_pass_ms": b.get("fp16_mfma_pass_device_ms"),
                                 "bf16_in_flight_pass_ms": b.get("bf16_in_flight_pass_device_ms"),
                                 "bit_identical": b.get("bit_identical_to_single_queries"),
                                 "route": b.get("route"),
                                 "parity_ok": _g(b, "parity", "ok"),
                                 "int8_shadow_device_ms": _g(b, "int8_shadow_extra", "device_ms_per_pass")})
    h = cfg.get("hybrid")
    if isinstance(h, dict) and "error" not in h:
        rec = {"warm_p50_ms": _g(h, "stream_freqs_only", "warm", "wall_ms_p50"), "cold_p50_ms": _g(h, "stream_freqs_only", "cold", "wall_ms_p50"),
               "full_codec_warm_p50_ms": _g(h, "stream_full_codec", "warm", "wall_ms_p50"),
               "full_codec_cold_p50_ms": _g(h, "stream_full_codec", "cold", "wall_ms_p50"),
               "tile_kernel_us": (_g(h, "repeat_same_query", "stage_device_ms", "tile_kernel_ms") or 0) * 1e3 or None,
               "reduce_kernel_us": (_g(h, "repeat_same_query", "stage_device_ms", "reduce_kernel_ms") or 0) * 1e3 or None,
               "tile_kernel_hbm_frac": _g(h, "repeat_same_query", "tile_kernel_hbm_frac"),
               "after_deletes_p50_over_pristine": _g(h, "after_deletes", "p50_over_pristine"),
               "parity_ok": _g(h, "parity", "ok")}
        lad = _g(h, "stream_freqs_only", "warm", "concurrent_callers")
        if isinstance(lad, dict):
            for key, val in lad.items():
                if isinstance(val, dict) and "qps" in val:
                    rec["qps_%s" % key] = val["qps"]
            rec["callers_same_answers"] = all(v.get("same_answers_as_serial", False) for v in lad.values() if isinstance(v, dict))
            rec["queries_per_shared_grid_16_threads"] = _g(lad, "16_threads", "queries_per_shared_grid")
        shapes = h.get("general_tile_kernel_shapes")
        if isinstance(shapes, dict):
            rec["shapes"] = len(shapes)
            rec["shapes_same_answers"] = all(v.get("same_answers", False) for v in shapes.values() if isinstance(v, dict))
            rec["shapes_on_tile_path"] = sum(1 for v in shapes.values() if isinstance(v, dict) and _g(v, "general_kernel", "path") in (1, 2))
        s["cfg5_hybrid"] = _scalars(rec)
    fd = cfg.get("first_delete_on_the_headline_index")
    if isinstance(fd, dict) and "first_delete_ms" in fd:
        s["first_delete_ms"] = _r(fd["first_delete_ms"])
    return s


def compact_line(out):
    """The driver's line from the full record: contract keys verbatim, config / roofline / cpu_baseline cut to what the
    contract names, sub-records as scalars."""
    cfg = out.get("config", {})
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data") if k in out}
    ver = cfg.get("verify") or {}
    line["config"] = {k: _r(cfg[k]) for k in ("workload", "rows_per_gpu", "dim", "k", "metric", "corpus_rows_total", "parallelism", "launch",
                                              "global_qps_on_sharded_corpus", "p50_ms", "p95_ms") if k in cfg}
    line["config"]["parallelism"] = str(line["config"].get("parallelism", ""))[:120]
    line["config"]["verify"] = {k: ver[k] for k in ("ok", "max_abs_err_vs_fp64", "queries", "skipped", "error") if k in ver}
    rf = out.get("roofline", {})
    line["roofline"] = {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches", "avg_kernel_ms",
                                           "algorithmic_bytes_per_launch", "kernel_source_sha256_16", "frac_min_over_devices",
                                           "frac_max_over_devices") if k in rf}
    if "traffic_source" in rf:
        line["roofline"]["traffic_source"] = rf["traffic_source"].split(" ")[0]
    cpu = out.get("cpu_baseline")
    if cpu is not None:
        c = {k: cpu[k] for k in ("value", "unit", "cores", "kind", "p50_ms", "host_cpus") if k in cpu}
        c["sample"] = str(cpu.get("sample", ""))[:200]
        if isinstance(cpu.get("eight_threads"), dict):
            c["eight_threads"] = {k: cpu["eight_threads"][k] for k in ("value", "cores") if k in cpu["eight_threads"]}
        chk = cpu.get("gpu_answers_checked_against_oracle_on_full_corpus")
        if isinstance(chk, dict):
            c["gpu_ids_identical_to_oracle"] = chk.get("ids_identical")
            c["gpu_max_abs_score_diff"] = chk.get("max_abs_score_diff")
        line["cpu_baseline"] = c
    col = out.get("collective")
    if col is not None:
        line["collective"] = {k: (_r(v) if not isinstance(v, str) else v[:100]) for k, v in col.items()
                              if k in ("kind", "timed_exchange", "ranks", "us_per_query", "queries", "payload_bytes_per_rank")}
    line["summary"] = summarise_extras(cfg)
    line["full_record"] = "gpurun_out/bench_full.json (+ stderr)"
    text = json.dumps(line)
    if len(text) >= LINE_LIMIT:      # never lose the headline to an oversized sub-record again
        line["summary"] = {"dropped": "summary exceeded the line budget; see full_record"}
        text = json.dumps(line)
    assert len(text) < LINE_LIMIT, len(text)
    return line


def emit_record(out):
    try:
        os.makedirs(os.path.dirname(FULL_RECORD), exist_ok=True)
        with open(FULL_RECORD, "w") as f:
            json.dump(out, f, indent=1)
    except OSError as e:
        print("bench: could not write %s: %r" % (FULL_RECORD, e), file=sys.stderr)
    print("bench full record: " + json.dumps(out), file=sys.stderr, flush=True)   # prefixed: no stderr line parses as the JSON line
    emit(compact_line(out))


def main():
    a = parse()
    _quiet_stdout()
    if a.cpu_config0:
        emit(cpu_config0())
        return
    import torch
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    env_world = os.environ.get("WORLD_SIZE")
    ranks_mode = env_world is not None and (int(env_world) > 1 or os.environ.get("RSGPU_BENCH_FORCE_DIST") == "1")
    world = int(env_world) if ranks_mode else 1
    if ranks_mode:
        assert world == a.gpus, "under torch.distributed.run: --nproc-per-node must equal --gpus"
    inproc = (not ranks_mode) and a.gpus > 1       # plain `python bench.py --gpus N`: one process, N devices
    n_shards = a.gpus
    if inproc:
        assert torch.cuda.device_count() >= a.gpus or os.environ.get("RSGPU_BENCH_OVERSUBSCRIBE") == "1", \
            "--gpus %d but %d devices visible (RSGPU_BENCH_OVERSUBSCRIBE=1 places several shards per device)" % (a.gpus, torch.cuda.device_count())
    rank = int(os.environ.get("RANK", "0")) if ranks_mode else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if ranks_mode else 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if ranks_mode:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from redisearch_amd import vecsim as V
    lib = V.load()   # (oracle/ is imported by the cpu_baseline leg only: verification of the answers + the CPU timing)
    pre = None
    if a.preflight or inproc:   # every visible device once, before 30 GB per device go up
        pre = preflight(lib, V, torch.cuda.device_count() if a.preflight else min(a.gpus, torch.cuda.device_count()),
                        n_shards=None if a.preflight else a.gpus)
        if a.preflight:
            emit({"preflight": pre})
            return
    for kv in a.tuning:
        key, val = kv.split("=")
        assert lib.RSGPU_SetTuning(key.encode(), int(val)) == 0, kv
    two_stage = any(kv.replace(" ", "") in ("shadow16=1", "shadow8=1") for kv in a.tuning)
    single = a.gpus == 1 and not ranks_mode
    want_two_stage_extra = single and not two_stage and not a.no_extras and not a.no_two_stage_extra and a.metric == "cosine"
    if want_two_stage_extra:   # the index also carries the int8 shadow; switched OFF for the timed headline loop
        lib.RSGPU_SetTuning(b"shadow8", 1)
        lib.RSGPU_SetTuning(b"two_stage", 0)

    rows, dim, k = a.rows, a.dim, a.k
    vmetric = {"cosine": V.VecSimMetric_Cosine, "l2": V.VecSimMetric_L2, "ip": V.VecSimMetric_IP}[a.metric]
    # ---- corpus: global row i (label i+1) = Philox(SEED; i); rank / shard r holds rows [r*rows, (r+1)*rows)
    if inproc:
        # the ordinary VecSim handle over N device shards ("shards" knob): everything below -- VecSimIndex_TopKQuery
        # included -- is the code a single-GPU run executes; the library fans out and merges (sharded_index.cpp)
        lib.RSGPU_SetTuning(b"shards", n_shards)
        lib.RSGPU_SetTuning(b"shard_replicas", int(a.replicas))
        index = V.VecSimIndex(V.VecSimType_FLOAT32, dim, vmetric)
        lib.RSGPU_SetTuning(b"shards", 0)
        lib.RSGPU_SetTuning(b"shard_replicas", 0)
        total_rows = rows if a.replicas else rows * n_shards
        index.reserve(total_rows)
        assert index.add_philox_rows(SEED, 0, total_rows, 1) == total_rows   # split in contiguous runs over the shards
        assert index.index_size() == total_rows
        assert lib.RSGPU_ShardedIndex_NumShards(lib.RSGPU_ShardedIndex_FromHandle(index.ptr)) == n_shards
    else:
        index = V.VecSimIndex(V.VecSimType_FLOAT32, dim, vmetric)
        index.reserve(rows)
        index.add_philox_rows(SEED, rank * rows, rows, rank * rows + 1)
        assert index.index_size() == rows
        total_rows = rows * world
    queries = philox_host_rows(V, QUERY_BASE, 1000, dim)   # the product's own generator, read back to the host

    if ranks_mode:
        # the exchange in C: ncclAllGather + merge kernel (shard_comm.cpp); torch.distributed only hands rank 0's unique id
        # to the other ranks and provides the barriers around the timed region
        from redisearch_amd.sharded import ShardComm
        sharded = ShardComm(index, k, dev)

    def one_query(i):
        q = queries[i % len(queries)]
        if ranks_mode:
            labels, _ = sharded.query(q)   # per-shard top-k -> RCCL all-gather -> merge (C)
            return len(labels)
        rep = lib.VecSimIndex_TopKQuery(index.ptr, q.ctypes.data_as(C.c_void_p), k, None, V.BY_SCORE)
        n = lib.VecSimQueryReply_Len(rep)
        lib.VecSimQueryReply_Free(rep)
        return n

    def barrier():
        if dist is not None:
            dist.barrier()
        for d in range(torch.cuda.device_count() if inproc else 1):
            torch.cuda.synchronize(d if inproc else local_rank)

    # `python bench.py --gpus N` in ONE process: the timed exchange is the north star's -- ONE ncclAllGather of the per-shard
    # top-k + a merge kernel (knob shard_exchange = 1) -- whenever RCCL can form a communicator (one device per shard); the
    # K-way host merge of rounds 2-4 is timed beside it (`collective_host_merge`).  Several shards on one device (tests,
    # RSGPU_BENCH_OVERSUBSCRIBE=1) cannot form one: the host merge stays the headline and the record says so.
    exchange_used = None
    if inproc and not a.replicas:
        exchange_used = "host_merge"
        if torch.cuda.device_count() >= a.gpus and os.environ.get("RSGPU_BENCH_EXCHANGE", "rccl") == "rccl":
            lib.RSGPU_SetTuning(b"shard_exchange", 1)
            try:
                assert one_query(0) == k
                exchange_used = "rccl"
            except Exception as e:
                lib.RSGPU_SetTuning(b"shard_exchange", 0)
                exchange_used = "host_merge (RCCL refused: %r / %s)" % (e, V.last_error())
    for i in range(a.warmup):
        assert one_query(i) == k
    if inproc:
        ex = (C.c_uint64 * 2)()
        lib.RSGPU_ShardedIndex_GetExchangeStats(lib.RSGPU_ShardedIndex_FromHandle(index.ptr), ex, 1)
        rst = (C.c_uint64 * 3)()
        lib.RSGPU_ShardedIndex_GetRcclStats(lib.RSGPU_ShardedIndex_FromHandle(index.ptr), rst, 1)
    if ranks_mode:
        sharded.stats(reset=True)
    lib.RSGPU_ResetProfile()
    lib.RSGPU_SetProfiling(1)
    lat = np.zeros(a.steps)
    barrier()
    with no_gc():
        t0 = time.perf_counter()
        for i in range(a.steps):
            s = time.perf_counter()
            one_query(a.warmup + i)
            lat[i] = time.perf_counter() - s
        barrier()
        elapsed = time.perf_counter() - t0
    lib.RSGPU_SetProfiling(0)
    launches, kern_ms, kern_bytes = V.scan_profile()
    kernel_name = V.last_scan_kernel()
    # the exchange step of the timed queries (N > 1): what it is, how many parties, what it cost per query
    collective = None
    if inproc and not a.replicas and exchange_used == "rccl":
        lib.RSGPU_ShardedIndex_GetRcclStats(lib.RSGPU_ShardedIndex_FromHandle(index.ptr), rst, 0)
        collective = {"kind": "in-process: ncclCommInitAll over the shard devices; per query ONE ncclAllGather of k*16 B per rank + "
                              "merge kernel (shard_comm.cpp), issued by the shard workers -- the exchange of the TIMED loop",
                      "timed_exchange": "rccl", "ranks": int(rst[2]), "us_per_query": rst[1] / max(rst[0], 1) / 1e3, "queries": int(rst[0]),
                      "payload_bytes_per_rank": k * 16}
        assert int(rst[2]) == n_shards, "the communicator spans %d ranks, %d shards" % (int(rst[2]), n_shards)
    elif inproc and not a.replicas:
        lib.RSGPU_ShardedIndex_GetExchangeStats(lib.RSGPU_ShardedIndex_FromHandle(index.ptr), ex, 0)
        collective = {"kind": "in-process: every shard's last kernel writes its top-k into pinned host memory; K-way host merge "
                              "by (score, label) (sharded_index.cpp merge_replies) -- no device collective inside one process",
                      "timed_exchange": exchange_used,
                      "ranks": n_shards, "us_per_query": ex[1] / max(ex[0], 1) / 1e3, "queries": int(ex[0]),
                      "payload_bytes_per_rank": k * 16}
    elif ranks_mode:
        ex_n, ex_ns = sharded.stats()
        collective = {"kind": sharded.kind, "ranks": world, "us_per_query": ex_ns / max(ex_n, 1) / 1e3, "queries": int(ex_n),
                      "payload_bytes_per_rank": k * 16}

    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- after the timed region: verification and the N=1 sub-records ------------------------------------------------
    verify, extras, gpu_answers = None, {}, None
    # per-device scan profile of the timed loop (ranks mode: one device per rank) and, in the multi-rank form, the answers of
    # three verification queries -- collective calls: EVERY rank issues them, rank 0 checks them below
    per_device, rank_answers = None, None
    if ranks_mode:
        mine = {"rank": rank, "launches": int(launches), "kernel_ms": float(kern_ms), "bytes": float(kern_bytes)}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_device = gathered
        if not a.no_cpu_baseline:
            rank_answers = {qi: sharded.query(queries[qi]) for qi in (0, 1, 2)}
    if rank == 0:
        try:
            if a.no_cpu_baseline:   # the host-side regeneration is the oracle's Philox twin: part of the CPU leg
                verify = {"skipped": "--no-cpu-baseline (the verification uses the CPU twin of the corpus generator)"}
            else:
                shards_v = world if ranks_mode else (n_shards if (inproc and not a.replicas) else 1)
                worst, gpu_answers = verify_answers(index, queries, k, a.metric, dim, total_rows, answers=rank_answers, shards=shards_v)
                verify = {"queries": 3, "returned_rows_regenerated_on_host": 3 * k, "max_abs_err_vs_fp64": worst,
                          "ok": bool(worst <= 1e-4), "kth_beats_regenerated_probe_rows": 3 * max(2048 // shards_v, 1) * shards_v,
                          "probe_rows_drawn_from_every_shard": shards_v,
                          "answers_from": ("the merged global top-k of the RCCL exchange (every rank took part; rank 0 regenerates the rows "
                                           "of ANY shard from the corpus key)" if ranks_mode else
                                           "VecSimIndex_TopKQuery on the handle the timed loop used" + (" (exchange: %s)" % exchange_used if exchange_used else ""))}
        except Exception as e:
            verify = {"ok": False, "error": repr(e)}
    if inproc and not a.replicas and exchange_used == "rccl":
        # the same queries through the K-way host merge of rounds 2-4, beside the RCCL exchange the timed loop used
        try:
            lib.RSGPU_SetTuning(b"shard_exchange", 0)
            for i in range(3):
                one_query(i)
            hl = np.zeros(max(a.steps, 1))
            t0 = time.perf_counter()
            for i in range(a.steps):
                s_ = time.perf_counter()
                one_query(a.warmup + i)
                hl[i] = time.perf_counter() - s_
            el_h = time.perf_counter() - t0
            extras["collective_host_merge"] = {"kind": "K-way host merge of the shards' pinned top-k lists (sharded_index.cpp merge_replies)",
                                               "global_qps": a.steps / el_h, "p50_ms": float(np.percentile(hl, 50) * 1e3),
                                               "p95_ms": float(np.percentile(hl, 95) * 1e3)}
        except Exception as e:
            extras["collective_host_merge"] = {"error": repr(e), "last_error": V.last_error()}
        finally:
            lib.RSGPU_SetTuning(b"shard_exchange", 1)
    elif inproc and not a.replicas:
        # the same queries through the RCCL exchange (knob shard_exchange = 1: ncclCommInitAll over the shard devices, one
        # ncclAllGather + merge kernel per query) -- reported next to the host merge the timed loop used
        try:
            lib.RSGPU_SetTuning(b"shard_exchange", 1)
            for i in range(3):
                one_query(i)
            st = (C.c_uint64 * 3)()
            hnd = lib.RSGPU_ShardedIndex_FromHandle(index.ptr)
            lib.RSGPU_ShardedIndex_GetRcclStats(hnd, st, 1)
            rl = np.zeros(max(a.steps, 1))
            t0 = time.perf_counter()
            for i in range(a.steps):
                s_ = time.perf_counter()
                one_query(a.warmup + i)
                rl[i] = time.perf_counter() - s_
            el_r = time.perf_counter() - t0
            lib.RSGPU_ShardedIndex_GetRcclStats(hnd, st, 0)
            extras["collective_rccl"] = {"kind": "in-process: ncclCommInitAll over the shard devices; per query ONE ncclAllGather of k*16 B per "
                                                 "rank + merge kernel (shard_comm.cpp), issued by the shard workers",
                                         "ranks": int(st[2]), "queries": int(st[0]), "global_qps": a.steps / el_r,
                                         "p50_ms": float(np.percentile(rl, 50) * 1e3), "p95_ms": float(np.percentile(rl, 95) * 1e3),
                                         "payload_bytes_per_rank": k * 16}
        except Exception as e:
            extras["collective_rccl"] = {"error": repr(e), "last_error": V.last_error()}
        finally:
            lib.RSGPU_SetTuning(b"shard_exchange", 0)
    if inproc and not a.replicas and not a.no_extras and not a.no_callers_extra:
        # concurrent callers on the sharded handle: every shard's worker answers up to eight callers per pass over its rows
        try:
            cl = _callers_lib(V)
            rec = {}
            for threads in (1, 8):
                V.coalesce_stats(reset=True)
                total, el, clat, _, _ = run_callers(cl, index, queries[:64], k, threads, 2.0)   # (clat: NOT the headline's lat)
                st = V.coalesce_stats()
                rec["%d_threads" % threads] = dict(qps=total / el, queries_per_shard_pass=st["queries"] / max(st["passes"], 1),
                                                   **_lat_summary(clat))
            rec["x_one_caller"] = rec["8_threads"]["qps"] / rec["1_threads"]["qps"]
            extras["concurrent_callers"] = rec
        except Exception as e:
            extras["concurrent_callers"] = {"error": repr(e)}
    if single and rank == 0 and not a.no_extras:
        if want_two_stage_extra:
            try:
                extras["two_stage_exact_scan_extra"] = extra_two_stage(lib, V, index, queries, k, a.steps, a.warmup)
            except Exception as e:  # an extra must never cost the headline line
                extras["two_stage_exact_scan_extra"] = {"error": repr(e)}
        if not a.no_callers_extra:
            try:
                extras["concurrent_callers"] = extra_concurrent_callers(lib, V, index, queries, k, rows, dim, a.steps / elapsed)
            except Exception as e:
                extras["concurrent_callers"] = {"error": repr(e)}
        try:
            extras["collective_rccl"] = extra_collective_rccl(lib, V, index, queries, k, dev)
        except Exception as e:
            extras["collective_rccl"] = {"error": repr(e)}
        if not a.no_batched_extra and a.metric == "cosine":
            try:
                extras["batched_mfma_f32"] = extra_batched_f32(lib, V, index, rows, dim, want_shadow8_back=want_two_stage_extra)
            except Exception as e:
                extras["batched_mfma_f32"] = {"error": repr(e)}
    cpu = None
    if single and rank == 0 and not a.no_cpu_baseline:
        try:
            cpu = cpu_baseline(dim, k, rows, a.cpu_sample_rows, a.metric, a.cpu_baseline_mode, gpu_answers)
        except Exception as e:  # the baseline leg must never cost the measured line
            cpu = {"value": None, "unit": "queries/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}
    if single and rank == 0 and not a.no_extras:
        # the first DeleteVector on the headline index (identity labels end: the device label table appears, csrc/label_table.hpp --
        # rounds 1-4 built a hash map of all 10 M rows under the writer lock here), a second one, and a query after them
        try:
            t0 = time.perf_counter()
            r1 = index.delete_vector(rows // 2)
            d1 = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            r2 = index.delete_vector(rows // 3)
            d2 = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            n_after = one_query(0)
            dq = (time.perf_counter() - t0) * 1e3
            extras["first_delete_on_the_headline_index"] = {"rows": rows, "first_delete_ms": d1, "second_delete_ms": d2, "deleted": [int(r1), int(r2)],
                                                            "label_table_after": index.label_table(), "query_after_ms": dq, "results": int(n_after)}
        except Exception as e:
            extras["first_delete_on_the_headline_index"] = {"error": repr(e)}
        index.free()
        lib.RSGPU_SetTuning(b"shadow8", 0)
        lib.RSGPU_ReleaseWorkspaces()
        torch.cuda.empty_cache()
        for name, fn, skip in (("batched_mfma", lambda: extra_batched(lib, V, rows, dim), a.no_batched_extra),
                               ("hybrid", lambda: extra_hybrid(lib, V), a.no_hybrid_extra)):
            if skip:
                continue
            try:
                t0 = time.perf_counter()
                res = fn()
                payload = None
                if isinstance(res, tuple):
                    res, payload = res
                extras[name] = res
                extras[name]["bench_wall_s"] = time.perf_counter() - t0
                if payload is not None and cpu is not None:   # cpu_baseline leg: the extra's answers vs the CPU oracle
                    try:
                        if payload.get("kind") == "batched":
                            chk = check_batched_with_oracle(payload)
                            extras[name]["cpu_oracle_ms_per_query"] = chk.pop("cpu_oracle_ms")
                            if chk.get("ok") is not None:
                                extras[name]["parity"] = chk
                        else:
                            chk = check_hybrid_with_oracle(payload)
                            extras[name]["cpu_oracle_intersect_ms"] = chk.pop("cpu_oracle_intersect_ms")
                            extras[name]["parity"] = chk
                    except Exception as e:
                        extras[name]["parity"] = {"ok": False, "error": repr(e)}
            except Exception as e:
                extras[name] = {"error": repr(e)}

    if rank == 0:
        n_gpus = a.gpus
        global_qps = a.steps / elapsed
        avg_kernel_s = (kern_ms / 1e3) / max(launches, 1)
        achieved = (kern_bytes / max(launches, 1)) / avg_kernel_s / 1e9 if launches else 0.0
        scale = 1 if (inproc and a.replicas) else n_gpus
        par = ("single GPU" if n_gpus == 1 else
               "%d full replicas in one process behind the plain VecSim handle, one caller thread (replica mode)" % n_gpus if (inproc and a.replicas) else
               "row-sharded x%d in ONE process behind the plain VecSim handle (\"shards\" knob: worker thread per device; exchange: %s)" % (n_gpus, exchange_used) if inproc else
               "row-sharded x%d, one rank per GPU: ncclAllGather of per-shard top-k issued from C + merge kernel on every rank" % n_gpus)
        out = {
            "metric": "KNN queries/sec + p50 latency, 10M×768 fp32 FLAT top-10, 1/2/4/8 GPU",  # BASELINE.json's metric
            "value": global_qps * scale,
            "unit": "queries/s" if (n_gpus == 1 or (inproc and a.replicas)) else "shard-scans/s (queries/s x %d shards of %d rows)" % (n_gpus, rows),
            "n_gpus": n_gpus,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (keyed Philox4x32-10 corpus, seed %d: regenerable row by row)" % SEED,
            "config": {
                "workload": "%dx%d fp32 FLAT %s top-%d per GPU, single-query stream via VecSimIndex_TopKQuery"
                            % (rows, dim, a.metric.upper(), k),
                "value_definition": "10M-row shard scans per second summed over the GPUs (= n_gpus x the global QPS on the "
                                    "row-sharded corpus; at n_gpus=1 it IS the QPS); p50/p95 latency below",
                "rows_per_gpu": rows, "dim": dim, "k": k, "metric": a.metric.upper(),
                "corpus_rows_total": total_rows,
                "parallelism": par,
                "launch": "torch.distributed.run" if ranks_mode else "single process",
                "global_qps_on_sharded_corpus": global_qps,
                "p50_ms": float(np.percentile(lat, 50) * 1e3), "p95_ms": float(np.percentile(lat, 95) * 1e3),
                "verify": verify,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": kernel_name + (" over the low-precision shadow (two-stage exact scan)" if two_stage else ""),
                "kernel_source": "RSGPU_GetLastScanKernel (reported by the library at launch)",
                "launches": int(launches), "avg_kernel_ms": avg_kernel_s * 1e3,
                "algorithmic_bytes_per_launch": kern_bytes / max(launches, 1),
            },
        }
        if per_device:
            fr = []
            for d_ in per_device:
                l_ = max(d_["launches"], 1)
                ach = (d_["bytes"] / l_) / ((d_["kernel_ms"] / 1e3) / l_) / 1e9 if d_["launches"] else 0.0
                fr.append({"rank": d_["rank"], "launches": d_["launches"], "avg_kernel_ms": d_["kernel_ms"] / l_, "achieved_gbs": ach,
                           "frac": ach / HBM_PEAK_GBS})
            out["roofline"]["per_device"] = fr
            out["roofline"]["frac_min_over_devices"] = min(x["frac"] for x in fr)
            out["roofline"]["frac_max_over_devices"] = max(x["frac"] for x in fr)
            out["roofline"]["note"] = "achieved / frac above are rank 0's device; per_device lists every rank's own scan profile"
        elif inproc:
            out["roofline"]["note"] = ("one process, %d devices: the scan profile is the mean over the launches of ALL shards (the library keeps one "
                                       "profile per process); the torch.distributed.run form reports every device separately" % n_gpus)
        out["config"].update(extras)
        if pre is not None:
            out["config"]["preflight"] = pre
        if collective is not None:
            out["collective"] = collective
            out["config"]["multi_gpu_note"] = ("no 2/4/8-GPU hardware curve has been measured by the builder (1-GPU boxes only): "
                                               "the multi-shard path is exercised with several shards on one device")
        # HBM traffic of the scan kernel: a committed rocprofv3 --pmc pass of this same command (bench.py cannot read
        # PMCs itself); used only when it was taken for the same shape, the same kernel instantiation AND the very
        # source of the kernel (sha256 of scan_kernels.hip + scan_ops.hpp recorded by scripts/gpu_prof.sh): a change
        # inside the kernel that keeps its template arguments must not inherit a stale figure
        src_hash = scan_source_hash()
        out["roofline"]["kernel_source_sha256_16"] = src_hash
        for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("scan_pmc_hbm_traffic.json")), reverse=True):
            if two_stage:
                break
            p = json.load(open(os.path.join(ROOT, "profiles", name)))
            same_kernel = p.get("kernel", "").split(" grid")[0] == kernel_name.split(" grid")[0]
            if p.get("rows") == rows and p.get("dim") == dim and same_kernel and p.get("kernel_source_sha256_16") == src_hash:
                out["roofline"]["traffic"] = p["traffic_bytes_per_launch"]
                out["roofline"]["traffic_source"] = "profiles/%s (FETCH_SIZE x2 + WRITE_SIZE, KB->B; same kernel source hash)" % name
                break
        if out["roofline"]["traffic"] is None:
            out["roofline"]["traffic_note"] = "no committed PMC pass matches this kernel's source hash: run scripts/gpu_prof.sh"
        if cpu is not None:
            out["cpu_baseline"] = cpu
        emit_record(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
