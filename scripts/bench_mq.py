#!/usr/bin/env python3
"""Device time of the multi-query scan (scan_mq_kernels.hip) per pass, by number of queries in the pass, next to the
single-query scan on the same corpus; checks bit-identity of every pass against single queries.
    python scripts/bench_mq.py [--rows N] [--dim D] [--type f32|f16|bf16] [--metric cosine|l2|ip] [--tuning k=v ...]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redisearch_amd import vecsim as V  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--type", default="f32")
    ap.add_argument("--metric", default="cosine")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--tuning", action="append", default=[])
    ap.add_argument("--ab", action="append", default=[], help="knob=v1,v2,...: repeat the measurement for each value, interleaved")
    ap.add_argument("--variants", default="", help="k=v,k2=v2;k=v3,k2=v4: knob settings measured in turn, interleaved (several knobs per variant)")
    a = ap.parse_args()
    lib = V.load()
    for kv in a.tuning:
        key, val = kv.split("=")
        assert lib.RSGPU_SetTuning(key.encode(), int(val)) == 0, kv
    vt = {"f32": V.VecSimType_FLOAT32, "f16": V.VecSimType_FLOAT16, "bf16": V.VecSimType_BFLOAT16}[a.type]
    esz = 4 if a.type == "f32" else 2
    vm = {"cosine": V.VecSimMetric_Cosine, "l2": V.VecSimMetric_L2, "ip": V.VecSimMetric_IP}[a.metric]
    idx = V.VecSimIndex(vt, a.dim, vm)
    idx.reserve(a.rows)
    assert idx.add_philox_rows(47, 0, a.rows, 1) == a.rows
    s = V.VecSimIndex(vt, a.dim, V.VecSimMetric_L2)
    s.add_philox_rows(47, 1 << 40, 64, 1)
    qs = s.read_rows(0, 64)
    s.free()
    bytes_pass = a.rows * a.dim * esz
    out = {"rows": a.rows, "dim": a.dim, "type": a.type, "metric": a.metric, "k": a.k}
    # single-query scan
    for i in range(5):
        idx.topk_query(qs[i], a.k)
    lib.RSGPU_ResetProfile()
    lib.RSGPU_SetProfiling(1)
    t0 = time.perf_counter()
    serial = [idx.topk_query(qs[i], a.k).results() for i in range(32)]
    wall = (time.perf_counter() - t0) / 32
    lib.RSGPU_SetProfiling(0)
    n, ms, _ = V.scan_profile()
    out["single"] = {"kernel_ms": ms / n, "wall_ms": wall * 1e3, "gbs": bytes_pass / (ms / n) / 1e6, "kernel": V.last_scan_kernel()}
    variants = [("default", None, None)]
    for ab in a.ab:
        key, vals = ab.split("=")
        variants = [("%s=%s" % (key, v), key, int(v)) for v in vals.split(",")]
    if a.variants:
        variants = [(v, v, None) for v in a.variants.split(";")]
    res = {}
    for rep_outer in range(2):
        for name, key, val in variants:
            if key and val is None:
                for kv in key.split(","):
                    k, x = kv.split("=")
                    assert lib.RSGPU_SetTuning(k.encode(), int(x)) == 0, kv
            elif key:
                assert lib.RSGPU_SetTuning(key.encode(), val) == 0
            for nq in (4, 8, 12, 16):
                idx.topk_batch(qs[:nq], a.k)
                V.coalesce_stats(reset=True)
                t0 = time.perf_counter()
                for r in range(a.reps):
                    ids, sc, cnt = idx.topk_batch(qs[r % 4 * nq % 32:][:nq], a.k)
                wall = (time.perf_counter() - t0) / a.reps
                st = V.coalesce_stats()
                ids, sc, cnt = idx.topk_batch(qs[:nq], a.k)
                same = all(ids[i].tolist() == serial[i][0].tolist() and sc[i].tolist() == serial[i][1].tolist() for i in range(nq))
                dev = st["mq_device_ns"] / max(st["mq_passes"], 1) / 1e6
                res.setdefault(name, {}).setdefault(str(nq), []).append(
                    {"scan_ms": round(dev, 4), "wall_ms": round(wall * 1e3, 4), "gbs": round(bytes_pass / dev / 1e6, 1),
                     "qps_wall": round(nq / wall, 1), "same": bool(same), "redo": st["mq_redo"]})
            res[name]["kernel"] = V.last_mq_scan_kernel()
    out["mq"] = res
    print(json.dumps(out, indent=1))
    idx.free()


if __name__ == "__main__":
    main()
