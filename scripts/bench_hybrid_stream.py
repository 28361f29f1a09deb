"""configs[4] as the distinct-query stream of bench.py (16 term pairs, a query vector each), with the knobs of the two-launch
hybrid query A/B'd inside ONE process (box-to-box variance is larger than the effects): CONFIGS="name:key=v,...;name2:...".
Prints wall p50 / p95, the device times of both kernels and whether the answers equal the first configuration's."""
import json
import os
import sys
import time

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
n_docs, n_vec, dim, n_a, n_b = 50_000_000, 5_000_000, 768, 4, 4
rng = np.random.default_rng(149)
doc_len = (50 + rng.poisson(150, n_docs + 1)).astype(np.uint32)
doc_score = np.ones(n_docs + 1, np.float32)
avg = float(doc_len[1:].mean())
table = S.DocTable(doc_len, doc_score)
idx = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2)
idx.reserve(n_vec)
idx.add_philox_rows(B.SEED, 0, n_vec, 1)
raws = [B._term_list(rng, n_docs, n_docs * 0.2 / r) for r in [2] * n_a + [4] * n_b]
codec = os.environ.get("CODEC", "freqs_only")
enc = [B.encode_freqs_only(d, f) for d, f, _, _ in raws] if codec == "freqs_only" else [B.encode_full(d, f, m, o) for d, f, m, o in raws]
qvecs = B.philox_host_rows(V, B.QUERY_BASE + 100, n_a * n_b, dim)
out = {}
ref = None
# CONFIGS="name:key=v,key=v;name2:..."  (knobs not named keep their defaults; every configuration is timed in turn, inside this
# one process); MODES=warm | warm,cold; OUT=<file under gpurun_out/>
DEFAULTS = {"hybrid_dir": 1, "hybrid_packed_docs": 1, "hybrid_poll": 1, "hybrid_knn_pipeline": 1, "decode_lean": 1, "decode_dense": 1, "hybrid_select_split": 1}
spec = os.environ.get("CONFIGS", "defaults:")
CONFIGS = []
for part in spec.split(";"):
    name, _, kv = part.partition(":")
    knobs = dict(DEFAULTS)
    for item in kv.split(","):
        if item:
            k, v = item.split("=")
            knobs[k] = int(v)
    CONFIGS.append((name, knobs))
modes = tuple(os.environ.get("MODES", "warm").split(","))
for name, knobs in CONFIGS:
    for key, val in knobs.items():
        assert lib.RSGPU_SetTuning(key.encode(), val) == 0, key
    rec, ans, pairs = B._hybrid_stream(lib, S, enc, raws, table, idx, qvecs, n_docs, n_vec, avg, dim, n_a, cycles=int(os.environ.get("CYCLES", 4)),
                                       modes=modes, concurrent_threads=tuple(int(x) for x in os.environ.get("THREADS", "").split(",") if x))
    if ref is None:
        ref = ans
    same = all(a["top"][0].tolist() == b["top"][0].tolist() and a["top"][1].tolist() == b["top"][1].tolist() and
               a["knn"][0].tolist() == b["knn"][0].tolist() and a["knn"][1].tolist() == b["knn"][1].tolist() and a["n_hits"] == b["n_hits"]
               for a, b in zip(ans, ref))
    out[name] = {"knobs": {k: v for k, v in knobs.items() if v != DEFAULTS[k]}, "warm_p50": rec["warm"]["wall_ms_p50"], "warm_p95": rec["warm"]["wall_ms_p95"],
                 "warm_dev": rec["warm"]["device_ms"], "tile_hbm_frac": rec["warm"].get("tile_kernel_hbm_frac"),
                 "tile_plus_reduce_hbm_frac": rec["warm"].get("tile_plus_reduce_hbm_frac"), "same_answers_as_first_config": bool(same)}
    if "concurrent_callers" in rec["warm"]:
        out[name]["concurrent_callers"] = rec["warm"]["concurrent_callers"]
    if "cold" in rec:
        out[name].update({"cold_p50": rec["cold"]["wall_ms_p50"], "cold_dev": rec["cold"]["device_ms"],
                          "cold_decode_gbs": rec["cold"].get("decode_gbs_of_encoded_bytes")})
    print(name, json.dumps(out[name]), flush=True)
for key, val in DEFAULTS.items():
    lib.RSGPU_SetTuning(key.encode(), val)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(os.path.join("gpurun_out", os.environ.get("OUT", "hybrid_stream_ab_%s.json" % codec)), "w"), indent=1)
