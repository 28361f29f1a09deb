"""configs[4]'s setup of bench.py once, then (a) the distinct-query stream on the shipped configuration (the two-launch tile
kernel after round 4's refactor of its KNN phase into a shared device function) and (b) the shapes the general tile kernel
took over from the staged pipeline, general vs staged in the same process (bench.py _hybrid_general_shapes)."""
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
enc_fo = [B.encode_freqs_only(d, f) for d, f, _, _ in raws]
enc_full = [B.encode_full(d, f, m, o) for d, f, m, o in raws]
qvecs = B.philox_host_rows(V, B.QUERY_BASE + 100, n_a * n_b, dim)
out = {}
if os.environ.get("SKIP_STREAM") != "1":
    rec, ans, pairs = B._hybrid_stream(lib, S, enc_fo, raws, table, idx, qvecs, n_docs, n_vec, avg, dim, n_a, cycles=int(os.environ.get("CYCLES", 3)))
    out["stream_freqs_only"] = {"warm_p50": rec["warm"]["wall_ms_p50"], "warm_dev": rec["warm"]["device_ms"],
                                "tile_hbm_frac": rec["warm"].get("tile_kernel_hbm_frac"), "cold_p50": rec["cold"]["wall_ms_p50"],
                                "cold_dev": rec["cold"]["device_ms"]}
    print("stream", json.dumps(out["stream_freqs_only"]), flush=True)
    lib.RSGPU_SetTuning(b"hybrid_force_general", 1)   # the same stream through the general kernel: what its generality costs
    rec, ans2, _ = B._hybrid_stream(lib, S, enc_fo, raws, table, idx, qvecs, n_docs, n_vec, avg, dim, n_a, cycles=int(os.environ.get("CYCLES", 3)))
    lib.RSGPU_SetTuning(b"hybrid_force_general", 0)
    out["stream_freqs_only_general_kernel_forced"] = {"path": rec["warm"]["path"], "warm_p50": rec["warm"]["wall_ms_p50"], "warm_dev": rec["warm"]["device_ms"],
                                                      "tile_hbm_frac": rec["warm"].get("tile_kernel_hbm_frac"),
                                                      "same_answers": all(x["top"][0].tolist() == y["top"][0].tolist() and x["top"][1].tolist() == y["top"][1].tolist()
                                                                          and x["knn"][0].tolist() == y["knn"][0].tolist() and x["n_hits"] == y["n_hits"]
                                                                          for x, y in zip(ans, ans2))}
    print("stream, general kernel forced", json.dumps(out["stream_freqs_only_general_kernel_forced"]), flush=True)
g = B._hybrid_general_shapes(lib, S, enc_fo, enc_full, raws, table, idx, qvecs, n_docs, avg, n_a)
out["general_tile_kernel_shapes"] = g
for k, v in g.items():
    print(k, json.dumps(v), flush=True)
# RSGPU_EvalTree on its own -- what creating the iterator seam's tree iterator costs: a (b|c), the hit list built by the tile kernel
# (tile + hit count + pack) against the staged evaluation (the union's own hit list first, then the intersection)
import time  # noqa: E402
fo = [S.Postings.from_flat(e) for e in enc_fo[:1] + enc_fo[n_a:n_a + 2]]
groups = [(S.OP_TERM, 1.0, [fo[0]]), (S.OP_UNION, 1.0, [fo[1], fo[2]])]
ev = {}
for mode, knob in (("tile_kernel", 1), ("staged", 0)):
    lib.RSGPU_SetTuning(b"hybrid_tree_tiles", knob)
    S.TreeHits(S.OP_INTERSECT, groups).free()
    t = []
    for _ in range(12):
        t0 = time.perf_counter()
        h = S.TreeHits(S.OP_INTERSECT, groups)
        t.append((time.perf_counter() - t0) * 1e3)
        n = len(h)
        h.free()
    ev[mode] = {"ms_p50": float(np.percentile(t, 50)), "hits": n, "path": S.hybrid_path()}
lib.RSGPU_SetTuning(b"hybrid_tree_tiles", 1)
out["eval_tree_term_and_union_of_two"] = ev
print("eval_tree", json.dumps(ev), flush=True)
# RSGPU_IntersectEx with a window alone -- `"a b"` as a phrase (slop 0, in order) over two Full-codec lists: the tile kernel's hit
# list against the staged probe -> prox_filter -> scan -> write
fu = [S.Postings.from_flat(e) for e in (enc_full[0], enc_full[n_a])]
ph = {}
for mode, knob in (("tile_kernel", 1), ("staged", 0)):
    lib.RSGPU_SetTuning(b"hybrid_tree_tiles", knob)
    S.intersect(fu, max_slop=0, in_order=True).free()
    t = []
    for _ in range(12):
        t0 = time.perf_counter()
        h = S.intersect(fu, max_slop=0, in_order=True)
        t.append((time.perf_counter() - t0) * 1e3)
        n = len(h)
        h.free()
    ph[mode] = {"ms_p50": float(np.percentile(t, 50)), "hits": n, "path": S.hybrid_path()}
lib.RSGPU_SetTuning(b"hybrid_tree_tiles", 1)
out["phrase_intersection_two_full_codec_terms"] = ph
print("phrase", json.dumps(ph), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/hybrid_general_shapes.json", "w"), indent=1)
