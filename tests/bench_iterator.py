"""What the per-document vtable costs on top of the device intersection: BASELINE configs[4]'s two lists (FreqsOnly, df 5 M /
2.5 M of 50 M docs, ~250 k hits) through RSGPU_NewIntersectionIterator -- creation (decode cached + intersect + doc-id
mirror), then Read() to EOF building every `current` (the harness's constructors), then a SkipTo sweep as a hybrid batch
loop would issue.  (Uses the oracle's block writer as encoder and oracle/ext_harness.c as the module: lives under tests/.)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O  # noqa: E402
import oracle.ext as X  # noqa: E402
from redisearch_amd import search as S  # noqa: E402

n_docs = int(os.environ.get("N_DOCS", 50_000_000))
rng = np.random.default_rng(49)
lists = []
for r in (2, 4):
    docs = np.flatnonzero(rng.random(n_docs + 1) < 0.2 / r).astype(np.uint64)
    docs = docs[docs > 0]
    ii = O.InvertedIndex(O.C_FREQS_ONLY)
    ii.add_many(docs, np.minimum(1 + rng.geometric(0.5, docs.size), 255).astype(np.uint32))
    lists.append(ii)
g = [S.Postings.from_flat(l.flatten()) for l in lists]
S.load_iterators(X.handle())
S.intersect(g).free()          # decode once (cached), as a live index would have
out = {}
t0 = time.perf_counter()
it = S.new_iterator("and", g)
out["create_ms"] = (time.perf_counter() - t0) * 1e3
# lean consumer first (touches the aggregate and every child record once per hit, as a scorer loop does), three drains;
# then the verifying drain (copies every field of every record out: the harness's own work is part of its figure)
lean = []
for rep in range(3):
    t0 = time.perf_counter()
    n_lean, _ = X.iter_drain_lean(it)
    lean.append((time.perf_counter() - t0) * 1e3)
    X.iter_script(it, [(X.OP_REWIND, 0)])
out["drain_ms"] = min(lean)
out["drain_ms_all"] = lean
t0 = time.perf_counter()
d = X.iter_drain(it, 400_000, 2)
out["verifying_drain_ms"] = (time.perf_counter() - t0) * 1e3
out["hits"] = int(d["n"])
assert n_lean == out["hits"]
out["ns_per_read"] = out["drain_ms"] * 1e6 / max(out["hits"], 1)
out["ns_per_read_first_drain_incl_paging"] = lean[0] * 1e6 / max(out["hits"], 1)
ids = d["ids"]
targets = ids[:: max(len(ids) // 20000, 1)]
ops = [(X.OP_REWIND, 0)] + [(X.OP_SKIP, int(t) - 1) for t in targets]
t0 = time.perf_counter()
X.iter_script(it, ops)
out["skipto_sweep_ms"] = (time.perf_counter() - t0) * 1e3
out["skips"] = len(targets)
X.iter_free(it)
want = O.intersect(lists)[0]
out["ids_match_oracle"] = bool(np.array_equal(ids, want))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/iterator_bench.json", "w"), indent=1)
print(json.dumps(out))
