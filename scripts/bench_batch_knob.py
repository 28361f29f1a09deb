"""A/B of one RSGPU_SetTuning knob inside ONE process and one index (box-to-box variance is larger than most effects):
BASELINE configs[2], 10M x 768 fp16 IP top-100, 256 queries per pass; I8_SHADOW=1 for the int8-shadow path.
KNOB=name VALUES=0,1 ROUNDS=3 REPS=8.  The answers must not depend on the knob."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redisearch_amd import vecsim as V  # noqa: E402

rows = int(os.environ.get("ROWS", 10_000_000))
dim, k, batch = 768, 100, 256
reps, rounds = int(os.environ.get("REPS", 8)), int(os.environ.get("ROUNDS", 3))
knob = os.environ.get("KNOB", "qs_ovl").encode()
values = [int(v) for v in os.environ.get("VALUES", "0,1").split(",")]
lib = V.load()
i8 = os.environ.get("I8_SHADOW") == "1"
lib.RSGPU_SetTuning(b"shadow8", int(i8))
idx = V.VecSimIndex(V.VecSimType_FLOAT16, dim, V.VecSimMetric_IP)
lib.RSGPU_SetTuning(b"shadow8", 0)
idx.reserve(rows)
idx.add_philox_rows(47, 0, rows, 1)
qs = np.random.default_rng(48).uniform(-1, 1, (8, batch, dim)).astype(np.float16)
ref_ids, ref_sc, _ = idx.topk_batch(qs[0], k)
res = {v: [] for v in values}
for rd in range(rounds):
    for v in values:
        assert lib.RSGPU_SetTuning(knob, v) == 0
        ids, sc, _ = idx.topk_batch(qs[0], k)
        assert np.array_equal(ids, ref_ids) and np.array_equal(sc, ref_sc), v
        lib.RSGPU_ResetProfile()
        lib.RSGPU_SetProfiling(1)
        for i in range(reps):
            idx.topk_batch(qs[(i + 1) % 8], k)
        lib.RSGPU_SetProfiling(0)
        launches, ms, _ = V.scan_profile()
        res[v].append(ms / launches)
        print("round %d %s=%d: %.4f ms per pass" % (rd, knob.decode(), v, ms / launches), flush=True)
out = {"workload": "%dx%d fp16 IP top-%d, batch %d%s" % (rows, dim, k, batch, ", int8 shadow" if i8 else ""), "knob": knob.decode(),
       "device_ms_per_pass": {str(v): res[v] for v in values}, "best_ms": {str(v): min(res[v]) for v in values}}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/batch_knob_%s.json" % knob.decode(), "w"), indent=1)
print(json.dumps(out["best_ms"]))
