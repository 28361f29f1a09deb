"""Timing experiment for the int8 query-stationary pass: a FLOAT16 index of dim 384 has 768-byte rows, the bytes an int8
shadow of a dim-768 index would have; RSGPU_SetTuning("qs_force_i8", 1) runs the int8 MFMA kernel over them (results are
meaningless, candidates are suppressed).  Prints device ms per 256-query pass for fp16 (dim 384) and forced int8 (768 int8)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redisearch_amd import vecsim as V  # noqa: E402

rows, dim, k, batch = int(os.environ.get("ROWS", 10_000_000)), 384, 100, 256
lib = V.load()
dev = torch.device("cuda", 0)
idx = V.VecSimIndex(V.VecSimType_FLOAT16, dim, V.VecSimMetric_IP)
idx.reserve(rows)
gen = torch.Generator(device=dev)
gen.manual_seed(47)
done = 0
while done < rows:
    m = min(1_000_000, rows - done)
    t = (torch.rand((m, dim), device=dev, generator=gen) * 2 - 1).to(torch.float16)
    torch.cuda.synchronize()
    idx.add_device_rows(t.data_ptr(), m, done + 1)
    done += m
    del t
qs = np.random.default_rng(48).uniform(-1, 1, (4, batch, dim)).astype(np.float16)
idx.topk_batch(qs[0], k)
out = {}
for rd in range(2):
    for force, gq in ((0, 1), (1, 1), (1, 2), (0, 2)):
        lib.RSGPU_SetTuning(b"qs_force_i8", force)
        lib.RSGPU_SetTuning(b"gemm_qs", gq)
        idx.topk_batch(qs[0], k)
        lib.RSGPU_ResetProfile()
        lib.RSGPU_SetProfiling(1)
        for i in range(6):
            idx.topk_batch(qs[i % 4], k)
        lib.RSGPU_SetProfiling(0)
        launches, ms, _ = V.scan_profile()
        out.setdefault("force_i8=%d gemm_qs=%d" % (force, gq), []).append(ms / launches)
        print(rd, force, gq, ms / launches, flush=True)
lib.RSGPU_SetTuning(b"qs_force_i8", 0)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/qs_i8_feasibility.json", "w"), indent=1)
print(json.dumps(out))
