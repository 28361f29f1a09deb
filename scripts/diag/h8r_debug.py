"""which rows does the register-staged h8 pass miss?  (diagnostics)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from redisearch_amd import vecsim as V
lib = V.load()
dim, n, k = int(os.environ.get("DIM", 256)), int(os.environ.get("N", 700000)), 10
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(dim * 13 + k)
x = ((torch.rand((n, dim), device=dev, generator=gen) * 2 - 1)).to(torch.float16)
q = np.random.default_rng(dim + k).uniform(-1, 1, (300, dim)).astype(np.float16)
g = V.VecSimIndex(V.VecSimType_FLOAT16, dim, V.VecSimMetric_IP)
torch.cuda.synchronize(); g.add_device_rows(x.data_ptr(), n, 1)
res = {}
for shape in (2, 5, 8):
    lib.RSGPU_SetTuning(b"gemm_qs_h8", shape)
    res[shape] = g.topk_batch(q, k)
    for rep in range(3):
        again = g.topk_batch(q, k)
        if not np.array_equal(again[0], res[shape][0]):
            print("shape", shape, "not deterministic at rep", rep, int((again[0] != res[shape][0]).sum()))
ref = res[2][0]
for shape in (5, 8):
    ids = res[shape][0]
    bad = np.argwhere((ids != ref).any(axis=1)).ravel()
    print("shape", shape, "queries differing:", bad.tolist()[:20])
    for qi in bad[:6]:
        miss = sorted(set(ref[qi].tolist()) - set(ids[qi].tolist()))
        for lab in miss:
            row = lab - 1
            tile = row // 32
            print("   query %d (wave %d lane-col %d) misses row %d: row%%32=%d tile=%d wg=%d seq=%d of %d" % (
                qi, (qi % 256) // 32, qi % 32, row, row % 32, tile, tile % 256, tile // 256, (n // 32 + 255) // 256))
