"""GPU check for scripts/diag/bf16_int8_shadow.patch (run by bf16_experiment.sh after the patch is applied and built):
BFLOAT16 IP / cosine indexes with the int8 shadow -- the batched path must give ids and distances BIT-IDENTICAL to one
VecSimIndex_TopKQuery per query on an index without the shadow, must really take the int8 passes (bytes accounted = one per
element), and must survive appends / deletes.  Mirrors tests/test_gpu_batch_i8_shadow.py's FLOAT16 cases."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from redisearch_amd import vecsim as V  # noqa: E402

BF16, IP, COS = V.VecSimType_BFLOAT16, V.VecSimMetric_IP, V.VecSimMetric_Cosine
lib = V.load()


def build(x, dim, metric):
    g = V.VecSimIndex(BF16, dim, metric)
    torch.cuda.synchronize()
    g.add_device_rows(x.data_ptr(), x.shape[0], 1)
    return g


def knobs(on):
    assert lib.RSGPU_SetTuning(b"shadow8", int(on)) == 0
    assert lib.RSGPU_SetTuning(b"shadow8_bf16", int(on)) == 0


bad = 0
for metric in (IP, COS):
    for dim, n in ((768, 530_001), (256, 700_000)):
        for k in (10, 100):
            dev = torch.device("cuda", 0)
            gen = torch.Generator(device=dev)
            gen.manual_seed(dim * 7 + k)
            x = (torch.rand((n, dim), device=dev, generator=gen) * 2 - 1).to(torch.bfloat16)
            queries = np.random.default_rng(dim + k).uniform(-1, 1, (300, dim)).astype(np.float32)
            knobs(False)
            p = build(x.clone(), dim, metric)
            want = [p.topk_query(q, k).results() for q in queries]
            p.free()
            knobs(True)
            g = build(x, dim, metric)
            knobs(False)
            lib.RSGPU_ResetProfile()
            lib.RSGPU_SetProfiling(1)
            ids, sc, cnt = g.topk_batch(queries, k)
            lib.RSGPU_SetProfiling(0)
            launches, _, by = V.scan_profile()
            ok = launches == 2 and by == 2 * n * dim
            for i, (wi, ws) in enumerate(want):
                ok = ok and cnt[i] == len(wi) and ids[i][: cnt[i]].tolist() == wi.tolist() and sc[i][: cnt[i]].tolist() == ws.tolist()
            # appends (one aligned with query 0) and deletes, then again
            extra = np.random.default_rng(3).uniform(-1, 1, (20, dim)).astype(np.float32)
            extra[4] = queries[0]
            for i in range(20):
                g.add_vector(extra[i], n + 1 + i)
            for lbl in (7, n + 3, 1234):
                g.delete_vector(lbl)
            single = [g.topk_query(q, k).results() for q in queries[:12]]
            ids, sc, cnt = g.topk_batch(queries[:12], k)
            for i, (wi, ws) in enumerate(single):
                ok = ok and ids[i][: cnt[i]].tolist() == wi.tolist() and sc[i][: cnt[i]].tolist() == ws.tolist()
            print("metric %d dim %d n %d k %d: %s (launches %d, bytes/elem %.2f)" % (metric, dim, n, k, "ok" if ok else "MISMATCH",
                                                                                   launches, by / (launches * n * dim) if launches else 0))
            bad += not ok
            g.free()
            del x
            torch.cuda.empty_cache()
print("bf16 int8 shadow: %d failing case(s)" % bad)
sys.exit(1 if bad else 0)
