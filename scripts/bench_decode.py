"""Cold decode of a posting list on the device (RSGPU_Postings_Decode, HIP-event time of the decode stage): BASELINE configs[4]'s
two lists (df 5 M and 2.5 M of 50 M docs) in the FreqsOnly and Full codecs.  Prints entries/s and GB/s of encoded bytes."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redisearch_amd import search as S  # noqa: E402
from redisearch_amd import vecsim as V  # noqa: E402

fl = np.load(sys.argv[1], allow_pickle=True)
V.load().RSGPU_SetTuning(b"cache_decoded", int(os.environ.get("CACHE_DECODED", "0")))   # decode-per-query mode
if os.environ.get("DECODE_SYNC"):
    V.load().RSGPU_SetTuning(b"decode_sync", int(os.environ["DECODE_SYNC"]))
out = []
for key in fl.files:
    d = fl[key].item()
    p = S.Postings.from_flat(d)
    lib = S.load()
    best = None
    first = None
    for _ in range(5):
        lib.RSGPU_SetProfiling(1)
        p.decode()
        prof = S.profile()
        lib.RSGPU_SetProfiling(0)
        ms = prof.get("decode_ms")
        first = ms if first is None else first
        best = ms if best is None else min(best, ms)
    r = {"list": key, "entries": int(p.num_entries), "bytes": int(p.num_bytes), "decode_ms": best, "first_decode_ms": first,
         "entries_per_s": p.num_entries / best * 1e3, "encoded_gbs": p.num_bytes / best / 1e6}
    out.append(r)
    print(json.dumps(r), flush=True)
    p.free()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/decode_bench.json", "w"), indent=1)
