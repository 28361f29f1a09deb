"""Encodes the posting lists scripts/bench_decode.py times (the oracle's block writer is the encoder -- test
infrastructure, hence under tests/): python tests/make_decode_lists.py out.npz"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O  # noqa: E402

rng = np.random.default_rng(49)
n_docs = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
out = {}
for r in (2, 4):
    docs = np.flatnonzero(rng.random(n_docs + 1) < 0.2 / r).astype(np.uint64)
    docs = docs[docs > 0]
    freqs = np.minimum(1 + rng.geometric(0.5, docs.size), 255).astype(np.uint32)
    for name, codec in (("freqs_only", O.C_FREQS_ONLY), ("docids_only", O.C_DOCIDS_ONLY)):
        ii = O.InvertedIndex(codec)
        ii.add_many(docs, freqs)
        out["%s_r%d" % (name, r)] = ii.flatten()
np.savez(sys.argv[1], **out)
