"""Writes oracle/_ref/gen/hybrid_reader_batched.c: the reference's src/iterators/hybrid_reader.c with its per-candidate
ad-hoc loop renamed (computeDistances_RAM -> computeDistances_RAM_perId) and THIS repository's batched body
(integration/hybrid_reader_batched.inc.c) included in its place -- what a maintainer applying the SURVEY.md 8(f)-2 shim
would end up with.  The output contains reference code: it lives under the git-ignored oracle/_ref/ and is never
committed.  Usage: python make_batched_hybrid_reader.py <reference root> <out file>"""
import os
import sys

ref, out = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(ref, "src", "iterators", "hybrid_reader.c")).read()
head = "static VecSimQueryReply_Code computeDistances_RAM(HybridIterator *hr) {"
anchor = "static VecSimQueryReply_Code computeDistances(HybridIterator *hr) {"
assert src.count(head) == 1 and src.count(anchor) == 1, "hybrid_reader.c no longer has the shape this shim targets"
src = src.replace(head, "static VecSimQueryReply_Code computeDistances_RAM_perId(HybridIterator *hr) {")
inc = os.path.join(os.path.dirname(here), "integration", "hybrid_reader_batched.inc.c")
src = src.replace(anchor, '#include "%s"\n\n%s' % (inc, anchor))
os.makedirs(os.path.dirname(out), exist_ok=True)
open(out, "w").write(src)
