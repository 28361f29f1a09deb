"""GPU: two-level query trees (RSGPU_EvalTree) -- Intersection of Unions / Union of Intersections / mixed -- against the
CPU oracle: doc ids and per-term frequencies from set algebra over the oracle's decoded lists, scores from the oracle's
result-tree scorers (O.Node) built per document in the aggregate's child order, slop and max_slop / in_order from the
oracle's proximity restatement with merged union positions."""
import numpy as np
import pytest

import oracle as O
from redisearch_amd import search as S

pytestmark = pytest.mark.gpu
SCORERS = ["BM25STD", "BM25STD.TANH", "BM25", "TFIDF", "TFIDF.DOCNORM", "DISMAX", "DOCSCORE"]


def rand_list(rng, codec, n_docs, max_doc, with_offsets):
    docs = np.unique(rng.integers(1, max_doc, n_docs))
    ii = O.InvertedIndex(codec)
    rec = {}
    for d in docs.tolist():
        f = int(rng.integers(1, 30))
        pos = sorted(set(int(x) for x in rng.integers(1, 50, int(rng.integers(1, 5))))) if with_offsets else []
        offs, last = b"", 0
        for p in pos:
            offs += O.varint_encode(p - last)
            last = p
        ii.add(d, f, 1, offs)
        rec[d] = (f, pos, offs)
    return ii, rec


class OracleTree:
    """root_op over groups [(op, weight, [list index...])]"""

    def __init__(self, root_op, groups, recs, sizes, max_slop=None, in_order=False):
        self.root_op, self.recs = root_op, recs
        gs = []
        for op, w, idx in groups:
            if op == S.OP_INTERSECT:
                idx = sorted(idx, key=lambda i: sizes[i])          # stable: by estimated size
                est = sizes[idx[0]]
                docs = set.intersection(*[set(recs[i]) for i in idx])
            elif op == S.OP_UNION:
                est = sum(sizes[i] for i in idx)
                docs = set.union(*[set(recs[i]) for i in idx])
            else:
                est, docs = sizes[idx[0]], set(recs[idx[0]])
            gs.append(dict(op=op, w=w, idx=list(idx), est=est, docs=docs))
        if root_op == S.OP_INTERSECT and not in_order:
            gs = sorted(gs, key=lambda g: g["est"])
        self.groups = gs
        docs = set.intersection(*[g["docs"] for g in gs]) if root_op == S.OP_INTERSECT else set.union(*[g["docs"] for g in gs])
        self.docs = sorted(docs)
        if root_op == S.OP_INTERSECT and (max_slop is not None or in_order):
            self.docs = [d for d in self.docs if O.within_range(self.children(d), max_slop, in_order)]
        self.leaf_order = [i for g in gs for i in g["idx"]]

    def children(self, d):
        out = []
        for g in self.groups:
            if g["op"] == S.OP_TERM:
                out.append((False, [self.recs[g["idx"][0]][d][2]]))
            else:
                out.append((True, [self.recs[i][d][2] if d in self.recs[i] else b"" for i in g["idx"]]))
        return out

    def node(self, d, idf, bidf, w):
        kids = []
        for g in self.groups:
            if d not in g["docs"]:
                continue                                              # a union root only holds the matched children
            if g["op"] == S.OP_TERM:
                i = g["idx"][0]
                f, pos, _ = self.recs[i][d]
                kids.append(O.term(f, idf[i], bidf[i], w[i], offsets=pos))
            else:
                sub = []
                for i in g["idx"]:
                    if d in self.recs[i]:
                        f, pos, _ = self.recs[i][d]
                        sub.append(O.term(f, idf[i], bidf[i], w[i], offsets=pos))
                kids.append((O.union if g["op"] == S.OP_UNION else O.intersection)(sub, g["w"]))
        return (O.intersection if self.root_op == S.OP_INTERSECT else O.union)(kids)


def run_case(rng, root_op, shape, with_offsets, max_slop=None, in_order=False):
    codec = O.C_FULL if with_offsets else O.C_FREQS_ONLY
    n_lists = sum(len(g[2]) for g in shape)
    built = [rand_list(rng, codec, int(rng.integers(300, 1500)), 2500, with_offsets) for _ in range(n_lists)]
    lists, recs = [b[0] for b in built], [b[1] for b in built]
    sizes = [l.unique_docs for l in lists]
    g = [S.Postings.from_flat(l.flatten()) for l in lists]
    groups = [(op, w, [g[i] for i in idx]) for op, w, idx in shape]
    h = S.TreeHits(root_op, groups, max_slop=max_slop, in_order=in_order)
    ot = OracleTree(root_op, shape, recs, sizes, max_slop, in_order)
    gi, gf = h.read()
    assert gi.tolist() == ot.docs, (root_op, shape)
    # a term's frequency is part of the result only where its GROUP matched the document (an intersection group that
    # misses one of its terms contributes nothing, even if this term is present)
    group_docs = {}
    for gr in ot.groups:
        for i in gr["idx"]:
            group_docs[i] = gr["docs"]
    for li in range(n_lists):
        assert gf[li].tolist() == [recs[li][d][0] if (d in recs[li] and d in group_docs[li]) else 0 for d in ot.docs]
    if not ot.docs:
        return 0
    n_docs = 2500
    doc_len = rng.integers(5, 200, n_docs + 1).astype(np.uint32)
    doc_score = rng.choice([1.0, 0.5], n_docs + 1).astype(np.float32)
    max_freq = rng.integers(1, 40, n_docs + 1).astype(np.uint32)
    table = S.DocTable(doc_len, doc_score, max_freq)
    idf = [S.calculate_idf(n_docs, s) for s in sizes]
    bidf = [S.calculate_idf_bm25(n_docs, s) for s in sizes]
    w = [float(x) for x in rng.choice([1.0, 0.5, 2.0], n_lists)]
    avg = float(doc_len[1:].mean())
    sample = rng.choice(len(ot.docs), min(60, len(ot.docs)), replace=False)
    for scorer in SCORERS:
        gs = h.score(table, scorer, idf, bidf, w, n_docs, avg, root_weight=1.5)
        for j in sample:
            d = ot.docs[j]
            node = ot.node(d, idf, bidf, w)
            node.c.weight = 1.5
            want = O.score(scorer, node, float(doc_score[d]), int(max_freq[d]), int(doc_len[d]), n_docs, avg)
            if scorer == "BM25STD.TANH":
                assert gs[j] == pytest.approx(want, rel=1e-12)
            else:
                assert gs[j] == want, (scorer, d, gs[j], want)
    return len(ot.docs)


T, U, I = S.OP_TERM, S.OP_UNION, S.OP_INTERSECT


@pytest.mark.parametrize("with_offsets", [False, True])
@pytest.mark.parametrize("name,root,shape", [
    ("and_of_ors", I, [(U, 1.0, [0, 1, 2]), (U, 0.5, [3, 4])]),                     # (a|a'|a'') (b|b')
    ("or_of_ands", U, [(I, 1.0, [0, 1]), (I, 2.0, [2, 3])]),                        # (a b) | (c d)
    ("term_and_or", I, [(T, 1.0, [0]), (U, 1.0, [1, 2, 3]), (T, 1.0, [4])]),        # a (b|c|d) e
    ("or_of_term_and_and", U, [(T, 1.0, [0]), (I, 0.7, [1, 2, 3])]),                # a | (b c d)
    ("and_of_ands", I, [(I, 1.0, [0, 1]), (I, 3.0, [2, 3])]),                       # (a b) (c d)
    ("single_union_group", I, [(U, 1.0, [0, 1])]),
])
def test_tree_matches_oracle(name, root, shape, with_offsets):
    rng = np.random.default_rng(abs(hash(name)) % 10000 + int(with_offsets))
    assert run_case(rng, root, shape, with_offsets) > 0 or name == "and_of_ands"


@pytest.mark.parametrize("max_slop,in_order", [(0, False), (2, False), (None, True), (1, True), (10, True)])
def test_tree_with_slop_and_order_merges_union_positions(max_slop, in_order):
    """a (b|b'|b'') c with max_slop / in_order: the union child's positions are the merge of its matched terms'
    (proximity.rs OffsetIter::Merge)."""
    rng = np.random.default_rng(77 + (max_slop or 0) + int(in_order))
    run_case(rng, I, [(T, 1.0, [0]), (U, 1.0, [1, 2, 3]), (T, 1.0, [4])], True, max_slop, in_order)
