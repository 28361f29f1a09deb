"""GPU: two-level query trees (RSGPU_EvalTree) -- Intersection of Unions / Union of Intersections / mixed -- against the
CPU oracle: doc ids and per-term frequencies from set algebra over the oracle's decoded lists, scores from the oracle's
result-tree scorers (O.Node) built per document in the aggregate's child order, slop and max_slop / in_order from the
oracle's proximity restatement with merged union positions."""
import zlib

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
            # the sort key of an intersection's children: estimate x intersection_sort_weight -- 1 / children for a child
            # intersection (intersection.rs:94-119,580-582), 1 for terms and (prioritizeIntersectUnionChildren off) unions
            gs = sorted(gs, key=lambda g: g["est"] * (1.0 / len(g["idx"]) if g["op"] == S.OP_INTERSECT else 1.0))
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
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 10000 + int(with_offsets))
    assert run_case(rng, root, shape, with_offsets) > 0 or name == "and_of_ands"


@pytest.mark.parametrize("max_slop,in_order", [(0, False), (2, False), (None, True), (1, True), (10, True)])
def test_tree_with_slop_and_order_merges_union_positions(max_slop, in_order):
    """a (b|b'|b'') c with max_slop / in_order: the union child's positions are the merge of its matched terms'
    (proximity.rs OffsetIter::Merge)."""
    rng = np.random.default_rng(77 + (max_slop or 0) + int(in_order))
    run_case(rng, I, [(T, 1.0, [0]), (U, 1.0, [1, 2, 3]), (T, 1.0, [4])], True, max_slop, in_order)


# ---- trees of any depth (RSGPU_EvalTreeNodes) ----------------------------------------------------------------------------
class DeepOracle:
    """Set algebra + result trees for nested tuples ("t", i) / ("and" | "or", weight, [children], max_slop, in_order): the
    reference's iterators, restated -- an intersection iterates its children by ascending estimate x sort weight (stable) unless
    in_order (intersection.rs:94-119; estimate = the smallest child's), a union keeps the query order (estimate = the
    sum) and its result holds the matched children only (union_flat.rs:297-320)."""

    def __init__(self, tree, recs, sizes):
        self.recs, self.sizes = recs, sizes
        self.tree = self._prep(tree)
        self.docs = sorted(self._docs(self.tree))
        self.leaf_order = self._leaves(self.tree)

    def _prep(self, t):
        if t[0] == "t":
            return dict(op="t", i=t[1], est=self.sizes[t[1]], docs=set(self.recs[t[1]]))
        kids = [self._prep(ch) for ch in t[2]]
        in_order = len(t) > 4 and bool(t[4])
        max_slop = t[3] if len(t) > 3 else None
        if t[0] == "and":
            if not in_order:
                # estimate x intersection_sort_weight (intersection.rs:94-119): 1 / children for a child intersection
                kids = sorted(kids, key=lambda k: k["est"] * (1.0 / len(k["kids"]) if k["op"] == "and" else 1.0))
            docs = set.intersection(*[k["docs"] for k in kids])
            if max_slop is not None or in_order:
                docs = {d for d in docs if O.within_range([self._offsets(k, d) for k in kids], max_slop, in_order)}
            return dict(op="and", w=t[1], kids=kids, est=min(k["est"] for k in kids), docs=docs)
        return dict(op="or", w=t[1], kids=kids, est=sum(k["est"] for k in kids), docs=set.union(*[k["docs"] for k in kids]))

    def _docs(self, n):
        return n["docs"]

    def _leaves(self, n):
        return [n["i"]] if n["op"] == "t" else [i for k in n["kids"] for i in self._leaves(k)]

    def _present_leaves(self, n, d):
        """term indices under n that are part of document d's result"""
        if d not in n["docs"]:
            return []
        if n["op"] == "t":
            return [n["i"]]
        return [i for k in n["kids"] for i in self._present_leaves(k, d)]

    def _offsets(self, n, d):
        if n["op"] == "t":
            return (False, [self.recs[n["i"]][d][2]])
        # an aggregate child presents the merged positions of the leaves that matched under it
        pres = set(self._present_leaves(n, d))
        return (True, [self.recs[i][d][2] if i in pres else b"" for i in self._leaves(n)])

    def node(self, n, d, idf, bidf, w):
        if n["op"] == "t":
            f, pos, _ = self.recs[n["i"]][d]
            return O.term(f, idf[n["i"]], bidf[n["i"]], w[n["i"]], offsets=pos)
        kids = [self.node(k, d, idf, bidf, w) for k in n["kids"] if d in k["docs"]]
        return (O.intersection if n["op"] == "and" else O.union)(kids, n["w"])

    def freq(self, li, d):
        return self.recs[li][d][0] if li in self._present_leaves(self.tree, d) else 0


def random_tree(rng, depth, next_leaf, max_leaves):
    """a random tree of exactly `depth` levels below the root along at least one path"""
    def grow(d, force):
        if d == 0 or (not force and rng.random() < 0.35) or next_leaf[0] >= max_leaves - 1:
            i = next_leaf[0]
            next_leaf[0] += 1
            return ("t", i)
        op = "and" if rng.random() < 0.5 else "or"
        n_kids = int(rng.integers(1, 4))
        kids = [grow(d - 1, force and k == 0) for k in range(n_kids)]
        return (op, float(rng.choice([1.0, 0.5, 2.0, 1.25])), kids)
    return grow(depth, True)


def tree_depth(t):
    return 0 if t[0] == "t" else 1 + max(tree_depth(ch) for ch in t[2])


def run_deep(rng, tree, n_lists, with_offsets):
    codec = O.C_FULL if with_offsets else O.C_FREQS_ONLY
    built = [rand_list(rng, codec, int(rng.integers(900, 1900)), 2500, with_offsets) for _ in range(n_lists)]
    lists, recs = [b[0] for b in built], [b[1] for b in built]
    sizes = [l.unique_docs for l in lists]
    g = [S.Postings.from_flat(l.flatten()) for l in lists]
    h = S.NodeHits(tree, g)
    ot = DeepOracle(tree, recs, sizes)
    gi, gf = h.read()
    assert gi.tolist() == ot.docs
    leaf_order = h.leaf_order() if hasattr(h, "leaf_order") else None
    for li in range(n_lists):
        assert gf[li].tolist() == [ot.freq(li, d) for d in ot.docs], li
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
    sample = rng.choice(len(ot.docs), min(80, len(ot.docs)), replace=False)
    for scorer in SCORERS:
        gs = h.score(table, scorer, idf, bidf, w, n_docs, avg, root_weight=1.5)
        for j in sample:
            d = ot.docs[j]
            node = ot.node(ot.tree, d, idf, bidf, w)
            node.c.weight = 1.5
            want = O.score(scorer, node, float(doc_score[d]), int(max_freq[d]), int(doc_len[d]), n_docs, avg)
            if scorer == "BM25STD.TANH":
                assert gs[j] == pytest.approx(want, rel=1e-12)
            else:
                assert gs[j] == want, (scorer, d, gs[j], want, tree)
    return len(ot.docs)


@pytest.mark.parametrize("with_offsets", [False, True])
@pytest.mark.parametrize("seed", range(10))
@pytest.mark.parametrize("depth", [3, 4, 6])
def test_random_deep_trees_match_the_oracle(depth, seed, with_offsets):
    """Depth 3 / 4 / 6 random trees: doc ids, per-term frequencies and every scorer against the oracle's result-tree
    scorers (themselves held to the reference's compiled default.c, tests/test_scorer_plugin.py), bit for bit."""
    rng = np.random.default_rng(1000 * depth + 10 * seed + int(with_offsets))
    nl = [0]
    tree = random_tree(rng, depth, nl, 14)
    if tree[0] == "t":
        tree = ("and", 1.0, [tree])
    assert tree_depth(tree) >= min(depth, 2)
    run_deep(rng, tree, nl[0], with_offsets)


def test_fixed_deep_trees():
    """hand-written shapes: ((a b) | (c (d | e))) f ;  a | (b (c | (d e)))  -- and the same tree through the two-level entry
    point gives the same hits and scores where both apply"""
    rng = np.random.default_rng(5)
    t1 = ("and", 1.0, [("or", 0.5, [("and", 2.0, [("t", 0), ("t", 1)]), ("and", 1.0, [("t", 2), ("or", 1.5, [("t", 3), ("t", 4)])])]),
                       ("t", 5)])
    assert run_deep(rng, t1, 6, True) >= 0
    t2 = ("or", 1.0, [("t", 0), ("and", 0.7, [("t", 1), ("or", 1.0, [("t", 2), ("and", 3.0, [("t", 3), ("t", 4)])])])])
    assert run_deep(rng, t2, 5, False) > 0


def test_nested_intersection_with_its_own_slop():
    """(a b)~slop inside a union inside an intersection: the nested node's max_slop / in_order filter its own hits"""
    rng = np.random.default_rng(9)
    t = ("and", 1.0, [("or", 1.0, [("and", 1.0, [("t", 0), ("t", 1)], 3, True), ("t", 2)]), ("t", 3)])
    run_deep(rng, t, 4, True)
