"""CPU: the order an intersection iterates its children in (oracle_intersection_sort_key / _child_order, oracle/postings_oracle.c)
against the reference's own unit tests for it, and the test oracles of the GPU tree tests (tests/test_gpu_tree.py OracleTree /
DeepOracle, which the device evaluation is held to) against that restatement."""
import numpy as np

import oracle as O
from redisearch_amd import search as S
from tests.test_gpu_tree import DeepOracle, OracleTree


def test_reference_kats_for_the_sort_weight():
    # rqe_iterators/tests/integration/intersection.rs:1412-1440 sort_weight_nested_intersection_sorts_first: a plain child of 10
    # documents passed first, a nested intersection of five 10-document children: keys 10 * 1.0 against 10 * (1 / 5) = 2.0
    assert O.intersection_sort_key(10, O.K_INTERSECTION, 5) == 2.0 and O.intersection_sort_key(10) == 10.0
    assert O.intersection_child_order([(10, O.K_TERM, 1), (10, O.K_INTERSECTION, 5)]) == [1, 0]
    # intersection.rs:1056-1075 children_sorted_by_estimated: large (1000), small (1), medium (7) -> small, medium, large
    assert O.intersection_child_order([(1000, 0, 1), (1, 0, 1), (7, 0, 1)]) == [1, 2, 0]
    # intersection.rs:1036-1049: in_order keeps the query's order
    assert O.intersection_child_order([(5, 0, 1), (3, 0, 1), (4, 0, 1)], in_order=True) == [0, 1, 2]
    # union_common.rs:1968-1995: a union weighs 1 without prioritizeIntersectUnionChildren, its children with it (one child: 1)
    assert O.intersection_sort_key(6, O.K_UNION, 2) == 6.0
    assert O.intersection_sort_key(6, O.K_UNION, 2, True) == 12.0 and O.intersection_sort_key(3, O.K_UNION, 1, True) == 3.0
    # equal keys keep the query's order (a stable sort: sort_by)
    assert O.intersection_child_order([(8, 0, 1), (16, O.K_INTERSECTION, 2), (8, 0, 1)]) == [0, 1, 2]


def _recs(rng, sizes):
    return [{int(d): (1, [], b"") for d in rng.choice(np.arange(1, 5000), s, replace=False)} for s in sizes]


def test_the_two_level_test_oracle_orders_children_like_the_restatement():
    rng = np.random.default_rng(5)
    T, U, I = S.OP_TERM, S.OP_UNION, S.OP_INTERSECT
    for _ in range(200):
        n_groups = int(rng.integers(2, 6))
        shape, sizes = [], []
        for _g in range(n_groups):
            op = int(rng.choice([T, U, I]))
            k = 1 if op == T else int(rng.integers(1, 4))
            shape.append((op, 1.0, list(range(len(sizes), len(sizes) + k))))
            sizes += [int(x) for x in rng.integers(1, 40, k) * 25]       # (many equal sizes: ties)
        ot = OracleTree(I, shape, _recs(rng, sizes), sizes)
        kids = []
        for op, _w, idx in shape:
            if op == I:
                kids.append((min(sizes[i] for i in idx), O.K_INTERSECTION, len(idx)))
            elif op == U:
                kids.append((sum(sizes[i] for i in idx), O.K_UNION, len(idx)))
            else:
                kids.append((sizes[idx[0]], O.K_TERM, 1))
        want = O.intersection_child_order(kids)
        got = [next(j for j, (op, _w, idx) in enumerate(shape) if sorted(idx) == sorted(g["idx"])) for g in ot.groups]
        assert got == want, (shape, sizes)


def test_the_deep_test_oracle_orders_children_like_the_restatement():
    rng = np.random.default_rng(6)
    sizes = [int(x) for x in rng.integers(1, 30, 9) * 40]
    recs = _recs(rng, sizes)
    # (a b c) d (e | f) ((g h) i): the root's children are an intersection of three, a term, a union, an intersection of two
    tree = ("and", 1.0, [("and", 1.0, [("t", 0), ("t", 1), ("t", 2)]), ("t", 3), ("or", 1.0, [("t", 4), ("t", 5)]),
                         ("and", 1.0, [("and", 1.0, [("t", 6), ("t", 7)]), ("t", 8)])])
    do = DeepOracle(tree, recs, sizes)
    est = lambda ix: min(sizes[i] for i in ix)
    kids = [(est([0, 1, 2]), O.K_INTERSECTION, 3), (sizes[3], O.K_TERM, 1), (sizes[4] + sizes[5], O.K_UNION, 2),
            (min(est([6, 7]), sizes[8]), O.K_INTERSECTION, 2)]
    want = O.intersection_child_order(kids)
    firsts = [0, 3, 4, 6]                     # a leaf that identifies each root child
    got = []
    for k in do.tree["kids"]:
        leaves = do._leaves(k)
        got.append(next(j for j, f in enumerate(firsts) if (f in leaves) or (j == 3 and 8 in leaves)))
    assert got == want
