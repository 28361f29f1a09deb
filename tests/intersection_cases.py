"""The fixtures of the reference's intersection iterator tests
(src/redisearch_rs/rqe_iterators/tests/integration/intersection.rs:30-80, union_common.rs): 2 / 5 / 25 children that
share a result set and carry 100 ids of their own each."""
import numpy as np

import oracle as O

NUM_CHILDREN_CASES = [2, 5, 25]
RESULT_SET_CASES = [
    [1, 2, 3, 40, 50],
    [5, 6, 7, 24, 25, 46, 47, 48, 49, 50, 51, 234, 2345, 3456, 4567, 5678, 6789, 7890, 8901, 9012, 12345, 23456, 34567,
     45678, 56789],
    [9, 25, 30, 40, 50, 60, 70, 80, 90, 100, 110, 120, 130, 140, 150, 160, 170, 180, 190, 200, 210, 220, 230, 240, 250],
]


def create_children(num_children, result_set):
    """intersection.rs:30-52: every child = result set + 100 ids no other child has (sorted, de-duplicated)."""
    children, nxt = [], 1
    for _ in range(num_children):
        ids = list(result_set)
        for _ in range(100):
            ids.append(nxt)
            nxt += 1
        children.append(sorted(set(ids)))
    return children


def to_index(ids, codec=O.C_FREQS_ONLY, freq_of=lambda d: 1 + d % 7):
    ii = O.InvertedIndex(codec)
    a = np.asarray(ids, dtype=np.uint64)
    ii.add_many(a, np.asarray([freq_of(int(d)) for d in ids], dtype=np.uint32))
    return ii
