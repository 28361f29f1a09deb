"""CPU tests pinning the union / NOT restatements (oracle/postings_oracle.c oracle_union, oracle_not) to the
reference's own iterator tests."""
import numpy as np
import pytest

import oracle as O


def ids_list(docs, freqs=None):
    ii = O.InvertedIndex(O.C_FREQS_ONLY)
    d = np.asarray(docs, np.uint64)
    if d.size:
        ii.add_many(d, np.asarray(freqs if freqs is not None else np.ones(d.size), np.uint32))
    return ii


def test_not_read_skips_child_docs():
    # rqe_iterators/tests/integration/not.rs:49-62: child [2,4,7], max 10 -> [1,3,5,6,8,9,10]
    assert O.not_list(ids_list([2, 4, 7]), 10).tolist() == [1, 3, 5, 6, 8, 9, 10]
    # :79-103 empty child behaves like a wildcard ; :105-123 child covering the range yields nothing
    assert O.not_list(ids_list([]), 5).tolist() == [1, 2, 3, 4, 5]
    assert O.not_list(ids_list([1, 2, 3, 4, 5]), 5).tolist() == []
    # child entries beyond max_doc_id are irrelevant
    assert O.not_list(ids_list([2, 50]), 4).tolist() == [1, 3, 4]


def test_not_optimized_against_a_universe():
    # not_optimized.rs: the complement is taken inside the wildcard list of existing documents
    uni = ids_list([1, 2, 3, 5, 8, 13, 21])
    assert O.not_list(ids_list([2, 3, 4, 13]), 20, universe=uni).tolist() == [1, 5, 8]


def test_union_edge_cases():
    # rqe_iterators/tests/integration/union_common.rs:243-284 disjoint / overlapping, :366-384 empty children mixed in
    ids, fr, _ = O.union_lists([ids_list([]), ids_list([10, 20, 30]), ids_list([15, 25, 35])])
    assert ids.tolist() == [10, 15, 20, 25, 30, 35]
    ids, fr, _ = O.union_lists([ids_list([1, 3, 5], [1, 2, 3]), ids_list([3, 4, 5], [7, 8, 9])])
    assert ids.tolist() == [1, 3, 4, 5] and fr.tolist() == [[1, 2, 0, 3], [0, 7, 8, 9]]
    assert len(O.union_lists([ids_list([]), ids_list([])])[0]) == 0
    ids, fr, _ = O.union_lists([ids_list([4, 9], [2, 3])])
    assert ids.tolist() == [4, 9] and fr.tolist() == [[2, 3]]


@pytest.mark.parametrize("num_children", [2, 5, 10])
@pytest.mark.parametrize("base", [[1, 2, 3, 40, 50],
                                  [5, 6, 7, 24, 25, 46, 47, 48, 49, 50, 51, 234, 2345],
                                  [9, 25, 30, 40, 50, 60, 70, 80, 90, 100, 110, 120, 130]])
def test_union_read_fixture_cases(num_children, base):
    # rqe_iterators/tests/integration/union_common.rs:40-98 with utils/mod.rs:184-197 (create_union_children):
    # child i holds base * i, i = 1..num_children; reading the union yields the sorted set of all of them
    children = [[x * i for x in base] for i in range(1, num_children + 1)]
    expected = sorted(set(x for c in children for x in c))
    lists = []
    for c in children:
        ii = O.InvertedIndex(O.C_FREQS_ONLY)
        ii.add_many(np.asarray(c, dtype=np.uint64), np.full(len(c), 3, dtype=np.uint32))
        lists.append(ii)
    ids, fr, _ = O.union_lists(lists)
    assert ids.tolist() == expected
    for li, c in enumerate(children):                      # matched children carry their freq, the others 0
        cs = set(c)
        assert fr[li].tolist() == [3 if d in cs else 0 for d in expected]
