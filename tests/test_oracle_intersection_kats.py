"""intersection.rs known answers against the CPU oracle: read_all_combinations (:82-149), empty / single-element
result sets (:325-354), single child (:411-434), many children (:524-556), large doc-id gaps (:876-910),
overlapping children (:912-933), num_estimated = the smallest child (:1015-1035)."""
import numpy as np
import pytest

import oracle as O
from tests.intersection_cases import NUM_CHILDREN_CASES, RESULT_SET_CASES, create_children, to_index


@pytest.mark.parametrize("num_children", NUM_CHILDREN_CASES)
@pytest.mark.parametrize("case", range(len(RESULT_SET_CASES)))
def test_read_all_combinations(num_children, case):
    rs = RESULT_SET_CASES[case]
    children = create_children(num_children, rs)
    want = sorted(set.intersection(*map(set, children)))
    assert want == rs                                     # what the reference asserts, doc id by doc id
    lists = [to_index(c) for c in children]
    ids, fr, _ = O.intersect(lists)
    assert ids.tolist() == want
    for li, c in enumerate(children):                     # every child's own frequency travels with the hit
        assert fr[li].tolist() == [1 + d % 7 for d in want]


def test_empty_and_single_element_result_sets():
    for rs in ([], [3000]):
        lists = [to_index(c) for c in create_children(3, rs)]
        assert O.intersect(lists)[0].tolist() == rs


def test_single_child_and_many_children():
    ids = [10, 20, 30, 40, 50]
    assert O.intersect([to_index(ids)])[0].tolist() == ids
    rs = [5000, 6000, 7000]
    lists = [to_index(c) for c in create_children(25, rs)]
    assert O.intersect(lists)[0].tolist() == rs


def test_large_doc_id_gaps_and_overlapping_children():
    big = [1, 1_000_000, 2_000_000_000, 4_000_000_000]
    a, b = to_index(big), to_index([1, 500, 1_000_000, 3_000_000_000, 4_000_000_000])
    assert O.intersect([a, b])[0].tolist() == [1, 1_000_000, 4_000_000_000]
    c1, c2, c3 = to_index([1, 2, 3, 4, 5, 6, 7, 8, 9, 10]), to_index([2, 4, 6, 8, 10]), to_index([4, 8, 12])
    assert O.intersect([c1, c2, c3])[0].tolist() == [4, 8]
