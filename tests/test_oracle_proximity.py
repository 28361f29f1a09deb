"""Pins the oracle's proximity restatement (oracle/postings_oracle.c oracle_within_range / oracle_min_offset_delta /
oracle_intersect_ex) on the reference's own known answers:
  index_result/src/core/proximity.rs:320-470 (unit tests of within_range_in_order / _unordered and the k-way merge),
  rqe_iterators/tests/integration/intersection.rs:1164-1370 (slop / in_order / slop_and_order / retry-hits-EOF),
and, where oracle/_ref ships it, on the reference's compiled IndexResult_MinOffsetDelta."""
import numpy as np
import pytest

import oracle as O


def T(*deltas):                       # a term child: its varint-delta offset bytes (all values < 128 here)
    return (False, [bytes(deltas)])


VW1, VW2 = T(1, 8, 4, 3, 6), T(4, 3, 25)     # positions 1 9 13 16 22 / 4 7 32 (proximity.rs:317-318)


def test_within_range_in_order_kats():
    # proximity.rs:327-337
    assert [O.within_range([VW1, VW2], s, True) for s in range(6)] == [False, False, True, True, True, True]
    assert O.within_range([T(3), T(4)], 0, True)                       # :340-347 exact consecutive
    assert not O.within_range([T(10), T(5)], 100, True)                # :349-357 out of order
    # (proximity.rs:359-365 feeds an empty iterator straight into within_range_in_order -> false; through
    # is_within_range a term WITHOUT offsets is filtered out first, :284-292, and one remaining child is trivially in range)
    assert O.within_range([(False, [b""]), T(5)], 100, True)


def test_within_range_unordered_kats():
    # proximity.rs:369-395
    assert [O.within_range([VW1, VW2], s, False) for s in range(5)] == [False, True, True, True, True]
    assert not O.within_range([T(10), T(5)], 3, False) and O.within_range([T(10), T(5)], 4, False)


def test_merge_children_kats():
    # proximity.rs:410-466: a union child merges its leaves' positions in ascending order; checked through the slop of
    # (merged child, single position p): |first merged position >= ... | -- and directly through in-order windows
    merged = (True, [bytes([2, 3, 4]), bytes([1, 3, 3])])              # positions 1 2 4 5 7 9
    for p, exp in ((3, True), (6, True), (8, True)):                   # a term at p directly behind a merged position
        assert O.within_range([merged, T(p)], 0, True) is exp
    assert not O.within_range([merged, T(11)], 0, True) and O.within_range([merged, T(11)], 1, True)
    three = (True, [bytes([5]), bytes([2, 6]), bytes([1, 3])])         # positions 1 2 4 5 8
    assert O.within_range([three, T(9)], 0, True) and not O.within_range([three, T(7)], 0, True)
    assert O.within_range([three, T(7)], 1, True)
    empty = (True, [b"", b""])
    assert not O.within_range([empty, T(5)], 100, False)               # an aggregate child takes part even when empty


def mock(codec, docs, positions):
    ii = O.InvertedIndex(codec)
    for d, p in zip(docs, positions):
        ii.add(d, 1, 1, O.varint_encode(p))
    return ii


@pytest.mark.parametrize("codec", [O.C_FULL, O.C_OFFSETS_ONLY, O.C_FREQS_OFFSETS, O.C_FIELDS_OFFSETS])
def test_intersection_slop_and_order_kats(codec):
    # intersection.rs:1173-1194: foo docs 1..4 at positions 1 1 2 1, bar docs 1 3 4 at positions 2 1 3
    foo, bar = mock(codec, [1, 2, 3, 4], [1, 1, 2, 1]), mock(codec, [1, 3, 4], [2, 1, 3])
    assert O.intersect_ex([foo, bar], 0, False)[0].tolist() == [1, 3]          # :1196-1248
    assert O.intersect_ex([foo, bar], None, True)[0].tolist() == [1, 4]        # :1250-1296
    assert O.intersect_ex([foo, bar], 0, True)[0].tolist() == [1]              # :1298-1340
    assert O.intersect_ex([foo, bar], None, False)[0].tolist() == [1, 3, 4]
    assert O.intersect_ex([foo, bar], 1, False)[0].tolist() == [1, 3, 4]
    # :1342-1368 relevancy retry hits EOF
    f2, b2 = mock(codec, [1, 2], [3, 1]), mock(codec, [1], [1])
    assert O.intersect_ex([f2, b2], None, True)[0].tolist() == []
    # slop of the hits (IndexResult_MinOffsetDelta): adjacent -> 1, distance 2 -> 2
    ids, _, sl = O.intersect_ex([foo, bar], None, False)
    assert sl.tolist() == [1, 1, 2]


def test_min_offset_delta_against_the_reference_and_the_scoring_oracle():
    """Same function three ways: this restatement over offset BYTES, oracle_slop over decoded positions (the scorers'
    oracle), and -- where oracle/_ref ships it -- the reference's compiled IndexResult_MinOffsetDelta."""
    from oracle import ext as X
    rng = np.random.default_rng(4)
    host = None
    if X.have_ref():
        host = X.Host()
        host.load_ref()
    for _ in range(300):
        n = int(rng.integers(1, 6))
        pos = [sorted(set(int(x) for x in rng.integers(1, 80, int(rng.integers(0, 6))))) for _ in range(n)]

        def enc(ps):
            out, last = b"", 0
            for p in ps:
                out += O.varint_encode(p - last)
                last = p
            return out
        got = O.min_offset_delta([(False, [enc(p)]) for p in pos])
        node = O.intersection([O.term(1, 1.0, 1.0, offsets=p) for p in pos])
        assert got == O.lib.oracle_slop(node.ptr)
        if host is not None:
            t = X.Tree(("intersection", 1.0, [("term", 1.0, 1, 1.0, 1.0, "t", p) for p in pos]))
            assert got == host.ref_slop(t)
