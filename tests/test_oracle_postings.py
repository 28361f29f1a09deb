"""Pins oracle/postings_oracle.c against the reference's byte-exact codec tests, reader and
intersection tests. Integer work: everything must match exactly. CPU only."""
import numpy as np
import pytest

import oracle as O

U32, U16 = 2 ** 32 - 1, 2 ** 16 - 1


def enc(codec, delta, freq=1, mask=1, offs=b""):
    """One record written into an empty block whose delta base we control via two adds."""
    ii = O.InvertedIndex(codec)
    base = 4294967296 - delta
    ii.add(base, 1, 1, b"")
    first_len = len(ii.flatten()["bytes"])
    ii.add(4294967296, freq, mask, offs)
    return bytes(ii.flatten()["bytes"][first_len:])


def test_qint_doc_examples_and_roundtrip():
    # reference src/redisearch_rs/qint/src/lib.rs:55-112
    for vals in ([1, 2], [256, 65536, 7], [U32, U32, U32, U32], [0, 0], [70000, 3, 255, 256]):
        b = O.qint_encode(vals)
        out, k = O.qint_decode(b, len(vals))
        assert out == vals and k == len(b)
    assert O.qint_decode(b"\x00\x00", 2)[1] == 0          # one byte short -> UnexpectedEof
    assert O.qint_decode(b"", 2)[1] == 0


def test_encode_freqs_only_kats():
    # reference inverted_index/tests/integration/codec/freqs_only.rs:26-49 (frequency, delta, bytes)
    tests = [(0, 0, [0, 0, 0]), (0, 1, [0, 1, 0]), (2, 0, [0, 0, 2]), (2, 1, [0, 1, 2]),
             (256, 0, [4, 0, 0, 1]), (256, 256, [5, 0, 1, 0, 1]), (2, 65536, [2, 0, 0, 1, 2]),
             (U16 + 1, U16 + 1, [10, 0, 0, 1, 0, 0, 1]), (2, U32, [3, 255, 255, 255, 255, 2]),
             (U32, U32, [15] + [255] * 8)]
    for freq, delta, expected in tests:
        assert O.qint_encode([delta, freq]) == bytes(expected)


def test_encode_full_kats():
    # reference codec/full.rs:21-62 (delta, freq, mask, offsets, bytes)
    tests = [(0, 1, 1, [1, 2, 3], [0, 0, 1, 1, 3, 1, 2, 3]),
             (10, 5, U32, [1, 2, 3, 4], [48, 10, 5, 255, 255, 255, 255, 4, 1, 2, 3, 4]),
             (256, 1, 1, [1, 2, 3], [1, 0, 1, 1, 1, 3, 1, 2, 3]),
             (65536, 1, 1, [1, 2, 3], [2, 0, 0, 1, 1, 1, 3, 1, 2, 3]),
             (U16, 1, 1, [1, 2, 3], [1, 255, 255, 1, 1, 3, 1, 2, 3]),
             (U32, 1, 1, [1, 2, 3], [3, 255, 255, 255, 255, 1, 1, 3, 1, 2, 3]),
             (U32, U32, U32, [1] * 100, [63] + [255] * 12 + [100] + [1] * 100)]
    for delta, freq, mask, offs, expected in tests:
        if delta == 0:
            ii = O.InvertedIndex(O.C_FULL)
            ii.add(4294967296, freq, mask, bytes(offs))
            got = bytes(ii.flatten()["bytes"])
        else:
            got = enc(O.C_FULL, delta, freq, mask, bytes(offs))
        assert got == bytes(expected)


def test_encode_other_codecs_kats():
    # codec/{freqs_fields,fields_only,offsets_only,freqs_offsets,fields_offsets}.rs
    assert enc(O.C_FREQS_FIELDS, 10, 5, U32) == bytes([48, 10, 5, 255, 255, 255, 255])
    assert enc(O.C_FREQS_FIELDS, 65536, 1, 1) == bytes([2, 0, 0, 1, 1, 1])
    assert enc(O.C_FREQS_FIELDS, U32, U32, U32) == bytes([63] + [255] * 12)
    assert enc(O.C_FIELDS_ONLY, 10, 1, U32) == bytes([12, 10, 255, 255, 255, 255])
    assert enc(O.C_FIELDS_ONLY, 256, 1, 1) == bytes([1, 0, 1, 1])
    assert enc(O.C_OFFSETS_ONLY, 10, 1, 1, bytes([1, 2, 3, 4])) == bytes([0, 10, 4, 1, 2, 3, 4])
    assert enc(O.C_OFFSETS_ONLY, U16, 1, 1, bytes([1, 2, 3])) == bytes([1, 255, 255, 3, 1, 2, 3])
    assert enc(O.C_FREQS_OFFSETS, 256, 3, 1, bytes([1, 2, 3])) == bytes([1, 0, 1, 3, 3, 1, 2, 3])
    assert enc(O.C_FREQS_OFFSETS, U32, 6, 1, bytes([1, 2, 3])) == bytes([3, 255, 255, 255, 255, 6, 3, 1, 2, 3])
    assert enc(O.C_FIELDS_OFFSETS, 10, 1, U32, bytes([1, 2, 3, 4])) == bytes([12, 10, 255, 255, 255, 255, 4, 1, 2, 3, 4])
    assert enc(O.C_FIELDS_OFFSETS, 65536, 1, 1, bytes([1, 2, 3])) == bytes([2, 0, 0, 1, 1, 3, 1, 2, 3])


def test_varint_and_docids_kats():
    # codec/doc_ids_only.rs:19-26 and raw_doc_ids_only.rs:19-26
    for v, b in ((0, [0]), (10, [10]), (256, [129, 0]), (65536, [130, 255, 0]), (U16, [130, 254, 127]),
                 (U32, [142, 254, 254, 254, 127])):
        assert O.varint_encode(v) == bytes(b)
        assert O.varint_decode(bytes(b)) == (v, len(b))
    assert O.varint_decode(b"") == (0, 0)
    for v, b in ((10, [10, 0, 0, 0]), (256, [0, 1, 0, 0]), (65536, [0, 0, 1, 0]), (U16, [255, 255, 0, 0])):
        assert enc(O.C_RAW_DOCIDS, v) == bytes(b)
    # varint big field-mask example (fields_only.rs:91): 10, u32::MAX -> [10, 142,254,254,254,127]
    assert O.varint_encode(10) + O.varint_encode(U32) == bytes([10, 142, 254, 254, 254, 127])


def test_seek_freqs_only():
    # codec/freqs_only.rs:121-160: docs 10,20,30,35,55,60 ; seek 30 -> 30 (freq 3); seek 40 -> 55
    ii = O.InvertedIndex(O.C_FREQS_ONLY)
    for d, f in ((10, 1), (20, 2), (30, 3), (35, 4), (55, 5), (60, 6)):
        ii.add(d, f)
    r = ii.reader()
    assert r.seek(30) == (30, 3) and r.seek(40) == (55, 5) and r.next() == (60, 6)
    assert r.next() is None and r.seek(61) is None


@pytest.mark.parametrize("codec,per_block", [(O.C_FULL, 100), (O.C_FREQS_ONLY, 100), (O.C_DOCIDS_ONLY, 1000),
                                             (O.C_RAW_DOCIDS, 1000)])
def test_block_layout_and_roundtrip(codec, per_block):
    # index/core.rs:250-330 (take_block / delta base) ; codec/mod.rs RECOMMENDED_BLOCK_ENTRIES
    rng = np.random.default_rng(11)
    docs = np.cumsum(rng.integers(1, 50, 2501)).astype(np.uint64)
    freqs = rng.integers(1, 300, docs.size).astype(np.uint32)
    ii = O.InvertedIndex(codec)
    ii.add_many(docs, freqs)
    assert ii.add(int(docs[-1])) == 0                       # same doc id again: skipped
    fl = ii.flatten()
    assert ii.num_blocks == -(-docs.size // per_block)
    assert fl["num_entries"].tolist() == [per_block] * (ii.num_blocks - 1) + [docs.size - per_block * (ii.num_blocks - 1)]
    assert fl["first"].tolist() == docs[::per_block].tolist()
    assert fl["last"][-1] == docs[-1]
    ids, fr, _ = ii.decode_all()
    assert ids.tolist() == docs.tolist()
    if codec in (O.C_FULL, O.C_FREQS_ONLY):
        assert fr.tolist() == freqs.tolist()
    # every skip target lands on the first doc >= target (reader/core.rs seek_record)
    r = ii.reader()
    for t in (1, int(docs[3]), int(docs[3]) + 1, int(docs[1500]), int(docs[-1])):
        got = r.seek(t)
        assert got[0] == int(docs[np.searchsorted(docs, t)])
    assert r.seek(int(docs[-1]) + 1) is None


def test_delta_overflow_opens_new_block():
    ii = O.InvertedIndex(O.C_FREQS_ONLY)
    ii.add(5, 1)
    ii.add(5 + 2 ** 32 + 7, 2)
    fl = ii.flatten()
    assert ii.num_blocks == 2 and fl["first"].tolist() == [5, 5 + 2 ** 32 + 7]
    assert ii.decode_all()[0].tolist() == [5, 5 + 2 ** 32 + 7]


def populate(size, step, codec=O.C_FULL):
    """createPopulateTermsInvIndex (reference tests/cpptests/index_utils.cpp:30-62)."""
    ii = O.InvertedIndex(codec)
    doc = step
    for i in range(size):
        offs = b"".join(O.varint_encode(n if k == 0 else 1) for k, n in enumerate(range(step, step + i % 4)))
        ii.add(doc, 1, 1, offs)
        doc += step
    return ii


def test_intersection_cpp_kat():
    # reference tests/cpptests/test_cpp_index.cpp:542-601: steps 4 and 2 over 100000 entries =>
    # 50000 hits, doc ids (2c+2)*2, freq 2
    w, w2 = populate(100000, 4), populate(100000, 2)
    ids, fr, _ = O.intersect([w, w2])
    assert ids.size == 50000
    assert ids.tolist() == [(c * 2 + 2) * 2 for c in range(50000)]
    assert (fr.sum(0) == 2).all()
    assert ids[-1] == 200000 and 8 in ids and ids[np.searchsorted(ids, 8) + 1] == 12


def test_intersection_matches_set_semantics():
    rng = np.random.default_rng(7)
    for nl in (2, 3, 4):
        for codec in (O.C_FREQS_ONLY, O.C_DOCIDS_ONLY, O.C_FULL):
            lists, sets = [], []
            for _ in range(nl):
                docs = np.unique(rng.integers(1, 6000, rng.integers(50, 4000))).astype(np.uint64)
                ii = O.InvertedIndex(codec)
                ii.add_many(docs, rng.integers(1, 20, docs.size).astype(np.uint32))
                lists.append(ii)
                sets.append(docs)
            want = sets[0]
            for s in sets[1:]:
                want = np.intersect1d(want, s)
            ids, fr, _ = O.intersect(lists)
            assert ids.tolist() == want.tolist()
            if codec != O.C_DOCIDS_ONLY:
                for li, l in enumerate(lists):
                    all_ids, all_fr, _ = l.decode_all()
                    assert fr[li].tolist() == all_fr[np.searchsorted(all_ids, ids)].tolist()


def test_intersection_edge_cases():
    a, b, e = O.InvertedIndex(O.C_FREQS_ONLY), O.InvertedIndex(O.C_FREQS_ONLY), O.InvertedIndex(O.C_FREQS_ONLY)
    for d in (1, 5, 9):
        a.add(d, d)
    for d in (2, 6, 10):
        b.add(d, d)
    assert O.intersect([a, b])[0].size == 0            # disjoint
    assert O.intersect([a, e])[0].size == 0            # empty child
    assert O.intersect([a, a])[0].tolist() == [1, 5, 9]
    assert O.intersect([a])[0].tolist() == [1, 5, 9]   # single child


def test_decode_offsets():
    buf = b"".join(O.varint_encode(x) for x in (3, 1, 200, 16511))
    out = np.zeros(8, np.uint32)
    b = np.frombuffer(buf, dtype=np.uint8)
    n = O.lib.oracle_decode_offsets(O._p(b), len(b), O._p(out), 8)
    assert out[:n].tolist() == [3, 4, 204, 16715]
