"""GPU tests of the specialised qint decoder (postings_kernels.hip: decode_qint_block_lds -- one LDS window per record,
every non-wide qint layout) and of the sub-block sync points (decode-per-query mode: the first decode of a list parses a
block per lane and leaves a sync point every 16 records, later decodes share a block among eight lanes): lists whose
records mix every field length (1 .. 4 bytes per value, so 3 .. 9 bytes per two-field record, every alignment of a block's
first byte), block counts around the blocks-per-wavefront boundaries, lists of one to five records, blocks of any size.
Decoded ids and values must equal the oracle's reader (reference src/redisearch_rs/qint/src/lib.rs:139-214,
inverted_index/src/codec/freqs_only.rs, fields_only.rs) bit for bit, decode after decode, with the sync points on and
off."""
import numpy as np
import pytest

import oracle as O
from redisearch_amd import search as S
from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu


def mixed_values(rng, n, weights):
    """n values whose encoded length is 1 .. 4 bytes with the given weights"""
    lens = rng.choice(4, size=n, p=weights)
    lo = np.array([1, 1 << 8, 1 << 16, 1 << 24], dtype=np.uint64)[lens]
    hi = np.array([1 << 8, 1 << 16, 1 << 24, 1 << 32], dtype=np.uint64)[lens]
    return (lo + (rng.random(n) * (hi - lo)).astype(np.uint64)).astype(np.uint64)


def make_list(codec, n, seed):
    rng = np.random.default_rng(seed)
    deltas = mixed_values(rng, n, [0.6, 0.4, 0.0, 0.0])
    # a limited number of 3- and 4-byte deltas: the list must stay inside 2^32 doc ids (the device path's range)
    three = rng.choice(n, size=min(n, 100), replace=False)
    deltas[three] = (1 << 16) + (rng.random(three.size) * ((1 << 24) - (1 << 16))).astype(np.uint64)
    four = rng.choice(n, size=min(n, 40), replace=False)
    deltas[four] = (1 << 24) + (rng.random(four.size) * (1 << 22)).astype(np.uint64)
    docs = np.cumsum(deltas).astype(np.uint64)
    vals = mixed_values(rng, n, [0.4, 0.2, 0.2, 0.2]).astype(np.uint32)
    ii = O.InvertedIndex(codec)
    if codec == O.C_FREQS_ONLY:
        ii.add_many(docs, vals)
    else:
        for d, v in zip(docs.tolist(), vals.tolist()):
            ii.add(d, 1, v)
    return ii


@pytest.mark.parametrize("codec", [O.C_FREQS_ONLY, O.C_FIELDS_ONLY])
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 99, 100, 101, 257, 6_399, 6_400, 6_401, 20_011])
def test_two_field_lists_decode_bit_for_bit_with_and_without_sync_points(codec, n):
    lib = V.load()
    ii = make_list(codec, n, 1000 * codec + n)
    want = ii.decode_all()
    fl = ii.flatten()
    try:
        lib.RSGPU_SetTuning(b"cache_decoded", 0)      # (lists uploaded in decode-per-query mode carry sync points)
        for sync in (1, 0):
            lib.RSGPU_SetTuning(b"decode_sync", sync)
            p = S.Postings.from_flat(fl)
            try:
                # the first decode parses a block per lane and leaves the sync points, the later ones share a block among
                # eight lanes
                for rep in range(3):
                    ids, fr, mk = p.decode()
                    assert np.array_equal(ids, want[0]), (sync, rep, "ids")
                    if codec == O.C_FREQS_ONLY:
                        assert np.array_equal(fr, want[1]), (sync, rep, "freqs")
                    else:
                        assert np.array_equal(mk, want[2]), (sync, rep, "masks")
            finally:
                p.free()
    finally:
        lib.RSGPU_SetTuning(b"cache_decoded", 1)
        lib.RSGPU_SetTuning(b"decode_sync", 1)


@pytest.mark.parametrize("codec", [O.C_FREQS_ONLY, O.C_FIELDS_ONLY])
def test_sync_points_of_wavefronts_that_do_not_fit_the_staging_buffer(codec):
    """records of 6-7 bytes: 64 blocks of 100 are ~42 KiB, more than the 30 KiB a wavefront stages, so the FIRST decode of
    the list parses every block straight from memory -- and still has to leave the sync points the later decodes (eight
    lanes per block) start from (it did not: the lanes started from whatever the allocation held)"""
    lib = V.load()
    n = 20_011
    rng = np.random.default_rng(77 + codec)
    docs = np.cumsum(rng.integers(256, 60_000, n)).astype(np.uint64)
    vals = rng.integers(1 << 24, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    ii = O.InvertedIndex(codec)
    if codec == O.C_FREQS_ONLY:
        ii.add_many(docs, vals)
    else:
        for d, v in zip(docs.tolist(), vals.tolist()):
            ii.add(d, 1, v)
    want = ii.decode_all()
    try:
        lib.RSGPU_SetTuning(b"cache_decoded", 0)
        p = S.Postings.from_flat(ii.flatten())
        try:
            for rep in range(3):
                ids, fr, mk = p.decode()
                assert np.array_equal(ids, want[0]), (rep, "ids")
                assert np.array_equal(fr if codec == O.C_FREQS_ONLY else mk, want[1 if codec == O.C_FREQS_ONLY else 2]), rep
        finally:
            p.free()
    finally:
        lib.RSGPU_SetTuning(b"cache_decoded", 1)


def test_every_length_pair_at_every_alignment():
    """16 length pairs x 8 alignments of the record inside its 8-byte FIFO word: a list of 3-record blocks is not possible
    (blocks hold 100), so the pairs are cycled with a period coprime to 8 and the list is long enough to meet them all"""
    lib = V.load()
    pairs = [(a, b) for a in range(4) for b in range(4)]
    n = 16 * 8 * 9 + 7
    rng = np.random.default_rng(8)
    order = [pairs[(i * 7) % 16] for i in range(n)]
    lo = [1, 1 << 8, 1 << 16, 1 << 24]
    deltas = np.array([lo[a] + int(rng.integers(0, 200)) for a, _ in order], dtype=np.uint64)
    # (4-byte deltas every 16th record would leave the 32-bit id range: keep a handful, shrink the rest to 3 bytes)
    big = np.flatnonzero(deltas >= (1 << 24))
    deltas[big[40:]] = (1 << 16) + 5
    docs = np.cumsum(deltas).astype(np.uint64)
    freqs = np.array([lo[b] + int(rng.integers(0, 200)) for _, b in order], dtype=np.uint32)
    ii = O.InvertedIndex(O.C_FREQS_ONLY)
    ii.add_many(docs, freqs)
    want = ii.decode_all()
    assert np.array_equal(want[0], docs) and np.array_equal(want[1], freqs)
    for cache in (0, 1):
        lib.RSGPU_SetTuning(b"cache_decoded", cache)
        p = S.Postings.from_flat(ii.flatten())
        try:
            for rep in range(2):
                ids, fr, _ = p.decode()
                assert np.array_equal(ids, docs) and np.array_equal(fr, freqs), (cache, rep)
        finally:
            p.free()
            lib.RSGPU_SetTuning(b"cache_decoded", 1)


def qint2(delta, value):
    """one two-field qint record (reference qint/src/lib.rs:139-214)"""
    out, hdr = bytearray([0]), 0
    for i, v in enumerate((int(delta), int(value))):
        ln = 1 if v < (1 << 8) else 2 if v < (1 << 16) else 3 if v < (1 << 24) else 4
        hdr |= (ln - 1) << (2 * i)
        out += v.to_bytes(ln, "little")
    out[0] = hdr
    return bytes(out)


def test_blocks_of_any_size_through_the_sync_points():
    """blocks other than the reference's 100 records (an uploader may hand any block list): fewer than one 16-record
    segment, exactly 7 and 8 segments, more than the sync points cover (the last lane takes the rest)"""
    lib = V.load()
    rng = np.random.default_rng(12)
    sizes = [250, 17, 112, 113, 128, 1, 16, 15, 100, 1000]
    first, last, nent, off, data = [], [], [], [0], bytearray()
    docs_all, freqs_all, doc = [], [], 10
    for n in sizes:
        deltas = mixed_values(rng, n, [0.7, 0.3, 0.0, 0.0])
        freqs = mixed_values(rng, n, [0.4, 0.2, 0.2, 0.2]).astype(np.uint32)
        ids = doc + np.cumsum(deltas)
        doc = int(ids[-1])
        first.append(int(ids[0]))
        last.append(doc)
        nent.append(n)
        prev = int(ids[0])                              # the first record of a block is relative to first_doc_id
        for d, f in zip(ids.tolist(), freqs.tolist()):
            data += qint2(d - prev, f)
            prev = d
        off.append(len(data))
        docs_all += ids.tolist()
        freqs_all += freqs.tolist()
    fl = dict(codec=O.C_FREQS_ONLY, first=np.array(first, np.uint64), last=np.array(last, np.uint64),
              num_entries=np.array(nent, np.uint32), offset=np.array(off, np.uint64), bytes=np.frombuffer(bytes(data), np.uint8))
    lib.RSGPU_SetTuning(b"cache_decoded", 0)
    p = S.Postings.from_flat(fl)
    try:
        for rep in range(3):
            ids, fr, _ = p.decode()
            assert ids.tolist() == docs_all and fr.tolist() == freqs_all, rep
    finally:
        p.free()
        lib.RSGPU_SetTuning(b"cache_decoded", 1)


@pytest.mark.parametrize("codecs", [(O.C_FREQS_ONLY, O.C_FREQS_ONLY), (O.C_FREQS_ONLY, O.C_FIELDS_ONLY),
                                    (O.C_FREQS_ONLY, O.C_DOCIDS_ONLY), (O.C_FREQS_ONLY, O.C_FREQS_ONLY, O.C_FREQS_ONLY)])
def test_the_lists_of_a_query_decoded_in_one_launch(codecs):
    """decode-per-query mode: once their sync points are there, two qint lists of a query are decoded by ONE launch
    (decode_blocks_pair_kernel; a third list, or a list of another record kind, keeps its own launch).  Same intersection --
    ids and frequencies -- as the oracle's, query after query, with the pairing on and off."""
    lib = V.load()
    rng = np.random.default_rng(len(codecs) * 31 + sum(codecs))
    lists = []
    for j, codec in enumerate(codecs):
        docs = np.unique(rng.integers(1, 300_000, 60_000 + 35_000 * j)).astype(np.uint64)
        ii = O.InvertedIndex(codec)
        if codec == O.C_FIELDS_ONLY:
            for d in docs.tolist():
                ii.add(d, 1, int(rng.integers(1, 1 << 20)))
        else:
            ii.add_many(docs, rng.integers(1, 70_000, docs.size).astype(np.uint32))
        lists.append(ii)
    oi, of, _ = O.intersect(lists)
    try:
        lib.RSGPU_SetTuning(b"cache_decoded", 0)
        g = [S.Postings.from_flat(l.flatten()) for l in lists]
        for pair in (1, 0, 1):
            lib.RSGPU_SetTuning(b"decode_pair", pair)
            for rep in range(3):     # (the first query of a list writes its sync points, one lane per block)
                h = S.intersect(g)
                gi, gf = h.read()
                assert gi.tolist() == oi.tolist(), (pair, rep)
                assert gf.tolist() == of.tolist(), (pair, rep)
                h.free()
        for p in g:
            p.free()
    finally:
        lib.RSGPU_SetTuning(b"cache_decoded", 1)
        lib.RSGPU_SetTuning(b"decode_pair", 1)


@pytest.mark.parametrize("codec", [O.C_FULL, O.C_FREQS_OFFSETS, O.C_FIELDS_OFFSETS, O.C_OFFSETS_ONLY])
@pytest.mark.parametrize("n,max_off", [(3, 4), (101, 4), (6_401, 6), (20_011, 3), (5_000, 120)])
def test_lists_with_inline_offsets_through_the_sync_points(codec, n, max_off):
    """Round 4: the qint layouts WITH inline offsets (Full is FT.CREATE's default) take the sync points too -- first decode a
    block per lane (32 or 16 blocks per wavefront, so that the longer blocks still fit the staging buffer), later decodes
    eight lanes per block.  ids / freqs / masks equal the oracle's reader decode after decode; the offsets index the decode
    leaves behind (where every record's position list lies) is checked through what reads it: slop / in-order intersections
    against the oracle.  max_off 120: blocks of ~12 KiB -- eight of them do not fit, such a list gets no sync points."""
    lib = V.load()
    rng = np.random.default_rng(31 * codec + n)

    def make(seed_shift):
        docs = np.unique(rng.integers(1, 4 * n + 10, n))
        ii = O.InvertedIndex(codec)
        for d in docs.tolist():
            pos = sorted(set(int(x) for x in rng.integers(1, 40 * max_off, int(rng.integers(0, max_off)))))
            offs, last = b"", 0
            for p in pos:
                offs += O.varint_encode(p - last)
                last = p
            ii.add(d, int(rng.integers(1, 300)), int(rng.integers(1, 2 ** 31)), offs)
        return ii
    a, b = make(0), make(1)
    want = a.decode_all()
    try:
        lib.RSGPU_SetTuning(b"cache_decoded", 0)
        ga, gb = S.Postings.from_flat(a.flatten()), S.Postings.from_flat(b.flatten())
        try:
            for rep in range(3):
                ids, fr, mk = ga.decode()
                assert np.array_equal(ids, want[0]), (rep, "ids")
                if codec in (O.C_FULL, O.C_FREQS_OFFSETS):
                    assert np.array_equal(fr, want[1]), (rep, "freqs")
                if codec in (O.C_FULL, O.C_FIELDS_OFFSETS):
                    assert np.array_equal(mk, want[2]), (rep, "masks")
            for max_slop, in_order in ((None, False), (0, False), (3, False), (None, True), (5, True)):
                for rep in range(2):
                    gi, gf = S.intersect([ga, gb], max_slop=max_slop, in_order=in_order).read()
                    oi, of, _ = O.intersect_ex([a, b], max_slop, in_order)
                    assert gi.tolist() == oi.tolist() and gf.tolist() == of.tolist(), (max_slop, in_order, rep)
        finally:
            ga.free()
            gb.free()
    finally:
        lib.RSGPU_SetTuning(b"cache_decoded", 1)
