"""GPU: the dense-group decoder (round 6, postings_kernels.hip decode_dense_kernel / decode_dense_group): sixteen consecutive blocks
of a qint list WITHOUT inline offsets whose records are all of the minimal length (every control byte zero: deltas and values
below 256 -- the lists that cost the most to decode) are decoded data-parallel, four records per lane and step, the doc ids from a
wave-wide prefix + a per-block constant; any other group goes through the sync points, eight lanes per block.  Decoded ids /
frequencies / field masks equal the oracle's reader (reference src/redisearch_rs/qint/src/lib.rs:139-214,
inverted_index/src/codec/freqs_only.rs, fields_only.rs, freqs_fields.rs) bit for bit with the knob on and off: every layout, list
lengths around the group and step boundaries, blocks of any size (block boundaries inside a lane's four records, empty tails), a
first record whose delta is not zero, lists that mix dense groups with groups that hold a longer record, doc ids near 2^32, and the
two lists of a query in one launch."""
import numpy as np
import pytest

import oracle as O
from redisearch_amd import search as S
from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu


@pytest.fixture
def lib():
    lb = V.load()
    lb.RSGPU_SetTuning(b"cache_decoded", 0)   # (lists uploaded in decode-per-query mode carry sync points)
    yield lb
    lb.RSGPU_SetTuning(b"cache_decoded", 1)
    lb.RSGPU_SetTuning(b"decode_dense", 1)
    lb.RSGPU_SetTuning(b"decode_pair", 1)


def decodes_equal(lib, fl, want, what):
    """the first decode leaves the sync points (a block per lane), the later ones take the dense kernel / the sync points"""
    for dense in (1, 0):
        lib.RSGPU_SetTuning(b"decode_dense", dense)
        p = S.Postings.from_flat(fl)
        try:
            for rep in range(3):
                got = p.decode()
                for name, g, w in zip(("ids", "freqs", "masks"), got, want):
                    if name in what:
                        assert np.array_equal(g, w), (dense, rep, name)
        finally:
            p.free()


def small_list(codec, n, seed, start=1):
    rng = np.random.default_rng(seed)
    docs = start + np.cumsum(rng.integers(1, 40, n)).astype(np.uint64)
    ii = O.InvertedIndex(codec)
    if codec == O.C_FREQS_ONLY:
        ii.add_many(docs, rng.integers(1, 200, n).astype(np.uint32))
    else:
        fr, mk = rng.integers(1, 200, n), rng.integers(1, 256, n)
        for d, f, m in zip(docs.tolist(), fr.tolist(), mk.tolist()):
            ii.add(d, f, m)
    return ii


@pytest.mark.parametrize("codec,what", [(O.C_FREQS_ONLY, ("ids", "freqs")), (O.C_FIELDS_ONLY, ("ids", "masks")),
                                        (O.C_FREQS_FIELDS, ("ids", "freqs", "masks"))])
@pytest.mark.parametrize("n", [1, 3, 4, 5, 100, 101, 255, 256, 257, 1_599, 1_600, 1_601, 3_333, 20_011, 160_007])
def test_dense_lists_of_every_layout(lib, codec, what, n):
    ii = small_list(codec, n, 100 * codec + n)
    fl = ii.flatten()
    assert int(fl["offset"][-1]) == n * (4 if codec == O.C_FREQS_FIELDS else 3)  # (every record minimal: the dense path it is)
    decodes_equal(lib, fl, ii.decode_all(), what)


def qint2(delta, value):
    out, hdr = bytearray([0]), 0
    for i, v in enumerate((int(delta), int(value))):
        ln = 1 if v < (1 << 8) else 2 if v < (1 << 16) else 3 if v < (1 << 24) else 4
        hdr |= (ln - 1) << (2 * i)
        out += v.to_bytes(ln, "little")
    out[0] = hdr
    return bytes(out)


def hand_built(sizes, rng, first_delta=0, long_every=0, doc=10):
    """blocks of the given sizes; first_delta: the delta the first record of a block carries (the reference writes 0; the chain
    parsers add whatever is there to first_doc_id, and so must this one); long_every: every so-manieth record has a two-byte delta"""
    first, last, nent, off, data = [], [], [], [0], bytearray()
    docs_all, freqs_all, k = [], [], 0
    for n in sizes:
        deltas = rng.integers(1, 30, n).astype(np.int64)
        if long_every:
            idx = np.arange(k, k + n)
            deltas[(idx % long_every) == long_every - 1] = 300
        k += n
        freqs = rng.integers(1, 256, n)
        ids = doc + np.cumsum(deltas)
        doc = int(ids[-1])
        first.append(int(ids[0]) - first_delta)       # first_doc_id + the first record's delta = the first doc id
        last.append(doc)
        nent.append(n)
        prev = int(ids[0]) - first_delta
        for d, f in zip(ids.tolist(), freqs.tolist()):
            data += qint2(d - prev, f)
            prev = d
        off.append(len(data))
        docs_all += ids.tolist()
        freqs_all += freqs.tolist()
    fl = dict(codec=O.C_FREQS_ONLY, first=np.array(first, np.uint64), last=np.array(last, np.uint64),
              num_entries=np.array(nent, np.uint32), offset=np.array(off, np.uint64), bytes=np.frombuffer(bytes(data), np.uint8))
    return fl, (np.array(docs_all, np.uint64), np.array(freqs_all, np.uint32), None)


@pytest.mark.parametrize("first_delta", [0, 5])
def test_blocks_of_any_size(lib, first_delta):
    """block boundaries inside a lane's four records, blocks shorter than a lane's share, one-record blocks side by side, more
    blocks than one group, a last group of one block"""
    rng = np.random.default_rng(7 + first_delta)
    sizes = [250, 17, 112, 113, 128, 1, 16, 15, 100, 1000, 3, 2, 1, 1, 1, 7, 99, 101, 64, 63, 65, 4, 4, 4, 5, 300, 31, 33, 100, 100, 100, 100, 9]
    fl, want = hand_built(sizes, rng, first_delta)
    decodes_equal(lib, fl, want, ("ids", "freqs"))


def test_lists_that_mix_dense_groups_with_longer_records(lib):
    """a two-byte delta every 2 500 records: about every other group of sixteen blocks holds one and takes the sync points"""
    rng = np.random.default_rng(3)
    fl, want = hand_built([100] * 403 + [37], rng, long_every=2_500)
    decodes_equal(lib, fl, want, ("ids", "freqs"))
    # and a list whose ids end just below 2^32 (the arithmetic is modulo 2^32 in both forms)
    fl, want = hand_built([100] * 40, rng, doc=(1 << 32) - 100_000)
    assert int(want[0][-1]) < (1 << 32)
    decodes_equal(lib, fl, want, ("ids", "freqs"))


@pytest.mark.parametrize("codecs", [(O.C_FREQS_ONLY, O.C_FREQS_ONLY), (O.C_FREQS_ONLY, O.C_FIELDS_ONLY), (O.C_FREQS_FIELDS, O.C_FREQS_ONLY),
                                    (O.C_FREQS_ONLY, O.C_FULL)])
def test_the_two_lists_of_a_query_in_one_launch(lib, codecs):
    """two dense lists share the dense kernel's launch; a list with inline offsets next to a dense one keeps the pair kernel.  Same
    intersection as the oracle's, query after query, with the knobs on and off."""
    rng = np.random.default_rng(sum(codecs) + 17)
    lists = []
    for j, codec in enumerate(codecs):
        docs = np.unique(rng.integers(1, 400_000, 90_000 + 45_000 * j)).astype(np.uint64)
        ii = O.InvertedIndex(codec)
        if codec == O.C_FREQS_ONLY:
            ii.add_many(docs, rng.integers(1, 200, docs.size).astype(np.uint32))
        else:
            for d in docs.tolist():
                ii.add(d, int(rng.integers(1, 200)), int(rng.integers(1, 256)), bytes([3]) if codec == O.C_FULL else b"")
        lists.append(ii)
    oi, of, _ = O.intersect(lists)
    g = [S.Postings.from_flat(l.flatten()) for l in lists]
    try:
        for dense, pair in ((1, 1), (0, 1), (1, 0), (1, 1)):
            lib.RSGPU_SetTuning(b"decode_dense", dense)
            lib.RSGPU_SetTuning(b"decode_pair", pair)
            for rep in range(3):
                h = S.intersect(g)
                gi, gf = h.read()
                assert gi.tolist() == oi.tolist(), (dense, pair, rep)
                assert gf.tolist() == of.tolist(), (dense, pair, rep)
                h.free()
    finally:
        for p in g:
            p.free()
