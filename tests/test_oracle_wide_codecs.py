"""Pins the oracle's *Wide codecs (128-bit field masks as varints) on the reference's byte vectors:
inverted_index/tests/integration/codec/full.rs:98-165, freqs_fields.rs:100-140, fields_only.rs:86-120,
fields_offsets.rs:76-130."""
import numpy as np
import pytest

import oracle as O

U32, U128 = 2 ** 32 - 1, 2 ** 128 - 1


def enc(codec, delta, freq, mask, offsets=b""):
    """bytes of ONE record with the given delta: a first record at doc 1000 (delta 0 in a fresh block), then the probe"""
    ii = O.InvertedIndex(codec)
    ii.add_wide(1000, 1, 1, b"")
    f0 = ii.flatten()["bytes"].size
    ii.add_wide(1000 + delta, freq, mask, offsets) if delta else None
    if not delta:   # delta 0 only occurs as the first record of a block
        jj = O.InvertedIndex(codec)
        jj.add_wide(77, freq, mask, offsets)
        return jj.flatten()["bytes"].tolist()
    return ii.flatten()["bytes"].tolist()[f0:]


@pytest.mark.parametrize("delta,freq,mask,offs,expected", [
    (0, 1, 1, [1, 2, 3], [0, 0, 1, 3, 1, 1, 2, 3]),
    (10, 5, U32, [1, 2, 3, 4], [0, 10, 5, 4, 142, 254, 254, 254, 127, 1, 2, 3, 4]),
    (256, 1, 1, [1, 2, 3], [1, 0, 1, 1, 3, 1, 1, 2, 3]),
    (65536, 1, 1, [1, 2, 3], [2, 0, 0, 1, 1, 3, 1, 1, 2, 3]),
    (65535, 1, 1, [1, 2, 3], [1, 255, 255, 1, 3, 1, 1, 2, 3]),
    (U32, 1, 1, [1, 2, 3], [3, 255, 255, 255, 255, 1, 3, 1, 1, 2, 3]),
    (U32, U32, U128, [1] * 100, [15, 255, 255, 255, 255, 255, 255, 255, 255, 100, 130] + [254] * 17 + [127] + [1] * 100),
])
def test_full_wide_vectors(delta, freq, mask, offs, expected):
    assert enc(O.C_FULL_WIDE, delta, freq, mask, bytes(offs)) == expected


@pytest.mark.parametrize("delta,freq,mask,expected", [
    (0, 1, 1, [0, 0, 1, 1]), (10, 5, U32, [0, 10, 5, 142, 254, 254, 254, 127]), (256, 1, 1, [1, 0, 1, 1, 1]),
    (65536, 1, 1, [2, 0, 0, 1, 1, 1]), (65535, 1, 1, [1, 255, 255, 1, 1]), (U32, 1, 1, [3, 255, 255, 255, 255, 1, 1]),
])
def test_freqs_fields_wide_vectors(delta, freq, mask, expected):
    assert enc(O.C_FREQS_FIELDS_WIDE, delta, freq, mask) == expected


@pytest.mark.parametrize("delta,mask,expected", [
    (0, 1, [0, 1]), (10, U32, [10, 142, 254, 254, 254, 127]), (256, 1, [129, 0, 1]), (65536, 1, [130, 255, 0, 1]),
    (65535, 1, [130, 254, 127, 1]), (U32, 1, [142, 254, 254, 254, 127, 1]),
    (U32, U32, [142, 254, 254, 254, 127, 142, 254, 254, 254, 127]),
])
def test_fields_only_wide_vectors(delta, mask, expected):
    assert enc(O.C_FIELDS_ONLY_WIDE, delta, 1, mask) == expected


@pytest.mark.parametrize("delta,mask,offs,expected", [
    (0, 1, [1, 2, 3], [0, 0, 3, 1, 1, 2, 3]), (10, U32, [1, 2, 3, 4], [0, 10, 4, 142, 254, 254, 254, 127, 1, 2, 3, 4]),
    (256, 1, [1, 2, 3], [1, 0, 1, 3, 1, 1, 2, 3]), (65536, 1, [1, 2, 3], [2, 0, 0, 1, 3, 1, 1, 2, 3]),
    (65535, 1, [1, 2, 3], [1, 255, 255, 3, 1, 1, 2, 3]), (U32, 1, [1, 2, 3], [3, 255, 255, 255, 255, 3, 1, 1, 2, 3]),
])
def test_fields_offsets_wide_vectors(delta, mask, offs, expected):
    assert enc(O.C_FIELDS_OFFSETS_WIDE, delta, 1, mask, bytes(offs)) == expected


@pytest.mark.parametrize("codec", O.WIDE_CODECS)
def test_wide_round_trip(codec):
    rng = np.random.default_rng(codec)
    ii = O.InvertedIndex(codec)
    docs = np.unique(rng.integers(1, 10 ** 7, 3000))
    masks = [int(rng.integers(1, 2 ** 62)) << int(rng.integers(0, 66)) | 1 for _ in docs]
    for d, m in zip(docs.tolist(), masks):
        ii.add_wide(d, int(rng.integers(1, 300)), m, bytes(rng.integers(1, 100, int(rng.integers(0, 4))).astype(np.uint8)))
    ids, fr, mk = ii.decode_all()
    assert ids.tolist() == docs.tolist()
    assert ii.decode_masks128() == [m & (2 ** 128 - 1) for m in masks]
    assert mk.tolist() == [m & 0xFFFFFFFF for m in masks]
