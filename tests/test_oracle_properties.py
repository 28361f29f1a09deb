"""Size-independent properties of the oracle's integer half, checked against plain Python sets / sorts
(an independent restatement): set algebra of intersect / union / not, codec round trips, fusion invariants."""
import numpy as np
import pytest

import oracle as O


def mk(rng, codec, n, hi):
    docs = np.unique(rng.integers(1, hi, n)).astype(np.uint64)
    freqs = rng.integers(1, 1000, docs.size).astype(np.uint32)
    ii = O.InvertedIndex(codec)
    ii.add_many(docs, freqs)
    return ii, docs, freqs


@pytest.mark.parametrize("seed", range(6))
def test_set_algebra_against_python_sets(seed):
    rng = np.random.default_rng(seed)
    lists = [mk(rng, O.C_FREQS_ONLY, int(rng.integers(1, 4000)), 6000) for _ in range(int(rng.integers(1, 6)))]
    sets = [set(d.tolist()) for _, d, _ in lists]
    inter, fr_i, _ = O.intersect([l[0] for l in lists])
    assert inter.tolist() == sorted(set.intersection(*sets))
    uni, fr_u, _ = O.union_lists([l[0] for l in lists])
    assert uni.tolist() == sorted(set.union(*sets))
    for t, (_, docs, freqs) in enumerate(lists):
        lut = dict(zip(docs.tolist(), freqs.tolist()))
        assert fr_i[t].tolist() == [lut[d] for d in inter.tolist()]
        assert fr_u[t].tolist() == [lut.get(d, 0) for d in uni.tolist()]
    child = lists[0][0]
    assert O.not_list(child, 6100).tolist() == sorted(set(range(1, 6101)) - sets[0])
    if len(lists) > 1:
        assert O.not_list(child, 5000, universe=lists[1][0]).tolist() == sorted(d for d in sets[1] - sets[0] if d <= 5000)


@pytest.mark.parametrize("codec", range(9))
def test_every_codec_round_trips_doc_ids(codec):
    rng = np.random.default_rng(codec)
    docs = np.unique(np.concatenate([rng.integers(1, 2 ** 31, 3000), rng.integers(1, 5000, 3000)])).astype(np.uint64)
    ii = O.InvertedIndex(codec)
    for d in docs.tolist():
        ii.add(d, int(d % 977) + 1, int(d % 65521) + 1, b"\x01\x05" if d % 3 else b"")
    ids, freqs, masks = ii.decode_all()
    assert ids.tolist() == docs.tolist()
    r = ii.reader()
    probe = docs[:: max(1, len(docs) // 50)]
    for d in probe.tolist():
        assert r.seek(d) and r.next is not None


def test_varint_and_qint_round_trip():
    rng = np.random.default_rng(1)
    for v in [0, 1, 127, 128, 16383, 16384, 2 ** 21, 2 ** 28 + 5, 2 ** 32 - 1] + rng.integers(0, 2 ** 32, 200).tolist():
        enc = O.varint_encode(int(v))
        assert O.varint_decode(enc)[0] == v
    for _ in range(200):
        n = int(rng.integers(2, 5))
        vals = [int(rng.integers(0, 2 ** int(rng.integers(1, 33)))) for _ in range(n)]
        enc = O.qint_encode(vals)
        assert list(O.qint_decode(enc, n)[0]) == vals


@pytest.mark.parametrize("scoring", [O.RRF, O.LINEAR])
def test_fusion_invariants(scoring):
    rng = np.random.default_rng(7 + scoring)
    for _ in range(20):
        na, nb, w = int(rng.integers(0, 60)), int(rng.integers(0, 60)), int(rng.integers(1, 70))
        a = rng.permutation(200)[:na] + 1
        b = rng.permutation(200)[:nb] + 1
        sa, sb = np.sort(rng.uniform(0, 5, na))[::-1], np.sort(rng.uniform(0, 2, nb))
        ids, sc = O.hybrid_fuse(scoring, a, sa, b, sb, w, constant=60.0, weights=(0.4, 0.6), metric=O.L2)
        assert len(set(ids.tolist())) == len(ids) == len(set(a[:w].tolist()) | set(b[:w].tolist()))
        order = sorted(range(len(ids)), key=lambda i: (-sc[i], ids[i]))
        assert order == list(range(len(ids)))
        ra = {d: i + 1 for i, d in enumerate(a[:w].tolist())}
        rb = {d: i + 1 for i, d in enumerate(b[:w].tolist())}
        for d, s in zip(ids.tolist(), sc.tolist()):
            if scoring == O.RRF:
                want = (1.0 / (60.0 + ra[d]) if d in ra else 0.0) + (1.0 / (60.0 + rb[d]) if d in rb else 0.0)
            else:
                want = (0.4 * sa[ra[d] - 1] if d in ra else 0.0) + (0.6 * (1.0 / (1.0 + sb[rb[d] - 1])) if d in rb else 0.0)
            assert abs(s - want) <= 1e-15 * max(1.0, abs(want))
