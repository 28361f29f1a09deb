"""Shared helpers of the parity tests: the HIP path (through the C ABI) against the CPU oracle."""
import numpy as np

import oracle as O
from redisearch_amd import vecsim as V

TYPE_TO_ORACLE = {V.VecSimType_FLOAT32: O.F32, V.VecSimType_FLOAT16: O.F16, V.VecSimType_BFLOAT16: O.BF16,
                  V.VecSimType_FLOAT64: O.F64, V.VecSimType_INT8: O.I8, V.VecSimType_UINT8: O.U8}
# fp32 parity tolerance stated by BASELINE.json's north_star: distances within 1e-4 (absolute, for the
# unit-scale distances of cosine / normalised IP; relative for large L2 magnitudes)
ATOL, RTOL = 1e-4, 1e-5


def close(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= ATOL + RTOL * np.abs(b))


def build_pair(vtype, dim, metric, data, labels=None, multi=False):
    """Same rows into a GPU index (C ABI) and the oracle."""
    g = V.VecSimIndex(vtype, dim, metric, multi=multi)
    o = O.FlatIndex(TYPE_TO_ORACLE[vtype], dim, metric, multi=multi)
    labels = labels if labels is not None else np.arange(1, len(data) + 1)
    for row, lab in zip(data, labels):
        rg = g.add_vector(row, int(lab))
        ro = o.add(row, int(lab))
        assert rg == ro
    return g, o


def quantize(data, vtype):
    """Round the float data to what the index type can hold, so both sides see identical inputs."""
    if vtype == V.VecSimType_FLOAT16:
        return np.asarray(data, dtype=np.float16).astype(np.float32)
    if vtype == V.VecSimType_BFLOAT16:
        u = V.to_blob(data, vtype).astype(np.uint32) << 16
        return u.view(np.float32).reshape(np.shape(data))
    if vtype == V.VecSimType_FLOAT64:
        return np.asarray(data, dtype=np.float64)
    if vtype == V.VecSimType_INT8:
        return np.clip(np.rint(np.asarray(data) * 127), -128, 127).astype(np.int8)
    if vtype == V.VecSimType_UINT8:
        return np.clip(np.rint(np.abs(np.asarray(data)) * 255), 0, 255).astype(np.uint8)
    return np.asarray(data, dtype=np.float32)


def assert_topk_parity(g, o, q, k, order=V.BY_SCORE):
    """ids identical (near-ties inside the fp32 tolerance may swap at rank k), distances within tol."""
    rep = g.topk_query(q, k, order=order)
    assert rep.code == V.VecSim_QueryReply_OK
    gi, gs = rep.results()
    oi, os_ = o.topk(q, k, order=O.BY_ID if order == V.BY_ID else O.BY_SCORE)
    assert len(gi) == len(oi)
    if gi.tolist() != oi.tolist():
        # only permissible difference: members/ordering that differ by less than the tolerance
        gd = dict(zip(gi.tolist(), gs.tolist()))
        od = dict(zip(oi.tolist(), os_.tolist()))
        kth = max(os_) if len(os_) else 0.0
        nq = o.normalized_query(q)
        for i in set(gd) ^ set(od):
            d = o.distance_from(int(i), nq)
            assert abs(d - kth) <= ATOL + RTOL * abs(kth), "id %d is not a near-tie of rank k" % i
        if order == V.BY_SCORE:
            assert close(np.sort(gs), np.sort(os_))
    else:
        assert close(gs, os_)
    if order == V.BY_ID:
        assert gi.tolist() == sorted(gi.tolist())
    else:
        assert np.all(np.diff(gs) >= 0)
    return gi, gs
