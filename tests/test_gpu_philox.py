"""GPU: the keyed synthetic corpus (corpus_kernels.hip, RSGPU_FlatIndex_AddPhiloxRows) against its CPU twin
(oracle.philox_rows): every element type, padded and unpadded strides, bit for bit -- so that any row of a corpus that
was generated in HBM can be regenerated on a host."""
import numpy as np
import pytest

import oracle as O
from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu
TYPES = [(V.VecSimType_FLOAT32, O.F32), (V.VecSimType_FLOAT64, O.F64), (V.VecSimType_FLOAT16, O.F16),
         (V.VecSimType_BFLOAT16, O.BF16), (V.VecSimType_INT8, O.I8), (V.VecSimType_UINT8, O.U8)]


@pytest.mark.parametrize("vt,ot", TYPES)
@pytest.mark.parametrize("dim", [1, 5, 7, 16, 30, 64, 129, 768])
def test_device_rows_equal_host_rows(vt, ot, dim):
    g = V.VecSimIndex(vt, dim, V.VecSimMetric_L2)
    assert g.add_philox_rows(99, 1000, 300, 1) == 300
    assert g.add_philox_rows(99, 2 ** 32 - 100, 200, 301) == 200       # the row counter crosses 2^32
    got = g.read_rows(0, 500)
    exp = np.concatenate([O.philox_rows(99, 1000, 300, dim, ot), O.philox_rows(99, 2 ** 32 - 100, 200, dim, ot)])
    assert got.dtype == exp.dtype or got.view(np.uint16).dtype == exp.view(np.uint16).dtype
    assert np.array_equal(got.view(np.uint8), exp.view(np.uint8))


def test_queries_over_a_generated_corpus_match_the_oracle_on_regenerated_rows():
    n, dim, seed = 50_000, 96, 5
    for metric, om in ((V.VecSimMetric_Cosine, O.COSINE), (V.VecSimMetric_L2, O.L2)):
        g = V.VecSimIndex(V.VecSimType_FLOAT32, dim, metric)
        g.add_philox_rows(seed, 0, n, 1)
        o = O.FlatIndex(O.F32, dim, om)
        o.add_bulk(O.philox_rows(seed, 0, n, dim), 1)
        for qi in (n + 7, n + 8):
            q = O.philox_rows(seed, qi, 1, dim)[0]
            gi, gs = g.topk_query(q, 10).results()
            oi, os_ = o.topk(q, 10)
            assert gi.tolist() == oi.tolist()
            assert np.allclose(gs, os_, rtol=1e-5, atol=1e-4)
