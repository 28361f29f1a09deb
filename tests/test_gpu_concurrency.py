"""Concurrent readers: RediSearch runs queries from a pool of worker threads under the spec read lock
(SURVEY.md 8b "Threading"); every thread leases its own stream + workspace.  Results must equal the
single-threaded answers, for all query kinds at once."""
import threading

import numpy as np
import pytest
import torch

from redisearch_amd import vecsim as V

pytestmark = pytest.mark.gpu


def test_concurrent_queries_match_serial_answers():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(21)
    n, dim = 400_000, 64
    x = torch.rand((n, dim), device=dev, generator=gen) * 2 - 1
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, dim, V.VecSimMetric_L2)
    torch.cuda.synchronize()
    idx.add_device_rows(x.data_ptr(), n, 1)
    qs = np.random.default_rng(22).uniform(-1, 1, (64, dim)).astype(np.float32)
    want_topk = [idx.topk_query(q, 10).results() for q in qs]
    want_range = [idx.range_query(q, float(want_topk[i][1][4]), order=V.BY_ID).results() for i, q in enumerate(qs)]
    errors = []

    def worker(t):
        try:
            for rep in range(3):
                for i in range(t, len(qs), 4):
                    ids, sc = idx.topk_query(qs[i], 10).results()
                    assert ids.tolist() == want_topk[i][0].tolist() and sc.tolist() == want_topk[i][1].tolist()
                    rid, _ = idx.range_query(qs[i], float(want_topk[i][1][4]), order=V.BY_ID).results()
                    assert rid.tolist() == want_range[i][0].tolist()
                    it = idx.batch_iterator(qs[i])
                    bi, _ = it.next(10, V.BY_SCORE).results()
                    assert bi.tolist() == want_topk[i][0].tolist()
                    nq = idx.normalized_query(qs[i])
                    assert idx.get_distance_from_unsafe(int(ids[0]), nq) == sc[0]
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
