"""GPU: the hybrid coalescer (round 6; hybrid_entry.hpp, hybrid_kernels.hip hybrid_tile_batch_kernel / hybrid_reduce_batch_kernel).

RediSearch runs queries from a pool of worker threads (src/util/workers.c:58,104); the hybrid iterator's loop
(src/iterators/hybrid_reader.c:309-327) runs on each.  Two-launch hybrid queries of concurrent callers share tile grids: a caller
that arrives while `hybrid_coalesce_depth` grids are in flight queues its description and is launched -- with up to six others,
ONE grid + ONE reduce launch -- by the caller whose grid finishes next.  Every answer must equal the answer the same query gives
alone, bit for bit: queries of different shapes (1-4 lists, different driving lists and tile counts, either branch off, every
scorer family, different top_n / k) side by side in one grid, both block orders, indexes of different types (two kernel
instantiations: those never share a grid), deadlines that pass while a query is queued, and the knob off."""
import threading

import numpy as np
import pytest

import oracle as O
from redisearch_amd import search as S
from redisearch_amd import vecsim as V
from tests.test_gpu_hybrid_general import flat_corpus, table_for

pytestmark = pytest.mark.gpu


def knob(name, value):
    V.load().RSGPU_SetTuning(name.encode(), int(value))


@pytest.fixture(autouse=True)
def defaults():
    yield
    knob("hybrid_coalesce", 1)
    knob("hybrid_coalesce_depth", 2)
    knob("hybrid_coalesce_interleave", 0)


@pytest.fixture(scope="module")
def world():
    n_docs = 2_000_000
    lists_o, rng = flat_corpus(n_docs, (0.30, 0.42, 0.2, 0.36, 0.05), 606)
    g = [S.Postings.from_flat(l.flatten()) for l in lists_o]
    table = table_for(rng, n_docs)
    idf = [S.calculate_idf(n_docs, l.unique_docs) for l in lists_o]
    bidf = [S.calculate_idf_bm25(n_docs, l.unique_docs) for l in lists_o]
    idx = V.VecSimIndex(V.VecSimType_FLOAT32, 64, V.VecSimMetric_L2)
    idx.add_philox_rows(61, 0, 400_000, 1)
    idx16 = V.VecSimIndex(V.VecSimType_FLOAT16, 96, V.VecSimMetric_IP)
    idx16.add_philox_rows(62, 0, 300_000, 1)
    qs = O.philox_rows(61, 1 << 40, 16, 64)
    qs16 = O.philox_rows(62, 1 << 40, 16, 96).astype(np.float16)
    yield dict(n_docs=n_docs, g=g, table=table, idf=idf, bidf=bidf, idx=idx, idx16=idx16, qs=qs, qs16=qs16)
    idx.free()
    idx16.free()


def shapes(w, with_f16=False):
    """one thread's own query objects: [(name, HybridQuery)] -- all of them two-launch shapes"""
    g, idf, bidf = w["g"], w["idf"], w["bidf"]

    def hq(ix, scorer="BM25STD", top_n=10, k=10, qi=0, index="f32"):
        kw = dict(table=w["table"] if top_n else None, scorer=scorer if top_n else None, idf=[idf[i] for i in ix],
                  bm25_idf=[bidf[i] for i in ix], weight=[1.0 + 0.25 * i for i in ix], num_docs=w["n_docs"], avg_doc_len=200.0, top_n=top_n,
                  root_weight=1.25)
        if k:
            kw.update(index=w["idx"] if index == "f32" else w["idx16"], q=(w["qs"] if index == "f32" else w["qs16"])[qi], k=k)
        return S.HybridQuery([g[i] for i in ix], **kw)
    out = [("pair", hq([0, 1])),
           ("pair_b", hq([2, 3], "TFIDF", 7, 12, 1)),
           ("triple", hq([1, 2, 3], "DISMAX", 20, 5, 2)),
           ("four", hq([0, 1, 2, 3], "BM25STD.TANH", 10, 10, 3)),
           ("one_list", hq([4], "DOCSCORE", 10, 10, 4)),
           ("short_driver", hq([4, 1], "BM25", 5, 30, 5)),
           ("score_only", hq([0, 3], "BM25STD", 16, 0)),
           ("knn_only", hq([1, 2], None, 0, 25, 6)),
           ("norm", hq([0, 2], "BM25STD.NORM", 9, 9, 7)),
           ("wide_k", hq([3, 1], "TFIDF.DOCNORM", 50, 64, 8))]
    if with_f16:
        out += [("f16", hq([0, 1], "BM25STD", 10, 10, 1, "f16")), ("f16_b", hq([2, 1, 3], "TFIDF", 10, 20, 2, "f16"))]
    return out


def same(a, b):
    return (a["n_hits"] == b["n_hits"] and a["top"][0].tolist() == b["top"][0].tolist() and a["top"][1].tolist() == b["top"][1].tolist()
            and a["knn"][0].tolist() == b["knn"][0].tolist() and a["knn"][1].tolist() == b["knn"][1].tolist())


def serial_answers(w, with_f16=False):
    knob("hybrid_coalesce", 0)
    out = {}
    for name, q in shapes(w, with_f16):
        assert q.run()
        assert S.hybrid_path() == 1, name
        out[name] = q.results()
        assert out[name]["n_hits"] > 0, name
    knob("hybrid_coalesce", 1)
    return out


def hammer(w, n_threads, rounds, with_f16=False, each=None):
    per_thread = [shapes(w, with_f16) for _ in range(n_threads)]
    results = [[] for _ in range(n_threads)]
    errors = []
    gate = threading.Barrier(n_threads)

    def work(t):
        try:
            gate.wait()
            for r in range(rounds):
                for j in range(len(per_thread[t])):
                    name, q = per_thread[t][(j * 3 + t + r) % len(per_thread[t])]
                    ok = q.run()
                    results[t].append((name, ok, S.hybrid_path(), q.results()))
                    if each:
                        each(t, name, q)
        except Exception as e:                                                   # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    return results


@pytest.mark.parametrize("depth,interleave", [(1, 1), (1, 0), (2, 1), (3, 0)])
def test_queries_of_concurrent_callers_share_grids_and_keep_their_answers(world, depth, interleave):
    serial = serial_answers(world)
    knob("hybrid_coalesce_depth", depth)
    knob("hybrid_coalesce_interleave", interleave)
    S.hybrid_coalesce_stats(reset=True)
    results = hammer(world, 8, 30)
    st = S.hybrid_coalesce_stats()
    n = 0
    for t, rs in enumerate(results):
        for name, ok, path, r in rs:
            assert ok and path == 1, (t, name, path)
            assert same(r, serial[name]), (t, name)
            n += 1
    assert st["alone"] + st["grid_queries"] == n and st["relaunched"] == 0, st
    if depth == 1:                       # eight callers behind one grid in flight: they queue, and they share
        assert st["queued"] > 0 and st["grids"] > 0 and st["grid_queries"] > st["grids"], st


def test_indexes_of_two_types_never_share_a_grid_and_the_knob_off_means_no_grids(world):
    serial = serial_answers(world, with_f16=True)
    knob("hybrid_coalesce_depth", 1)
    S.hybrid_coalesce_stats(reset=True)
    for t, rs in enumerate(hammer(world, 6, 20, with_f16=True)):
        for name, ok, path, r in rs:
            assert ok and path == 1 and same(r, serial[name]), (t, name)
    assert S.hybrid_coalesce_stats()["relaunched"] == 0
    knob("hybrid_coalesce", 0)
    S.hybrid_coalesce_stats(reset=True)
    for t, rs in enumerate(hammer(world, 4, 5)):
        for name, ok, path, r in rs:
            assert ok and same(r, serial[name]), (t, name)
    assert S.hybrid_coalesce_stats() == dict(alone=0, grids=0, grid_queries=0, queued=0, relaunched=0)


def test_deadlines_that_pass_in_the_queue_leave_it_and_everything_else_goes_on(world):
    """every third query of every thread carries a callback that fires from its second poll on: queued or launched, the query
    answers RSGPU_TIMED_OUT with empty outputs (or, when it was quicker than the callback, its answer); nobody else is disturbed
    and nothing hangs; afterwards every shape answers as it does alone"""
    serial = serial_answers(world)
    knob("hybrid_coalesce_depth", 1)
    per_thread = [shapes(world) for _ in range(8)]
    errors, timed_out = [], [0]
    gate = threading.Barrier(8)

    def work(t):
        try:
            gate.wait()
            for r in range(40):
                name, q = per_thread[t][(r * 3 + t) % len(per_thread[t])]
                polls = [0]

                def fire():
                    polls[0] += 1
                    return polls[0] >= 2
                q.set_timeout(fire if r % 3 == 0 else None)
                ok = q.run()
                res = q.results()
                if ok:
                    assert same(res, serial[name]), (t, name)
                else:
                    timed_out[0] += 1
                    assert res["n_hits"] == 0 and len(res["top"][0]) == 0 and len(res["knn"][0]) == 0
                assert polls[0] >= 1 or r % 3
        except Exception as e:                                                   # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
        assert not th.is_alive(), "a caller hangs in the coalescer"
    assert not errors, errors
    for name, q in shapes(world):
        assert q.run() and same(q.results(), serial[name]), name
