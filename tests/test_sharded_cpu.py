"""N>1 path on CPU: world_size-2 and -4 gloo runs of the row-sharded top-k exchange (redisearch_amd/sharded.py).
Each rank holds a contiguous label range; the merged result must equal the single-index answer."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, rows, dim, k, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    from redisearch_amd.sharded import ShardedTopK
    data = np.random.default_rng(47).uniform(-1, 1, (rows * world, dim)).astype(np.float32)
    shard = O.FlatIndex(O.F32, dim, O.COSINE)
    shard.add_bulk(data[rank * rows:(rank + 1) * rows], first_label=rank * rows + 1)

    def local_topk(q, kk):                       # the CPU stand-in of RSGPU_FlatIndex_TopKDevice
        ids, sc = shard.topk(q, kk)
        s = torch.full((kk,), float("inf"), dtype=torch.float32)
        l = torch.full((kk,), -1, dtype=torch.int64)
        s[: len(sc)] = torch.from_numpy(sc.astype(np.float32))
        l[: len(ids)] = torch.from_numpy(ids.astype(np.int64))
        return s, l

    sh = ShardedTopK(local_topk, k, torch.device("cpu"))
    res = []
    for q in np.random.default_rng(48).uniform(-1, 1, (6, dim)).astype(np.float32):
        labels, scores = sh.query(q)
        res.append((labels.tolist(), scores.tolist()))
    if rank == 0:
        np.save(os.path.join(out_dir, "res.npy"), np.array([r[0] for r in res], dtype=np.uint64))
        np.save(os.path.join(out_dir, "sc.npy"), np.array([r[1] for r in res], dtype=np.float64))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,rows,k", [(2, 500, 10), (2, 7, 10), (4, 300, 10), (4, 3, 10)])
def test_sharded_topk_matches_single_index(tmp_path, world, rows, k):
    # (rows < k: shards pad with +inf / UINT64_MAX and the merge returns fewer than world*k real candidates)
    import oracle as O
    dim = 24
    mp.spawn(_worker, args=(world, _free_port(), rows, dim, k, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "res.npy", allow_pickle=True)
    sc = np.load(tmp_path / "sc.npy", allow_pickle=True)
    data = np.random.default_rng(47).uniform(-1, 1, (rows * world, dim)).astype(np.float32)
    full = O.FlatIndex(O.F32, dim, O.COSINE)
    full.add_bulk(data)
    for i, q in enumerate(np.random.default_rng(48).uniform(-1, 1, (6, dim)).astype(np.float32)):
        ids, s = full.topk(q, k)
        assert list(got[i]) == ids.tolist()
        assert np.allclose(sc[i], s.astype(np.float32), atol=1e-6)


def _uid_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from redisearch_amd.sharded import broadcast_unique_id
    mine = bytes((7 * i + 3) % 256 for i in range(128)) if rank == 0 else b""
    got = broadcast_unique_id(mine, torch.device("cpu"))
    open(os.path.join(out_dir, "uid%d.bin" % rank), "wb").write(got)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_unique_id_reaches_every_rank(tmp_path, world):
    """the one launcher-side step of the RCCL bootstrap (RSGPU_ShardComm_Init takes the id every rank must share): rank 0's
    128 bytes arrive unchanged on every rank -- here over gloo; bench.py uses the same function over the nccl backend"""
    mp.spawn(_uid_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = bytes((7 * i + 3) % 256 for i in range(128))
    for r in range(world):
        assert open(tmp_path / ("uid%d.bin" % r), "rb").read() == want


def test_merge_topk_ties_and_padding():
    from redisearch_amd.sharded import merge_topk, merge_topk_numpy
    rng = np.random.default_rng(3)
    for _ in range(50):                       # the C merge (RSGPU_MergeTopKHost) against the numpy ordering
        m = int(rng.integers(0, 60))
        sc = rng.integers(0, 6, m).astype(np.float32)
        lb = rng.integers(1, 40, m).astype(np.uint64)
        lb[rng.random(m) < 0.2] = np.uint64(0xFFFFFFFFFFFFFFFF)
        kk = int(rng.integers(0, 25))
        a, b = merge_topk(sc, lb, kk), merge_topk_numpy(sc, lb, kk)
        assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()
    s = np.array([0.5, 0.1, np.inf, 0.1, 0.3, np.inf], dtype=np.float32)
    l = np.array([9, 7, -1, 3, 8, -1], dtype=np.int64).view(np.uint64)
    labels, scores = merge_topk(s, l, 3)
    assert labels.tolist() == [3, 7, 8] and np.allclose(scores, [0.1, 0.1, 0.3])
    labels, _ = merge_topk(s, l, 10)
    assert labels.tolist() == [3, 7, 8, 9]


def test_merge_topk_is_a_total_order_with_nan_scores():
    """Scores can be NaN (a NaN element stored in an index): the merge's comparator must stay a strict weak order -- NaN after
    every number, ties by label -- or std::partial_sort may leave its range.  Many random draws with a third of the scores
    NaN, checked against the same order in numpy (lexsort puts NaN last as well)."""
    from redisearch_amd.sharded import merge_topk
    rng = np.random.default_rng(9)
    for _ in range(300):
        m = int(rng.integers(1, 400))
        sc = rng.integers(0, 5, m).astype(np.float32)
        sc[rng.random(m) < 0.33] = np.nan
        sc[rng.random(m) < 0.05] = np.inf
        lb = rng.permutation(m).astype(np.uint64) + 1
        kk = int(rng.integers(1, m + 3))
        labels, scores = merge_topk(sc, lb, kk)
        order = np.lexsort((lb, sc))[:kk]                 # (NaN sorts last in numpy too)
        assert labels.tolist() == lb[order].tolist()
        assert np.array_equal(scores, sc[order].astype(np.float64), equal_nan=True)
