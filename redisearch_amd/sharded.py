"""Row-sharded FLAT KNN across the GPUs of one node, ONE PROCESS PER GPU (how torch.distributed.run launches bench.py):
each rank holds a contiguous label range of the corpus; a query runs on every shard and the per-shard top-k
(fp32 score, u64 label) are exchanged with ONE all-gather of k*(8+8) bytes per rank (RCCL over xGMI on GPUs, gloo in the
CPU tests) and merged by (score, label) in C (RSGPU_MergeTopKHost) -- the collective analogue of the reference
coordinator's per-shard top-K -> heap merge (reference src/module.c:3541-3547).  The payload is k*16 bytes
per rank, so the exchange is latency-bound; nothing is reduced.

(Several GPUs driven by ONE process -- the shape of a Redis module -- need no collective at all: RSGPU_ShardedIndex_* in
include/rsgpu_ext.h, redisearch_amd/csrc/sharded_index.cpp.)
"""
import ctypes as C

import numpy as np

UINT64_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)


def merge_topk(scores, labels, k):
    """k best of the gathered candidates by (score, label) ascending; padding slots carry label == UINT64_MAX.
    Host code stays C: the merge is the library's RSGPU_MergeTopKHost."""
    from . import vecsim as V
    lib = V.load()
    scores = np.ascontiguousarray(np.asarray(scores, dtype=np.float32).ravel())
    labels = np.asarray(labels).ravel()
    labels = np.ascontiguousarray(labels.view(np.uint64) if labels.dtype != np.uint64 else labels)
    out_s, out_l = np.zeros(max(k, 1), np.float64), np.zeros(max(k, 1), np.uint64)
    m = lib.RSGPU_MergeTopKHost(scores.ctypes.data_as(C.c_void_p), labels.ctypes.data_as(C.c_void_p), scores.size, k,
                                out_s.ctypes.data_as(C.c_void_p), out_l.ctypes.data_as(C.c_void_p))
    if m < 0:
        raise RuntimeError(V.last_error())
    return out_l[:m], out_s[:m]


def merge_topk_numpy(scores, labels, k):
    """The same order in numpy (cross-check of the C merge in tests)."""
    scores = np.asarray(scores, dtype=np.float64).ravel()
    labels = np.asarray(labels).ravel()
    labels = labels.view(np.uint64) if labels.dtype != np.uint64 else labels
    keep = labels != UINT64_MAX
    scores, labels = scores[keep], labels[keep]
    order = np.lexsort((labels, scores))[:k]
    return labels[order], scores[order]


def broadcast_unique_id(uid, device, group=None):
    """rank 0's 128-byte RCCL unique id to every rank of the process group (the one launcher-side step of the bootstrap;
    an MPI program would MPI_Bcast it).  uid: bytes on rank 0, anything elsewhere.  Returns the 128 bytes on every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bytes(uid)
    if dist.get_rank(group) == 0:
        t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device=device)
    else:
        t = torch.zeros(128, dtype=torch.uint8, device=device)
    assert t.numel() == 128
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return bytes(t.cpu().numpy().tobytes())


class ShardComm:
    """The exchange in C (redisearch_amd/csrc/shard_comm.cpp, include/rsgpu_ext.h RSGPU_ShardComm_*): ncclAllGather of the
    per-shard top-k + a merge kernel, one communicator per rank.  This wrapper only bootstraps it -- rank 0's unique id
    reaches the other ranks through ONE torch.distributed broadcast, as an MPI launcher would MPI_Bcast it -- and forwards
    query(): no torch collective, no numpy merge on the query path."""

    def __init__(self, index, k, device, group=None):
        import torch
        import torch.distributed as dist
        from . import vecsim as V
        self.lib, self.V, self.index, self.k = V.load(), V, index, k
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        uid = (C.c_char * 128)()
        if rank == 0 and self.lib.RSGPU_ShardComm_GetUniqueId(uid) != 0:
            raise RuntimeError(V.last_error())
        uid = (C.c_char * 128).from_buffer_copy(broadcast_unique_id(bytes(uid), device, group))
        self.ptr = self.lib.RSGPU_ShardComm_Init(rank, self.world, uid, device.index if device.index is not None else 0)
        if not self.ptr:
            raise RuntimeError(V.last_error())
        self._labels, self._scores = np.zeros(k, np.uint64), np.zeros(k, np.float64)
        self.kind = "RCCL all-gather issued from C (ncclAllGather, shard_comm.cpp) of k*16 B per rank over xGMI + merge kernel on every rank"

    def query(self, q):
        qb = self.index._q(q)
        n = self.lib.RSGPU_ShardComm_TopK(self.ptr, self.index.ptr, qb.ctypes.data_as(C.c_void_p), self.k,
                                          self._labels.ctypes.data_as(C.c_void_p), self._scores.ctypes.data_as(C.c_void_p))
        if n < 0:
            raise RuntimeError(self.V.last_error())
        return self._labels[:n].copy(), self._scores[:n].copy()

    def stats(self, reset=False):
        a = (C.c_uint64 * 2)()
        self.lib.RSGPU_ShardComm_GetStats(self.ptr, a, 1 if reset else 0)
        return int(a[0]), int(a[1])

    @property
    def exchanges(self):
        return self.stats()[0]

    @property
    def exchange_ns(self):
        return self.stats()[1]

    def free(self):
        if getattr(self, "ptr", None):
            self.lib.RSGPU_ShardComm_Free(self.ptr)
            self.ptr = None

    __del__ = free


class ShardedTopK:
    """The same exchange through torch.distributed (what the world-size-2 / 4 gloo tests run on CPU; on GPUs ShardComm above
    is the product path).  `local` is this rank's shard: a redisearch_amd.vecsim.VecSimIndex (GPU) or a callable
    local_topk(q, k) -> (scores tensor[k] float32, labels tensor[k] int64) padded with +inf / -1 (CPU tests)."""

    def __init__(self, local, k, device, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.k, self.group = k, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.kind = "all-gather through torch.distributed + RSGPU_MergeTopKHost"
        self.exchanges, self.exchange_ns = 0, 0   # the collective + D2H + merge of every query (bench.py's `collective`)
        # one collective per query: labels and the fp32 score bits travel in the same int64 buffer
        self.pack = torch.empty(2 * k, dtype=torch.int64, device=device)
        self.all_p = torch.empty(2 * k * self.world, dtype=torch.int64, device=device)
        if callable(local):
            self.local_topk = local
        else:
            self._s = torch.empty(k, device=device, dtype=torch.float32)
            self._l = torch.empty(k, device=device, dtype=torch.int64)

            def local_topk(q, kk):
                local.topk_device(q, kk, self._s.data_ptr(), self._l.data_ptr())
                return self._s, self._l
            self.local_topk = local_topk

    def query(self, q):
        import time
        s, l = self.local_topk(q, self.k)   # (returns when this shard's top-k is complete in device memory)
        t0 = time.perf_counter_ns()
        k = self.k
        self.pack[:k] = l
        self.pack[k:] = s.view(self.torch.int32)
        if not self.dist.is_initialized():               # no process group at all: a plain single-GPU caller
            self.all_p.copy_(self.pack)
        else:
            self.dist.all_gather_into_tensor(self.all_p, self.pack, group=self.group)
        ap = self.all_p.cpu().numpy().reshape(self.world, 2, k)
        labels = np.ascontiguousarray(ap[:, 0, :]).ravel().view(np.uint64)
        scores = np.ascontiguousarray(ap[:, 1, :]).astype(np.int32).ravel().view(np.float32)
        out = merge_topk(scores, labels, k)
        self.exchanges += 1
        self.exchange_ns += time.perf_counter_ns() - t0
        return out
