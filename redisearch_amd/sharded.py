"""Row-sharded FLAT KNN across the GPUs of one node: one process per GPU, each rank holds a
contiguous label range of the corpus; a query runs on every shard and the per-shard top-k
(fp32 score, u64 label) are exchanged with ONE all-gather of k*(8+8) bytes per rank (RCCL over xGMI on GPUs, gloo in the CPU
tests) and merged -- the collective analogue of the reference coordinator's per-shard top-K -> heap
merge (reference src/module.c:3541-3547).  The payload is k*12 bytes per rank, so the
exchange is latency-bound; nothing is reduced.
"""
import numpy as np

UINT64_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)


def merge_topk(scores, labels, k):
    """k best of the gathered candidates by (score, label) ascending; padding slots carry
    label == UINT64_MAX.  The same order RSGPU_MergeTopK implements in C."""
    scores = np.asarray(scores, dtype=np.float64).ravel()
    labels = np.asarray(labels).ravel().view(np.uint64) if np.asarray(labels).dtype != np.uint64 else np.asarray(labels).ravel()
    keep = labels != UINT64_MAX
    scores, labels = scores[keep], labels[keep]
    order = np.lexsort((labels, scores))[:k]
    return labels[order], scores[order]


class ShardedTopK:
    """local_topk(q, k) -> (scores tensor[k] float32, labels tensor[k] int64) on `device`, padded with
    +inf / -1 (== UINT64_MAX)."""

    def __init__(self, local_topk, k, device, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.local_topk, self.k, self.group = local_topk, k, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # one collective per query: labels and the fp32 score bits travel in the same int64 buffer
        self.pack = torch.empty(2 * k, dtype=torch.int64, device=device)
        self.all_p = torch.empty(2 * k * self.world, dtype=torch.int64, device=device)

    def query(self, q):
        s, l = self.local_topk(q, self.k)
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
        return merge_topk(scores, labels, k)
