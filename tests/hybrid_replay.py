"""A faithful replay of the reference's HybridIterator (src/iterators/hybrid_reader.c) -- the host code that drives the
VecSim seam for `(filter)=>[KNN k @v $blob]` -- as TEST infrastructure: the reference keeps this code (SURVEY.md 8 a8 /
a14); the tests use it to drive an index (the CPU oracle, or the GPU engine through the C ABI) exactly the way the
reference does and compare the outcomes.

  NewHybridVectorIterator  :627-703  mode selection (no child / k == 0 -> STANDARD_KNN; explicit policy; else
                                     VecSimIndex_PreferAdHocSearch(index, min(child estimate, index size), k, true))
  prepareResults           :372-443  batches loop, batch size = n_res_left * (index_size / child_estimate) + 1
  alternatingIterate       :140-170  merge-join of a BY_ID batch with the child's sorted ids
  insertResultToHeap_Metric:88-105   K-bounded min-max heap, strict `<` admission against the current worst
  reviewHybridSearchPolicy :346-370  re-estimates the child size after every batch, may switch to ad-hoc BF
  computeDistances_RAM     :289-335  one distance per child id, NaN dropped
  HR_ReadHybridUnsortedSingle :446-468  results are yielded by mmh_pop_min (ascending distance)

The heap is the reference's OWN min-max heap (src/util/minmax_heap.c compiled in place into oracle/_ref/, see
oracle/ref_wrap.c) with the reference's comparator cmpVecSimResByScore (:34-44) whenever that library is available;
otherwise a Python stand-in that is exact whenever no two candidates have equal distances.
"""
import ctypes as C
import math

import numpy as np

import oracle as O

STANDARD_KNN, HYBRID_ADHOC_BF, HYBRID_BATCHES, HYBRID_BATCHES_TO_ADHOC_BF = 1, 2, 3, 4


# ---- heaps ---------------------------------------------------------------------------------------------
class _Res(C.Structure):
    _fields_ = [("doc_id", C.c_uint64), ("score", C.c_double)]


_CMP = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)


def _cmp_by_score(p1, p2, _udata):                 # hybrid_reader.c:34-44, including its asymmetric tie rule
    e1, e2 = C.cast(p1, C.POINTER(_Res)).contents, C.cast(p2, C.POINTER(_Res)).contents
    if e1.score < e2.score:
        return -1
    if e1.score > e2.score:
        return 1
    return 1 if e1.doc_id < e2.doc_id else 0


class RefHeap:
    """ctypes view of the reference's mm_heap_t."""

    def __init__(self, lib, k):
        self.lib, self.cmp, self.live = lib, _CMP(_cmp_by_score), {}
        for name, res, args in (("mmh_init_with_size", C.c_void_p, [C.c_size_t, _CMP, C.c_void_p, C.c_void_p]),
                                ("mmh_insert", None, [C.c_void_p, C.c_void_p]), ("mmh_pop_min", C.c_void_p, [C.c_void_p]),
                                ("mmh_peek_max", C.c_void_p, [C.c_void_p]), ("mmh_exchange_max", C.c_void_p, [C.c_void_p, C.c_void_p]),
                                ("mmh_clear", None, [C.c_void_p]), ("mmh_free", None, [C.c_void_p])):
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        self.h = lib.mmh_init_with_size(max(k, 1), self.cmp, None, None)
        self.count = 0

    def _new(self, doc_id, score):
        r = _Res(doc_id, score)
        self.live[C.addressof(r)] = r
        return C.addressof(r)

    def insert(self, doc_id, score):
        self.lib.mmh_insert(self.h, self._new(doc_id, score))
        self.count += 1

    def exchange_max(self, doc_id, score):
        old = self.lib.mmh_exchange_max(self.h, self._new(doc_id, score))
        self.live.pop(old, None)

    def peek_max_score(self):
        return C.cast(self.lib.mmh_peek_max(self.h), C.POINTER(_Res)).contents.score

    def pop_min(self):
        p = self.lib.mmh_pop_min(self.h)
        r = C.cast(p, C.POINTER(_Res)).contents
        out = (int(r.doc_id), float(r.score))
        self.live.pop(p, None)
        self.count -= 1
        return out

    def clear(self):
        self.lib.mmh_clear(self.h)        # (free_func is NULL: the elements are ours)
        self.live.clear()
        self.count = 0

    def __del__(self):
        try:
            self.lib.mmh_free(self.h)
        except Exception:
            pass


class PyHeap:
    """Stand-in with the same interface.  It keeps the same SET as the reference heap (the evicted "max" is the highest
    distance and, among equals, the SMALLEST doc id -- cmpVecSimResByScore's tie rule; checked against traces of the
    real heap, tests/golden/ref_minmax_heap.json); the order in which equal distances are yielded is the reference
    heap's internal business and is only reproduced by RefHeap."""

    def __init__(self, k):
        self.items, self.count = [], 0

    def insert(self, doc_id, score):
        self.items.append((score, doc_id))
        self.count += 1

    def exchange_max(self, doc_id, score):
        self.items.remove(max(self.items, key=lambda t: (t[0], -t[1])))
        self.items.append((score, doc_id))

    def peek_max_score(self):
        return max(self.items)[0]

    def pop_min(self):
        s, d = min(self.items)
        self.items.remove((s, d))
        self.count -= 1
        return d, s

    def clear(self):
        self.items, self.count = [], 0


def reference_heap_available():
    return O.ref_lib("libref_mmheap") is not None


def make_heap(k, force_python=False):
    lib = None if force_python else O.ref_lib("libref_mmheap")
    return RefHeap(lib, k) if lib is not None else PyHeap(k)


# ---- the child iterator (a sorted id list, like NewSortedIdListIterator) ----------------------------------
class IdListChild:
    OK, NOTFOUND, EOF = 0, 1, 2

    def __init__(self, ids, estimate=None):
        self.ids, self.pos, self.last = sorted(int(i) for i in ids), 0, 0
        self.estimate = len(self.ids) if estimate is None else estimate

    def num_estimated(self):
        return self.estimate

    def rewind(self):
        self.pos, self.last = 0, 0

    def read(self):
        if self.pos >= len(self.ids):
            return self.EOF
        self.last = self.ids[self.pos]
        self.pos += 1
        return self.OK

    def skip_to(self, doc_id):
        import bisect
        self.pos = bisect.bisect_left(self.ids, doc_id, self.pos)
        if self.pos >= len(self.ids):
            return self.EOF
        self.last = self.ids[self.pos]
        self.pos += 1
        return self.OK if self.last == doc_id else self.NOTFOUND


# ---- index adapters -----------------------------------------------------------------------------------------
class OracleIndex:
    def __init__(self, idx):
        self.idx = idx

    def index_size(self):
        return len(self.idx)

    def topk(self, q, k):
        ids, sc = self.idx.topk(q, k)
        return ids.tolist(), sc.tolist()

    def batches(self, q):
        it = self.idx.batches(q)
        return it.has_next, lambda n: tuple(x.tolist() for x in it.next(n, O.BY_ID)), lambda: None

    def adhoc_distances(self, q, labels):
        nq = self.idx.normalized_query(q)
        return [self.idx.distance_from(int(l), nq) for l in labels]

    def prefer_adhoc(self, subset, k, initial):
        return bool(O.prefer_adhoc(len(self.idx), self.idx.dim, subset, k, initial)[0])


class GpuIndex:
    """the MI355X engine through the VecSim C ABI (redisearch_amd.vecsim.VecSimIndex)"""

    def __init__(self, idx):
        self.idx = idx

    def index_size(self):
        return self.idx.index_size()

    def topk(self, q, k):
        ids, sc = self.idx.topk_query(q, k).results()
        return ids.tolist(), sc.tolist()

    def batches(self, q):
        from redisearch_amd import vecsim as V
        it = self.idx.batch_iterator(q)
        return it.has_next, lambda n: tuple(x.tolist() for x in it.next(n, V.BY_ID).results()), it.free

    def adhoc_distances(self, q, labels):
        nq = self.idx.normalized_query(q)
        return [self.idx.get_distance_from_unsafe(int(l), nq) for l in labels]

    def prefer_adhoc(self, subset, k, initial):
        return bool(self.idx.prefer_adhoc_search(subset, k, initial))


# ---- the iterator -------------------------------------------------------------------------------------------
class HybridReplay:
    def __init__(self, index, q, k, child=None, policy=0, batch_size=0, force_python_heap=False):
        self.index, self.q, self.k, self.child = index, q, k, child
        self.batch_size, self.num_iterations, self.batch_sizes = batch_size, 0, []
        if child is None or k == 0:
            self.mode = STANDARD_KNN
        else:
            subset = min(child.num_estimated(), index.index_size())
            if policy:
                self.mode = policy
            else:
                self.mode = HYBRID_ADHOC_BF if index.prefer_adhoc(subset, k, True) else HYBRID_BATCHES
            self.heap = make_heap(k, force_python_heap)
        self.policy = policy

    def _insert(self, doc_id, score):                       # insertResultToHeap_Metric
        if self.heap.count < self.k:
            self.heap.insert(doc_id, score)
        else:
            self.heap.exchange_max(doc_id, score)
        return self.heap.peek_max_score()

    def _compute_distances(self):                           # computeDistances_RAM
        upper = math.inf
        ids = []
        while self.child.read() != IdListChild.EOF:
            ids.append(self.child.last)
        for doc_id, metric in zip(ids, self.index.adhoc_distances(self.q, ids)):
            if math.isnan(metric):
                continue
            if self.heap.count < self.k or metric < upper:
                upper = self._insert(doc_id, metric)

    def _alternating(self, batch_ids, batch_scores, upper):
        i, n = 0, len(batch_ids)
        child = self.child
        cs = child.read()
        vs = IdListChild.OK if n else IdListChild.EOF
        while cs == IdListChild.OK and vs == IdListChild.OK:
            if batch_ids[i] == child.last:
                if self.heap.count < self.k or batch_scores[i] < upper:
                    upper = self._insert(batch_ids[i], batch_scores[i])
                cs = child.read()
                i += 1
                vs = IdListChild.OK if i < n else IdListChild.EOF
            elif batch_ids[i] > child.last:
                cs = child.skip_to(batch_ids[i])
                if cs == IdListChild.NOTFOUND:
                    cs = IdListChild.OK
            elif i + 1 < n:                                 # HR_SkipToInBatch: next batch entry with id >= child.last
                i += 1
                while i < n and batch_ids[i] < child.last:
                    i += 1
                vs = IdListChild.OK if i < n else IdListChild.EOF
            else:
                break
        return upper

    def _prepare(self):
        idx, k = self.index, self.k
        if self.mode == STANDARD_KNN:
            ids, sc = idx.topk(self.q, k) if k else ([], [])
            return list(zip(ids, sc))
        if self.mode == HYBRID_ADHOC_BF:
            self._compute_distances()
            return None
        if self.child.num_estimated() == 0:
            return None
        has_next, nxt, free = idx.batches(self.q)
        upper = math.inf
        est = min(self.child.num_estimated(), idx.index_size())
        child_upper = est
        try:
            while has_next():
                self.num_iterations += 1
                size = idx.index_size()
                left = k - self.heap.count
                # size_t batch = n_res_left * ((float)vec_index_size / child_num_estimated) + 1   (fp32, truncated)
                bs = self.batch_size or int(np.float32(left) * (np.float32(size) / np.float32(est)) + np.float32(1))
                self.batch_sizes.append(bs)
                ids, sc = nxt(bs)
                self.child.rewind()
                upper = self._alternating(ids, sc, upper)
                if self.heap.count == k:
                    break
                # reviewHybridSearchPolicy
                if self.policy == HYBRID_BATCHES and self.batch_size:
                    continue
                new_results = self.heap.count - (k - left)
                cur = int((np.float32(new_results) / np.float32(left)) * np.float32(size))   # float cur_ratio * size
                est = min((est + cur) // 2, child_upper)
                if self.policy == HYBRID_BATCHES:
                    continue
                if idx.prefer_adhoc(est, k, False):
                    self.mode = HYBRID_BATCHES_TO_ADHOC_BF
                    self.heap.clear()
                    self.child.rewind()
                    self._compute_distances()
                    return None
        finally:
            free()
        return None

    def results(self):
        """[(doc id, distance)] in the order the iterator's Read() yields them."""
        knn = self._prepare()
        if knn is not None:
            return knn
        out = []
        while self.heap.count:
            out.append(self.heap.pop_min())
        return out
