"""ctypes binding of include/rsgpu_search.h: GPU posting-list decode, intersection, scorers, score
top-N and the hybrid ad-hoc KNN step.  Thin binding only -- all work happens in the HIP library."""
import ctypes as C

import numpy as np

from . import vecsim as V

_sz, _vp, _dbl, _i = C.c_size_t, C.c_void_p, C.c_double, C.c_int

(CODEC_FULL, CODEC_FREQS_FIELDS, CODEC_FREQS_ONLY, CODEC_FIELDS_ONLY, CODEC_FIELDS_OFFSETS, CODEC_OFFSETS_ONLY,
 CODEC_FREQS_OFFSETS, CODEC_DOCIDS_ONLY, CODEC_RAW_DOCIDS, CODEC_FULL_WIDE, CODEC_FREQS_FIELDS_WIDE,
 CODEC_FIELDS_ONLY_WIDE, CODEC_FIELDS_OFFSETS_WIDE) = range(13)
SCORERS = {"BM25STD": 0, "BM25STD.TANH": 1, "BM25": 2, "TFIDF": 3, "TFIDF.DOCNORM": 4, "DOCSCORE": 5, "DISMAX": 6}
# scorer names that are a registered scorer + a result processor chained behind it in the reference pipeline
PIPELINE_SCORERS = {"BM25STD.NORM": 7}


class ScoreArgs(C.Structure):
    _fields_ = [("scorer", _i), ("num_docs", _sz), ("avg_doc_len", _dbl), ("tanh_factor", C.c_uint64),
                ("root_weight", _dbl), ("min_score", _dbl), ("idf", _vp), ("bm25_idf", _vp), ("weight", _vp)]


class HybridQueryArgs(C.Structure):
    _fields_ = [("lists", _vp), ("n_lists", _sz), ("table", _vp), ("score", C.POINTER(ScoreArgs)), ("top_n", _sz),
                ("index", _vp), ("query", _vp), ("k", _sz), ("top_ids", _vp), ("top_scores", _vp), ("knn_ids", _vp),
                ("knn_dists", _vp), ("n_hits", _sz), ("n_top", _sz), ("n_knn", _sz), ("hits_out", _vp),
                ("timeout_cb", _vp), ("timeout_ctx", _vp)]   # round 6: int (*)(void *ctx), NULL = no deadline


TIMEOUT_CB = C.CFUNCTYPE(C.c_int, C.c_void_p)     # the reference's / VecSim's timeoutCallbackFunction shape
TIMED_OUT = 1                                     # RSGPU_TIMED_OUT


class TreeQuery(C.Structure):
    _fields_ = [("root_op", _i), ("n_groups", _sz), ("group_first", _vp), ("group_op", _vp), ("group_weight", _vp),
                ("lists", _vp), ("max_slop", C.c_long), ("in_order", _i)]


class TreeNode(C.Structure):
    _fields_ = [("op", _i), ("list", _sz), ("n_children", _sz), ("weight", C.c_double), ("max_slop", C.c_long),
                ("in_order", _i)]


OP_TERM, OP_UNION, OP_INTERSECT, OP_NOT = 0, 1, 2, 3

ABI = {
    "RSGPU_EvalTree": (_vp, [C.POINTER(TreeQuery)]),
    "RSGPU_EvalTreeNodes": (_vp, [C.POINTER(TreeNode), _sz, _vp, _sz]),
    "RSGPU_Hits_TreeNodes": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "RSGPU_HybridQuery": (_i, [C.POINTER(HybridQueryArgs)]),
    "RSGPU_HybridTreeQuery": (_i, [C.POINTER(TreeQuery), C.POINTER(HybridQueryArgs)]),
    "RSGPU_HybridTreeNodesQuery": (_i, [C.POINTER(TreeNode), _sz, C.POINTER(HybridQueryArgs)]),
    "RSGPU_HybridQueryPath": (_i, []),
    "RSGPU_GetHybridCoalesceStats": (None, [_vp, _i]),
    "RSGPU_HybridTrace": (C.c_long, [_vp, _sz]),
    "RSGPU_Postings_Upload": (_vp, [_i, _sz, _vp, _vp, _vp, _vp, _vp]),
    "RSGPU_Postings_Free": (None, [_vp]),
    "RSGPU_Postings_NumEntries": (_sz, [_vp]),
    "RSGPU_Postings_NumBytes": (_sz, [_vp]),
    "RSGPU_Postings_Decode": (C.c_long, [_vp, _vp, _vp, _vp]),
    "RSGPU_Intersect": (_vp, [_vp, _sz]),
    "RSGPU_IntersectEx": (_vp, [_vp, _sz, C.c_long, _i]),
    "RSGPU_Postings_DecodeWideMasks": (C.c_long, [_vp, _vp, _vp]),
    "RSGPU_Hits_Free": (None, [_vp]),
    "RSGPU_Hits_Len": (_sz, [_vp]),
    "RSGPU_Hits_Read": (_i, [_vp, _vp, _vp]),
    "RSGPU_DocTable_Upload": (_vp, [_sz, _vp, _vp, _vp]),
    "RSGPU_DocTable_UploadWindow": (_vp, [C.c_uint64, _sz, _vp, _vp, _vp]),
    "RSGPU_DocTable_Free": (None, [_vp]),
    "RSGPU_Hits_Score": (_i, [_vp, _vp, C.POINTER(ScoreArgs), _vp]),
    "RSGPU_Hits_TopN": (C.c_long, [_vp, _sz, _vp, _vp]),
    "RSGPU_Hits_KnnRerank": (C.c_long, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "RSGPU_Union": (_vp, [_vp, _sz]),
    "RSGPU_Not": (_vp, [_vp, _vp, C.c_uint64]),
    "RSGPU_HybridFuse": (C.c_long, [_i, _dbl, _vp, _i, _vp, _vp, _sz, _vp, _vp, _sz, _sz, _sz, _vp, _vp]),
    "RSGPU_CalculateIDF": (_dbl, [_sz, _sz]),
    "RSGPU_CalculateIDF_BM25": (_dbl, [_sz, _sz]),
    "RSGPU_SearchProfile": (None, [C.POINTER(_dbl)] * 5),
    # record access for the iterator seam (include/rsgpu_search.h)
    "RSGPU_Postings_Codec": (_i, [_vp]),
    "RSGPU_Hits_NumLeaves": (_sz, [_vp]),
    "RSGPU_Hits_IsUnion": (_i, [_vp]),
    "RSGPU_Hits_LeafOrder": (_i, [_vp, _vp]),
    "RSGPU_Hits_Tree": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "RSGPU_Hits_ReadRange": (C.c_long, [_vp, _sz, _sz, _vp]),
    "RSGPU_Hits_ReadRecords": (C.c_long, [_vp, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "RSGPU_Postings_ReadBytes": (_i, [_vp, _sz, _sz, _vp]),
}
V.EXTRA_ABI = ABI

_lib = None


def load():
    global _lib
    if _lib is None:
        lib = V.load()
        for name, (res, args) in ABI.items():
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(_vp)


def _check(ptr, what):
    if not ptr:
        raise RuntimeError(what + " failed: " + V.last_error())
    return ptr


class Postings:
    """A posting list resident in HBM in the reference's block format (uploaded as-is)."""

    def __init__(self, codec, first, last, num_entries, offset, data):
        self.lib = load()
        first, last = np.ascontiguousarray(first, np.uint64), np.ascontiguousarray(last, np.uint64)
        nent, off = np.ascontiguousarray(num_entries, np.uint32), np.ascontiguousarray(offset, np.uint64)
        data = np.ascontiguousarray(data, np.uint8)
        if data.size == 0:
            data = np.zeros(1, np.uint8)
        self.codec = codec
        self.ptr = _check(self.lib.RSGPU_Postings_Upload(codec, len(first), _p(first), _p(last), _p(nent), _p(off), _p(data)),
                          "RSGPU_Postings_Upload")

    @classmethod
    def from_flat(cls, fl):
        """fl: dict(first,last,num_entries,offset,bytes,codec) as produced by the block writer."""
        return cls(fl["codec"], fl["first"], fl["last"], fl["num_entries"], fl["offset"], fl["bytes"])

    @property
    def num_entries(self):
        return self.lib.RSGPU_Postings_NumEntries(self.ptr)

    @property
    def num_bytes(self):
        return self.lib.RSGPU_Postings_NumBytes(self.ptr)

    def decode(self):
        n = self.num_entries
        ids, fr, mk = np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32)
        m = self.lib.RSGPU_Postings_Decode(self.ptr, _p(ids), _p(fr), _p(mk))
        if m < 0:
            raise RuntimeError(V.last_error())
        return ids[:m], fr[:m], mk[:m]

    def decode_wide_masks(self):
        n = self.num_entries
        lo, hi = np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint64)
        m = self.lib.RSGPU_Postings_DecodeWideMasks(self.ptr, _p(lo), _p(hi))
        if m < 0:
            raise RuntimeError(V.last_error())
        return [int(a) | (int(b) << 64) for a, b in zip(lo[:m].tolist(), hi[:m].tolist())]

    def read_bytes(self, pos, length):
        out = np.zeros(max(length, 1), np.uint8)
        if self.lib.RSGPU_Postings_ReadBytes(self.ptr, pos, length, _p(out)) != 0:
            raise RuntimeError(V.last_error())
        return out[:length]

    def free(self):
        if getattr(self, "ptr", None):
            self.lib.RSGPU_Postings_Free(self.ptr)
            self.ptr = None

    __del__ = free


class DocTable:
    def __init__(self, doc_len, doc_score, max_freq=None, first_doc_id=0):
        """entry j describes doc id first_doc_id + j"""
        self.lib = load()
        dl, ds = np.ascontiguousarray(doc_len, np.uint32), np.ascontiguousarray(doc_score, np.float32)
        mf = np.ascontiguousarray(max_freq, np.uint32) if max_freq is not None else None
        self.ptr = _check(self.lib.RSGPU_DocTable_UploadWindow(int(first_doc_id), len(dl), _p(dl), _p(ds),
                                                               _p(mf) if mf is not None else None),
                          "RSGPU_DocTable_UploadWindow")

    def free(self):
        if getattr(self, "ptr", None):
            self.lib.RSGPU_DocTable_Free(self.ptr)
            self.ptr = None

    __del__ = free


class Hits:
    def __init__(self, lists, op="and", universe=None, max_doc_id=0, max_slop=None, in_order=False):
        self.lib = load()
        self._lists = list(lists)   # the hit list borrows the postings (offset bytes): keep them alive
        if op == "not":
            self.n_lists = 1
            self.ptr = _check(self.lib.RSGPU_Not(lists[0].ptr, universe.ptr if universe is not None else None,
                                                 max_doc_id), "RSGPU_Not")
            return
        self.n_lists = len(lists)
        arr = (_vp * len(lists))(*[l.ptr for l in lists])
        if op == "and":
            self.ptr = _check(self.lib.RSGPU_IntersectEx(C.cast(arr, _vp), len(lists), -1 if max_slop is None else int(max_slop),
                                                         int(in_order)), "RSGPU_IntersectEx")
        else:
            self.ptr = _check(self.lib.RSGPU_Union(C.cast(arr, _vp), len(lists)), "RSGPU_Union")

    def __len__(self):
        return self.lib.RSGPU_Hits_Len(self.ptr)

    def read(self):
        n = len(self)
        ids = np.zeros(max(n, 1), np.uint64)
        fr = np.zeros((self.n_lists, max(n, 1)), np.uint32)
        if self.lib.RSGPU_Hits_Read(self.ptr, _p(ids), _p(fr)) != 0:
            raise RuntimeError(V.last_error())
        return ids[:n], fr[:, :n]

    def score(self, table, scorer, idf, bm25_idf, weight, num_docs, avg_doc_len, root_weight=1.0, min_score=0.0,
              tanh_factor=4, want_scores=True):
        idf, bidf, w = (np.ascontiguousarray(x, np.float64) for x in (idf, bm25_idf, weight))
        a = ScoreArgs(SCORERS[scorer] if scorer in SCORERS else PIPELINE_SCORERS[scorer], num_docs, avg_doc_len, tanh_factor, root_weight, min_score, _p(idf).value,
                      _p(bidf).value, _p(w).value)
        out = np.zeros(max(len(self), 1), np.float64) if want_scores else None
        if self.lib.RSGPU_Hits_Score(self.ptr, table.ptr, C.byref(a), _p(out) if want_scores else None) != 0:
            raise RuntimeError(V.last_error())
        return out[: len(self)] if want_scores else None

    def topn(self, n):
        ids, sc = np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.float64)
        m = self.lib.RSGPU_Hits_TopN(self.ptr, n, _p(ids), _p(sc))
        if m < 0:
            raise RuntimeError(V.last_error())
        return ids[:m], sc[:m]

    def knn_rerank(self, index, q, k):
        ids, d = np.zeros(max(k, 1), np.uint64), np.zeros(max(k, 1), np.float64)
        m = self.lib.RSGPU_Hits_KnnRerank(self.ptr, index.ptr, V._p(index._q(q)), k, _p(ids), _p(d))
        if m < 0:
            raise RuntimeError(V.last_error())
        return ids[:m], d[:m]

    def leaf_order(self):
        """list index (in the caller's array) of every child slot of the aggregate, in iteration order"""
        o = np.zeros(32, np.int32)
        n = self.lib.RSGPU_Hits_LeafOrder(self.ptr, _p(o))
        return o[:n].tolist()

    def read_range(self, first, count):
        ids = np.zeros(max(count, 1), np.uint64)
        m = self.lib.RSGPU_Hits_ReadRange(self.ptr, first, count, _p(ids))
        if m < 0:
            raise RuntimeError(V.last_error())
        return ids[:m]

    def read_records(self, lst, first=0, count=None):
        """-> dict(entry, freq, mask (Python ints, 128 bit), off_pos, off_len) of hits [first, first+count) for list `lst`"""
        count = len(self) - first if count is None else count
        c = max(count, 1)
        e, f, ol = np.zeros(c, np.uint32), np.zeros(c, np.uint32), np.zeros(c, np.uint32)
        lo, hi, op = np.zeros(c, np.uint64), np.zeros(c, np.uint64), np.zeros(c, np.uint64)
        m = self.lib.RSGPU_Hits_ReadRecords(self.ptr, lst, first, count, _p(e), _p(f), _p(lo), _p(hi), _p(op), _p(ol))
        if m < 0:
            raise RuntimeError(V.last_error())
        return dict(entry=e[:m], freq=f[:m], mask=[int(a) | (int(b) << 64) for a, b in zip(lo[:m].tolist(), hi[:m].tolist())],
                    off_pos=op[:m], off_len=ol[:m])

    def free(self):
        if getattr(self, "ptr", None):
            self.lib.RSGPU_Hits_Free(self.ptr)
            self.ptr = None

    __del__ = free


class HybridQuery:
    """RSGPU_HybridQuery with its argument block prepared once: intersection -> (score + top_n) || (ad-hoc KNN top-k).
    run() is the bare C call (what a C caller pays); results() reads the outputs of the last run."""

    def __init__(self, lists, table=None, scorer=None, idf=None, bm25_idf=None, weight=None, num_docs=0, avg_doc_len=1.0,
                 top_n=0, index=None, q=None, k=0, root_weight=1.0, min_score=0.0, tanh_factor=4, want_hits=False):
        self.lib = load()
        self._fn = self.lib.RSGPU_HybridQuery
        self._arr = (_vp * len(lists))(*[l.ptr for l in lists])
        a = self.args = HybridQueryArgs()
        a.lists, a.n_lists = C.cast(self._arr, _vp), len(lists)
        self._keep = [lists, table, index]
        self._lists = list(lists)
        self._hits_ptr = _vp() if want_hits else None
        if want_hits:
            a.hits_out = C.cast(C.pointer(self._hits_ptr), _vp)
        if table is not None and scorer is not None and top_n:
            idf_, bidf_, w_ = (np.ascontiguousarray(x, np.float64) for x in (idf, bm25_idf, weight))
            sa = ScoreArgs(SCORERS[scorer] if scorer in SCORERS else PIPELINE_SCORERS[scorer], num_docs, avg_doc_len, tanh_factor,
                           root_weight, min_score, _p(idf_).value, _p(bidf_).value, _p(w_).value)
            self._keep += [idf_, bidf_, w_, sa]
            a.table, a.score, a.top_n = table.ptr, C.pointer(sa), top_n
        self.ti, self.ts = np.zeros(max(top_n, 1), np.uint64), np.zeros(max(top_n, 1), np.float64)
        self.ki, self.kd = np.zeros(max(k, 1), np.uint64), np.zeros(max(k, 1), np.float64)
        a.top_ids, a.top_scores = _p(self.ti).value, _p(self.ts).value
        a.knn_ids, a.knn_dists = _p(self.ki).value, _p(self.kd).value
        if index is not None and q is not None and k:
            qb = index._q(q)
            self._keep.append(qb)
            a.index, a.query, a.k = index.ptr, _p(qb).value, k
        self._ref = C.byref(a)

    def _call(self):
        return self._fn(self._ref)

    def set_timeout(self, fn):
        """fn() -> truthy once the deadline has passed (polled by the library: RSGPU_HybridQueryArgs.timeout_cb); None clears it"""
        self._tcb = TIMEOUT_CB(lambda ctx: int(bool(fn()))) if fn is not None else None
        self.args.timeout_cb = C.cast(self._tcb, _vp) if fn is not None else None

    def run(self):
        """the bare C call; returns False when the deadline passed (RSGPU_TIMED_OUT: empty outputs), raises on an error"""
        if self._hits_ptr is not None and self._hits_ptr.value:      # the previous run's hit list
            self.lib.RSGPU_Hits_Free(self._hits_ptr)
            self._hits_ptr.value = None
        rc = self._call()
        if rc == TIMED_OUT:
            return False
        if rc != 0:
            raise RuntimeError(V.last_error())
        return True

    def results(self):
        a = self.args
        return dict(n_hits=a.n_hits, top=(self.ti[:a.n_top].copy(), self.ts[:a.n_top].copy()),
                    knn=(self.ki[:a.n_knn].copy(), self.kd[:a.n_knn].copy()))

    def take_hits(self):
        """the hit list of the last run (want_hits=True) as a Hits object that owns it"""
        h = Hits.__new__(Hits)
        h.lib, h._lists, h.n_lists = self.lib, self._lists, len(self._lists)
        h.ptr = _check(self._hits_ptr.value, "hits_out")
        self._hits_ptr.value = None
        return h

    def __del__(self):
        if getattr(self, "_hits_ptr", None) is not None and self._hits_ptr.value:
            self.lib.RSGPU_Hits_Free(self._hits_ptr)
            self._hits_ptr.value = None


def hybrid_path():
    """how this thread's last RSGPU_HybridQuery / RSGPU_HybridTreeQuery ran: 0 staged pipeline, 1 two launches, 2 the general
    tile kernel (hybrid_kernels.hip)"""
    return load().RSGPU_HybridQueryPath()


def hybrid_coalesce_stats(reset=False):
    """the hybrid coalescer's counters (rsgpu_search.h RSGPU_GetHybridCoalesceStats)"""
    out = (C.c_uint64 * 5)()
    load().RSGPU_GetHybridCoalesceStats(out, 1 if reset else 0)
    return dict(zip(("alone", "grids", "grid_queries", "queued", "relaunched"), (int(v) for v in out)))


def hybrid_trace(max_tiles=1 << 15):
    """[tiles, 9] phase clock (100 MHz ticks) of the last two-launch query of this thread (knob hybrid_trace = 1)"""
    out = np.zeros((max_tiles, 9), np.uint64)
    n = load().RSGPU_HybridTrace(_p(out), max_tiles)
    if n < 0:
        raise RuntimeError(V.last_error())
    return out[:n]


def hybrid_query(lists, table=None, scorer=None, idf=None, bm25_idf=None, weight=None, num_docs=0, avg_doc_len=1.0,
                 top_n=0, index=None, q=None, k=0, root_weight=1.0, min_score=0.0, tanh_factor=4):
    """One-shot form of HybridQuery.  Returns dict(n_hits, top=(ids, scores), knn=(ids, dists))."""
    hq = HybridQuery(lists, table, scorer, idf, bm25_idf, weight, num_docs, avg_doc_len, top_n, index, q, k, root_weight,
                     min_score, tanh_factor)
    hq.run()
    return hq.results()


class HybridTreeQuery(HybridQuery):
    """RSGPU_HybridTreeQuery: HybridQuery over a two-level tree -- groups as TreeHits takes them; idf / bm25_idf / weight per
    list in the flattened order of `groups`."""

    def __init__(self, root_op, groups, max_slop=None, in_order=False, **kw):
        flat, first, ops, ws = [], [0], [], []
        for op, w, ls in groups:
            flat += list(ls)
            first.append(len(flat))
            ops.append(op)
            ws.append(w)
        HybridQuery.__init__(self, flat, **kw)
        self._gf, self._go, self._gw = np.asarray(first, np.uint64), np.asarray(ops, np.int32), np.asarray(ws, np.float64)
        self._tq = TreeQuery(root_op, len(groups), _p(self._gf).value, _p(self._go).value, _p(self._gw).value,
                             C.cast(self._arr, _vp).value, -1 if max_slop is None else int(max_slop), int(in_order))
        self._tfn = self.lib.RSGPU_HybridTreeQuery

    def _call(self):
        return self._tfn(C.byref(self._tq), self._ref)


def tree_nodes(tree):
    """nested tuples -- ("t", list_index) | ("and" | "or", weight, [children...][, max_slop, in_order]) | ("not", weight, [terms...]) --
    as the post-order RSGPU_TreeNode array"""
    nodes = []

    def walk(t):
        if t[0] == "t":
            nodes.append(TreeNode(OP_TERM, int(t[1]), 0, 1.0, -1, 0))
            return
        for ch in t[2]:
            walk(ch)
        ms = t[3] if len(t) > 3 and t[3] is not None else -1
        io = int(bool(t[4])) if len(t) > 4 else 0
        nodes.append(TreeNode({"and": OP_INTERSECT, "or": OP_UNION, "not": OP_NOT}[t[0]], 0, len(t[2]), float(t[1]), int(ms), io))
    walk(tree)
    return (TreeNode * len(nodes))(*nodes), len(nodes)


class HybridNodesQuery(HybridQuery):
    """RSGPU_HybridTreeNodesQuery: HybridQuery over a query tree of any depth (nested tuples as NodeHits takes them); idf /
    bm25_idf / weight per list in the order of `lists`."""

    def __init__(self, tree, lists, **kw):
        HybridQuery.__init__(self, lists, **kw)
        self._nodes, self._n_nodes = tree_nodes(tree)
        self._nfn = self.lib.RSGPU_HybridTreeNodesQuery

    def _call(self):
        return self._nfn(self._nodes, self._n_nodes, self._ref)


class TreeHits(Hits):
    """RSGPU_EvalTree: root_op over groups; groups = [(op, weight, [Postings...]), ...] (op OP_TERM takes one list).
    Scoring arrays (idf, bm25_idf, weight) are per list in the flattened order of `groups`."""

    def __init__(self, root_op, groups, max_slop=None, in_order=False):
        self.lib = load()
        flat, first, ops, ws = [], [0], [], []
        for op, w, ls in groups:
            flat += list(ls)
            first.append(len(flat))
            ops.append(op)
            ws.append(w)
        self._lists = flat
        self.n_lists = len(flat)
        arr = (_vp * len(flat))(*[l.ptr for l in flat])
        gf = np.asarray(first, np.uint64)
        go = np.asarray(ops, np.int32)
        gw = np.asarray(ws, np.float64)
        q = TreeQuery(root_op, len(groups), _p(gf).value, _p(go).value, _p(gw).value, C.cast(arr, _vp).value,
                      -1 if max_slop is None else int(max_slop), int(in_order))
        self.ptr = _check(self.lib.RSGPU_EvalTree(C.byref(q)), "RSGPU_EvalTree")


class NodeHits(Hits):
    """RSGPU_EvalTreeNodes: a query tree of any depth.  `tree` is nested tuples: ("t", list_index) for a term,
    ("and" | "or", weight, [children...]) or ("and", weight, [children...], max_slop, in_order) for an aggregate.
    Scoring arrays (idf, bm25_idf, weight) are per list, in the order of `lists`."""

    def __init__(self, tree, lists):
        self.lib = load()
        self._lists = list(lists)
        self.n_lists = len(lists)
        nodes = []

        def walk(t):
            if t[0] == "t":
                nodes.append(TreeNode(OP_TERM, int(t[1]), 0, 1.0, -1, 0))
                return
            for ch in t[2]:
                walk(ch)
            ms = t[3] if len(t) > 3 and t[3] is not None else -1
            io = int(bool(t[4])) if len(t) > 4 else 0
            nodes.append(TreeNode(OP_INTERSECT if t[0] == "and" else OP_UNION, 0, len(t[2]), float(t[1]), int(ms), io))
        walk(tree)
        arr = (TreeNode * len(nodes))(*nodes)
        lp = (_vp * len(lists))(*[l.ptr for l in lists])
        self.ptr = _check(self.lib.RSGPU_EvalTreeNodes(arr, len(nodes), C.cast(lp, _vp), len(lists)), "RSGPU_EvalTreeNodes")

    def tree_nodes(self):
        """[(op, leaf, n_children, weight)] post-order, as the hit list holds it"""
        op, leaf, nch = (np.zeros(64, np.int32) for _ in range(3))
        w = np.zeros(64, np.float64)
        n = self.lib.RSGPU_Hits_TreeNodes(self.ptr, _p(op), _p(leaf), _p(nch), _p(w))
        return [(int(op[i]), int(leaf[i]), int(nch[i]), float(w[i])) for i in range(n)]


def intersect(lists, max_slop=None, in_order=False):
    return Hits(lists, "and", max_slop=max_slop, in_order=in_order)


def union(lists):
    return Hits(lists, op="or")


def negate(child, max_doc_id, universe=None):
    return Hits([child], op="not", universe=universe, max_doc_id=max_doc_id)


def calculate_idf(total, term):
    return load().RSGPU_CalculateIDF(total, term)


def calculate_idf_bm25(total, term):
    return load().RSGPU_CalculateIDF_BM25(total, term)


RRF, LINEAR = 0, 1


def hybrid_fuse(scoring, search_ids, search_scores, vec_ids, vec_scores, window, top_n=None, constant=60.0,
                weights=(0.5, 0.5), metric=-1):
    """FT.HYBRID fusion on the device (RSGPU_HybridFuse): ranked search list + ranked vector list -> fused list."""
    a_ids, b_ids = np.ascontiguousarray(search_ids, np.uint64), np.ascontiguousarray(vec_ids, np.uint64)
    a_sc, b_sc = np.ascontiguousarray(search_scores, np.float64), np.ascontiguousarray(vec_scores, np.float64)
    w = np.ascontiguousarray(weights, np.float64)
    cap = len(a_ids) + len(b_ids) + 1
    top_n = cap if top_n is None else top_n
    ids, sc = np.zeros(cap, np.uint64), np.zeros(cap, np.float64)
    m = load().RSGPU_HybridFuse(scoring, constant, _p(w), metric, _p(a_ids), _p(a_sc), len(a_ids), _p(b_ids), _p(b_sc),
                                len(b_ids), window, min(top_n, cap), _p(ids), _p(sc))
    if m < 0:
        raise RuntimeError(V.last_error())
    return ids[:m], sc[:m]


def profile():
    v = [_dbl(0) for _ in range(5)]
    load().RSGPU_SearchProfile(*[C.byref(x) for x in v])
    return dict(zip(("decode_ms", "intersect_ms", "score_ms", "topn_ms", "knn_ms"), [x.value for x in v]))


# ---- Boundary 3: librsgpu_iterators.so, the reference's QueryIterator vtable over device hit lists (include/rs_iterator.h) --
class TermArg(C.Structure):
    _fields_ = [("postings", _vp), ("term", _vp), ("weight", _dbl)]


class QueryIteratorStruct(C.Structure):
    """reference src/iterators/iterator_api.h:46-151"""
    _fields_ = [("type", C.c_uint32), ("atEOF", C.c_bool), ("lastDocId", C.c_uint64), ("current", _vp),
                ("NumEstimated", _vp), ("Read", _vp), ("SkipTo", _vp), ("Revalidate", _vp), ("Free", _vp), ("Rewind", _vp),
                ("ProfileChildren", _vp), ("PrintProfile", _vp)]


IT_OK, IT_NOTFOUND, IT_EOF, IT_TIMEOUT = 0, 1, 2, 3
_itlib = None


def load_iterators(result_api_handle=None):
    """Loads librsgpu_iterators.so next to the engine.  result_api_handle: dlopen handle (ctypes CDLL._handle) of the
    library that implements the module's RSIndexResult constructors; None = look them up in the whole process."""
    global _itlib
    if _itlib is None:
        load()
        from . import build as B
        L = C.CDLL(B.lib_path("librsgpu_iterators.so"))
        L.RSGPU_Iterators_SetResultAPI.restype, L.RSGPU_Iterators_SetResultAPI.argtypes = _i, [_vp, _vp]
        L.RSGPU_Iterators_LastError.restype = C.c_char_p
        L.RSGPU_Iterators_SetBlock.restype, L.RSGPU_Iterators_SetBlock.argtypes = None, [_sz]
        L.RSGPU_NewIntersectionIterator.restype = _vp
        L.RSGPU_NewIntersectionIterator.argtypes = [_vp, _sz, C.c_int32, C.c_bool, _dbl]
        L.RSGPU_NewUnionIterator.restype, L.RSGPU_NewUnionIterator.argtypes = _vp, [_vp, _sz, _dbl]
        L.RSGPU_NewNotIterator.restype, L.RSGPU_NewNotIterator.argtypes = _vp, [_vp, _vp, C.c_uint64, _dbl]
        L.RSGPU_NewHitsIterator.restype, L.RSGPU_NewHitsIterator.argtypes = _vp, [_vp, _vp, _sz, _dbl, C.c_bool]
        L.RSGPU_Iterator_Hits.restype, L.RSGPU_Iterator_Hits.argtypes = _vp, [_vp]
        L.RSGPU_NewTreeIterator.restype, L.RSGPU_NewTreeIterator.argtypes = _vp, [C.POINTER(TreeQuery), _vp, _dbl]
        L.RSGPU_NewTreeNodesIterator.restype = _vp
        L.RSGPU_NewTreeNodesIterator.argtypes = [C.POINTER(TreeNode), _sz, _vp, _sz, _dbl]
        _itlib = L
    if result_api_handle is not None:
        if _itlib.RSGPU_Iterators_SetResultAPI(None, _vp(result_api_handle)) != 0:
            raise RuntimeError(_itlib.RSGPU_Iterators_LastError().decode())
    return _itlib


def term_args(lists, terms=None, weights=None):
    """RSGPU_TermArg array for posting lists `lists`; terms: RSQueryTerm pointers (ints) or None"""
    n = len(lists)
    arr = (TermArg * n)()
    for i, l in enumerate(lists):
        arr[i].postings = l.ptr
        arr[i].term = terms[i] if terms is not None else None
        arr[i].weight = 1.0 if weights is None else float(weights[i])
    return arr


def new_iterator(kind, lists, terms=None, weights=None, weight=1.0, max_slop=-1, in_order=False, universe=None, max_doc_id=0):
    """-> QueryIterator* (int).  kind: "and" | "or" | "not".  Free it through its own vtable (Free)."""
    L = load_iterators()
    if kind == "not":
        it = L.RSGPU_NewNotIterator(lists[0].ptr, universe.ptr if universe is not None else None, max_doc_id, weight)
    else:
        arr = term_args(lists, terms, weights)
        if kind == "and":
            it = L.RSGPU_NewIntersectionIterator(C.cast(arr, _vp), len(lists), max_slop, in_order, weight)
        else:
            it = L.RSGPU_NewUnionIterator(C.cast(arr, _vp), len(lists), weight)
    if not it:
        raise RuntimeError("iterator: " + L.RSGPU_Iterators_LastError().decode())
    return it


def new_tree_iterator(root_op, groups, terms=None, weights=None, weight=1.0, max_slop=None, in_order=False):
    """-> QueryIterator* over a two-level tree; groups = [(op, group_weight, [Postings...]), ...] as for TreeHits;
    terms / weights are per list in the flattened order of `groups`."""
    L = load_iterators()
    flat, first, ops, ws = [], [0], [], []
    for op, w, ls in groups:
        flat += list(ls)
        first.append(len(flat))
        ops.append(op)
        ws.append(w)
    gf, go, gw = np.asarray(first, np.uint64), np.asarray(ops, np.int32), np.asarray(ws, np.float64)
    q = TreeQuery(root_op, len(groups), _p(gf).value, _p(go).value, _p(gw).value, None,
                  -1 if max_slop is None else int(max_slop), int(in_order))
    arr = term_args(flat, terms, weights)
    it = L.RSGPU_NewTreeIterator(C.byref(q), C.cast(arr, _vp), weight)
    if not it:
        raise RuntimeError("tree iterator: " + L.RSGPU_Iterators_LastError().decode())
    return it


def tree_node_array(tree):
    """nested tuples (see NodeHits) -> ctypes array of RSGPU_TreeNode in post-order"""
    nodes = []

    def walk(t):
        if t[0] == "t":
            nodes.append(TreeNode(OP_TERM, int(t[1]), 0, 1.0, -1, 0))
            return
        for ch in t[2]:
            walk(ch)
        ms = t[3] if len(t) > 3 and t[3] is not None else -1
        io = int(bool(t[4])) if len(t) > 4 else 0
        nodes.append(TreeNode(OP_INTERSECT if t[0] == "and" else OP_UNION, 0, len(t[2]), float(t[1]), int(ms), io))
    walk(tree)
    return (TreeNode * len(nodes))(*nodes), len(nodes)


def new_tree_nodes_iterator(tree, lists, terms=None, weights=None, weight=1.0):
    """-> QueryIterator* over a query tree of any depth (nested tuples as for NodeHits); terms / weights per list."""
    L = load_iterators()
    arr, n = tree_node_array(tree)
    ta = term_args(lists, terms, weights)
    it = L.RSGPU_NewTreeNodesIterator(arr, n, C.cast(ta, _vp), len(lists), weight)
    if not it:
        raise RuntimeError("tree iterator: " + L.RSGPU_Iterators_LastError().decode())
    return it
