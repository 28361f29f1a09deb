"""ctypes front-end of the CPU oracle (oracle/*.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from redisearch_amd/.  See the headers of oracle/*.c for what each C
function restates (reference file:line) and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

# VecSimType / VecSimMetric numeric values (include/VecSim/vec_sim_common.h)
F32, F64, BF16, F16, I8, U8 = 0, 1, 2, 3, 4, 5
L2, IP, COSINE = 0, 1, 2
BY_SCORE, BY_ID = 0, 1
TYPE_NP = {F32: np.float32, F64: np.float64, F16: np.float16, BF16: np.uint16, I8: np.int8, U8: np.uint8}

(C_FULL, C_FREQS_FIELDS, C_FREQS_ONLY, C_FIELDS_ONLY, C_FIELDS_OFFSETS, C_OFFSETS_ONLY,
 C_FREQS_OFFSETS, C_DOCIDS_ONLY, C_RAW_DOCIDS, C_FULL_WIDE, C_FREQS_FIELDS_WIDE, C_FIELDS_ONLY_WIDE,
 C_FIELDS_OFFSETS_WIDE) = range(13)
WIDE_CODECS = (C_FULL_WIDE, C_FREQS_FIELDS_WIDE, C_FIELDS_ONLY_WIDE, C_FIELDS_OFFSETS_WIDE)
OFFSET_CODECS = (C_FULL, C_FIELDS_OFFSETS, C_OFFSETS_ONLY, C_FREQS_OFFSETS, C_FULL_WIDE, C_FIELDS_OFFSETS_WIDE)

R_UNION, R_INTERSECTION, R_TERM, R_VIRTUAL, R_NUMERIC, R_METRIC, R_HYBRID = 1, 2, 4, 8, 16, 32, 64


def build(native=False):
    """Compile oracle/*.c (gcc).  native=True also builds the -march=native flavour."""
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["native"] if native else []))
    return _SO


def _load():
    srcs = [os.path.join(_HERE, f) for f in ("flat_oracle.c", "scoring_oracle.c", "postings_oracle.c")]
    if (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        build()
    try:
        return C.CDLL(_SO)
    except OSError:
        build()
        return C.CDLL(_SO)


lib = _load()

REFERENCE_ROOT = "/root/reference"


def ref_lib(name):
    """oracle/_ref/<name>.so -- the reference's OWN code for the pieces of the path that compile from their own
    sources (oracle/ref_wrap.c, `make -C oracle ref`); None where neither a prebuilt file nor the reference tree exists."""
    path = os.path.join(_HERE, "_ref", name + ".so")
    if not os.path.exists(path) and os.path.isdir(os.path.join(REFERENCE_ROOT, "src")):
        subprocess.call(["make", "-s", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        return C.CDLL(path) if os.path.exists(path) else None
    except OSError:
        return None
_vp, _sz, _dbl, _i = C.c_void_p, C.c_size_t, C.c_double, C.c_int


def _sig(name, res, *args):
    f = getattr(lib, name)
    f.restype, f.argtypes = res, list(args)
    return f


_sig("oflat_new", _vp, _i, _sz, _i, _i, _sz)
_sig("oflat_free", None, _vp)
_sig("oflat_size", _sz, _vp)
_sig("oflat_add", _i, _vp, _vp, _sz)
_sig("oflat_add_bulk", None, _vp, _vp, _sz, _sz)
_sig("oflat_delete", _i, _vp, _sz)
_sig("oflat_distance_from", _dbl, _vp, _sz, _vp)
_sig("oflat_topk", _sz, _vp, _vp, _sz, _i, _vp, _vp)
_sig("oflat_topk_heap", _sz, _vp, _vp, _sz, _vp, _vp)
_sig("oflat_range", _sz, _vp, _vp, _dbl, _i, _vp, _vp)
_sig("obatch_new", _vp, _vp, _vp)
_sig("obatch_has_next", _i, _vp)
_sig("obatch_next", _sz, _vp, _sz, _i, _vp, _vp)
_sig("obatch_free", None, _vp)
_sig("oracle_normalize", None, _vp, _sz, _i)
_sig("oracle_blob_size", _sz, _i, _sz, _i)
_sig("oracle_distance", _dbl, _vp, _vp, _sz, _i, _i)
_sig("oracle_distance_f64", _dbl, _vp, _vp, _sz, _i, _i)
_sig("oracle_prefer_adhoc", _i, _sz, _sz, _sz, _sz, _i, C.POINTER(_i))
_sig("oracle_prefer_adhoc2", _i, _sz, _sz, _sz, _sz, _sz, _i, C.POINTER(_i))
_sig("oracle_f32_to_f16", C.c_uint16, C.c_float)
_sig("oracle_f32_to_bf16", C.c_uint16, C.c_float)
_sig("oracle_f16_to_f32", C.c_float, C.c_uint16)
_sig("oracle_bf16_to_f32", C.c_float, C.c_uint16)
_sig("oracle_idf", _dbl, _sz, _sz)
_sig("oracle_idf_bm25", _dbl, _sz, _sz)
_sig("oracle_bm25std_term", _dbl, _dbl, _dbl, _i, _dbl, _dbl)
_sig("oracle_hamming", _dbl, _vp, _sz, _vp, _sz)
_sig("oracle_qint_encode", _sz, _vp, _vp, _i)
_sig("oracle_qint_decode", _sz, _vp, _sz, _vp, _i)
_sig("oracle_varint_encode", _sz, _vp, C.c_uint64)
_sig("oracle_varint_decode", _sz, _vp, _sz, C.POINTER(C.c_uint64))
_sig("oinv_new", _vp, _i)
_sig("oinv_free", None, _vp)
_sig("oinv_add", _i, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _vp, C.c_uint32)
_sig("oreader_offsets", C.c_uint32, _vp, C.POINTER(C.c_void_p))
_sig("oracle_decode_offsets", _sz, _vp, _sz, _vp, _sz)
_sig("oinv_add_wide", _i, _vp, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, _vp, C.c_uint32)
_sig("oinv_decode_masks128", _sz, _vp, _vp, _vp)
_sig("oinv_add_many", None, _vp, _vp, _vp, _sz)
_sig("oinv_num_blocks", _sz, _vp)
_sig("oinv_unique_docs", C.c_uint32, _vp)
_sig("oinv_total_bytes", _sz, _vp)
_sig("oinv_flatten", None, _vp, _vp, _vp, _vp, _vp, _vp)
_sig("oinv_decode_all", _sz, _vp, _vp, _vp, _vp)
_sig("oreader_new", _vp, _vp)
_sig("oreader_free", None, _vp)
_sig("oreader_rewind", None, _vp)
_sig("oreader_next", _i, _vp)
_sig("oreader_seek", _i, _vp, C.c_uint64)
_sig("oreader_doc", C.c_uint64, _vp)
_sig("oreader_freq", C.c_uint32, _vp)
_sig("oreader_mask", C.c_uint32, _vp)
_sig("oracle_intersect", _sz, _vp, _sz, _sz, _vp, _vp, _vp)
_sig("oracle_decode_offsets", _sz, _vp, _sz, _vp, _sz)


def _p(a):
    return a.ctypes.data_as(_vp)


def to_blob(vec, vtype):
    """numpy vector -> element array of the index's type (bf16 as uint16 bit patterns)."""
    if vtype == BF16:
        f = np.ascontiguousarray(vec, dtype=np.float32)
        return np.array([lib.oracle_f32_to_bf16(float(x)) for x in f.ravel()], dtype=np.uint16).reshape(f.shape)
    return np.ascontiguousarray(vec, dtype=TYPE_NP[vtype])


class FlatIndex:
    """CPU restatement of a VecSim FLAT index (oracle/flat_oracle.c)."""

    def __init__(self, vtype, dim, metric, multi=False, block_size=1024):
        self.vtype, self.dim, self.metric, self.multi = vtype, dim, metric, multi
        self.h = lib.oflat_new(vtype, dim, metric, int(multi), block_size)
        if not self.h:
            raise ValueError("bad FLAT params")

    def __del__(self):
        if getattr(self, "h", None):
            lib.oflat_free(self.h)
            self.h = None

    def __len__(self):
        return lib.oflat_size(self.h)

    def _q(self, q):
        b = to_blob(q, self.vtype)
        assert b.size == self.dim
        return b

    def add(self, vec, label):
        return lib.oflat_add(self.h, _p(self._q(vec)), label)

    def add_bulk(self, mat, first_label=1):
        m = to_blob(mat, self.vtype)
        assert m.ndim == 2 and m.shape[1] == self.dim
        lib.oflat_add_bulk(self.h, _p(m), m.shape[0], first_label)

    def delete(self, label):
        return lib.oflat_delete(self.h, label)

    def normalized_query(self, q):
        """Query blob as hybrid_reader.c:295-305 prepares it for GetDistanceFrom_Unsafe."""
        nbytes = lib.oracle_blob_size(self.vtype, self.dim, self.metric)
        buf = np.zeros(nbytes, dtype=np.uint8)
        raw = self._q(q).view(np.uint8).ravel()
        buf[: raw.size] = raw
        if self.metric == COSINE:
            lib.oracle_normalize(_p(buf), self.dim, self.vtype)
        return buf

    def distance_from(self, label, nq):
        return lib.oflat_distance_from(self.h, label, _p(nq))

    def topk(self, q, k, order=BY_SCORE, heap=False):
        n = min(k, len(self))
        ids = np.zeros(max(n, 1), dtype=np.uint64)
        sc = np.zeros(max(n, 1), dtype=np.float64)
        qq = self._q(q)
        if heap:
            m = lib.oflat_topk_heap(self.h, _p(qq), k, _p(ids), _p(sc))
        else:
            m = lib.oflat_topk(self.h, _p(qq), k, order, _p(ids), _p(sc))
        return ids[:m].copy(), sc[:m].copy()

    def range(self, q, radius, order=BY_ID):
        n = max(len(self), 1)
        ids = np.zeros(n, dtype=np.uint64)
        sc = np.zeros(n, dtype=np.float64)
        m = lib.oflat_range(self.h, _p(self._q(q)), radius, order, _p(ids), _p(sc))
        return ids[:m].copy(), sc[:m].copy()

    def batches(self, q):
        return BatchIterator(self, q)


class BatchIterator:
    def __init__(self, index, q):
        self.index = index
        self.h = lib.obatch_new(index.h, _p(index._q(q)))

    def __del__(self):
        if getattr(self, "h", None):
            lib.obatch_free(self.h)
            self.h = None

    def has_next(self):
        return bool(lib.obatch_has_next(self.h))

    def next(self, n, order=BY_ID):
        cap = max(min(n, len(self.index)), 1)
        ids = np.zeros(cap, dtype=np.uint64)
        sc = np.zeros(cap, dtype=np.float64)
        m = lib.obatch_next(self.h, n, order, _p(ids), _p(sc))
        return ids[:m].copy(), sc[:m].copy()


_sig("oracle_philox_rows", None, C.c_uint64, C.c_uint64, _sz, _sz, _i, _vp)


def philox_rows(seed, first_index, n, dim, vtype=F32, threads=None):
    """Rows first_index .. first_index+n-1 of the keyed synthetic corpus (oracle_philox_rows), as an [n, dim] array;
    large requests are split over threads (the C function releases the GIL)."""
    out = np.empty((n, dim), dtype=TYPE_NP[vtype])
    if n == 0:
        return out
    threads = threads or (min(32, os.cpu_count() or 1) if n * dim > (1 << 22) else 1)
    if threads <= 1:
        lib.oracle_philox_rows(seed, first_index, n, dim, vtype, _p(out))
        return out
    from concurrent.futures import ThreadPoolExecutor
    step = (n + threads - 1) // threads

    def part(t):
        a, b = t * step, min(n, (t + 1) * step)
        if a < b:
            lib.oracle_philox_rows(seed, first_index + a, b - a, dim, vtype, _p(out[a:b]))
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(part, range(threads)))
    return out


def prefer_adhoc(index_size, dim, subset, k, initial_check=True, label_count=None):
    mode = _i(0)
    r = lib.oracle_prefer_adhoc2(index_size, index_size if label_count is None else label_count, dim, subset, k,
                                 int(initial_check), C.byref(mode))
    return bool(r), mode.value


# ---- scoring ---------------------------------------------------------------------------------------
class _ONode(C.Structure):
    pass


_ONode._fields_ = [("tag", _i), ("weight", _dbl), ("freq", C.c_uint32), ("has_term", _i), ("idf", _dbl),
                   ("bm25_idf", _dbl), ("offsets", C.POINTER(C.c_uint32)), ("n_offsets", _sz),
                   ("children", C.POINTER(C.POINTER(_ONode))), ("n_children", _sz)]


class _ODoc(C.Structure):
    _fields_ = [("score", C.c_float), ("max_term_freq", C.c_uint32), ("doc_len", C.c_uint32)]


class _OStats(C.Structure):
    _fields_ = [("num_docs", _sz), ("num_terms", _sz), ("avg_doc_len", _dbl), ("tanh_factor", C.c_uint64)]


for _n, _args in (("oracle_tfidf", (_vp, _vp, _dbl, _i)), ("oracle_bm25", (_vp, _vp, _vp, _dbl)),
                  ("oracle_bm25std", (_vp, _vp, _vp)), ("oracle_bm25std_tanh", (_vp, _vp, _vp)),
                  ("oracle_dismax", (_vp,))):
    _sig(_n, _dbl, *_args)
_sig("oracle_slop", _i, _vp)
_sig("oracle_max_normalize", None, _vp, _sz)
_sig("oracle_score_flat", None, _i, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _vp, _dbl, _i, _vp)


class Node:
    """Result-tree node mirroring RSIndexResult for the scorers."""

    def __init__(self, tag, weight=1.0, freq=0, idf=None, bm25_idf=0.0, offsets=None, children=()):
        self.c = _ONode()
        self.c.tag, self.c.weight, self.c.freq = tag, weight, freq
        self.c.has_term = int(idf is not None)
        self.c.idf = idf or 0.0
        self.c.bm25_idf = bm25_idf
        self._off = np.asarray(offsets if offsets is not None else [], dtype=np.uint32)
        self.c.offsets = self._off.ctypes.data_as(C.POINTER(C.c_uint32))
        self.c.n_offsets = self._off.size
        self.kids = list(children)
        self._arr = (C.POINTER(_ONode) * max(len(self.kids), 1))(*[C.pointer(k.c) for k in self.kids])
        self.c.children = C.cast(self._arr, C.POINTER(C.POINTER(_ONode)))
        self.c.n_children = len(self.kids)

    @property
    def ptr(self):
        return C.addressof(self.c)


def term(freq, idf=0.0, bm25_idf=0.0, weight=1.0, offsets=None):
    return Node(R_TERM, weight, freq, idf, bm25_idf, offsets)


def intersection(children, weight=1.0):
    return Node(R_INTERSECTION, weight, sum(k.c.freq for k in children), children=children)


def union(children, weight=1.0):
    return Node(R_UNION, weight, sum(k.c.freq for k in children), children=children)


def _doc(score, max_freq, doc_len):
    return _ODoc(score, max_freq, doc_len)


def _stats(num_docs, avg_doc_len, tanh_factor=4, num_terms=0):
    return _OStats(num_docs, num_terms, avg_doc_len, tanh_factor)


def score(scorer, node, doc_score=1.0, max_freq=1, doc_len=1, num_docs=1, avg_doc_len=1.0, min_score=0.0,
          tanh_factor=4):
    d, st = _doc(doc_score, max_freq, doc_len), _stats(num_docs, avg_doc_len, tanh_factor)
    dp, sp = C.addressof(d), C.addressof(st)
    if scorer == "TFIDF":
        return lib.oracle_tfidf(node.ptr, dp, min_score, 1)
    if scorer == "TFIDF.DOCNORM":
        return lib.oracle_tfidf(node.ptr, dp, min_score, 2)
    if scorer == "BM25":
        return lib.oracle_bm25(sp, node.ptr, dp, min_score)
    if scorer == "BM25STD":
        return lib.oracle_bm25std(sp, node.ptr, dp)
    if scorer == "BM25STD.TANH":
        return lib.oracle_bm25std_tanh(sp, node.ptr, dp)
    if scorer == "DISMAX":
        return lib.oracle_dismax(node.ptr)
    if scorer == "DOCSCORE":
        return float(d.score)
    raise ValueError(scorer)


SCORER_IDS = {"BM25STD": 0, "BM25STD.TANH": 1, "BM25": 2, "TFIDF": 3, "TFIDF.DOCNORM": 4, "DOCSCORE": 5, "DISMAX": 6}


def score_flat(scorer, freq, doc_len, max_freq, doc_score, idf, bm25_idf, weight, root_weight, num_docs,
               avg_doc_len, min_score=0.0, tanh_factor=4):
    """SoA scoring loop over an N-term intersection: freq is [T, M]."""
    freq = np.ascontiguousarray(freq, dtype=np.uint32)
    T, M = freq.shape
    doc_len = np.ascontiguousarray(doc_len, dtype=np.uint32)
    max_freq = np.ascontiguousarray(max_freq, dtype=np.uint32)
    doc_score = np.ascontiguousarray(doc_score, dtype=np.float32)
    idf = np.ascontiguousarray(idf, dtype=np.float64)
    bm25_idf = np.ascontiguousarray(bm25_idf, dtype=np.float64)
    weight = np.ascontiguousarray(weight, dtype=np.float64)
    st = _stats(num_docs, avg_doc_len, tanh_factor)
    out = np.zeros(M, dtype=np.float64)
    slop_const = 1 if T <= 1 else T - 1
    lib.oracle_score_flat(SCORER_IDS[scorer], M, T, _p(freq), _p(doc_len), _p(max_freq), _p(doc_score), _p(idf),
                          _p(bm25_idf), _p(weight), root_weight, C.addressof(st), min_score, slop_const, _p(out))
    return out


def max_normalize(scores):
    """BM25STD.NORM epilogue (RPMaxScoreNormalizer): a copy of `scores` divided by max(0, max(scores)) unless that is 0."""
    out = np.array(scores, dtype=np.float64, copy=True)
    lib.oracle_max_normalize(_p(out), out.size)
    return out


# ---- postings --------------------------------------------------------------------------------------
class InvertedIndex:
    """Posting list in the reference's block format (oracle/postings_oracle.c)."""

    def __init__(self, codec):
        self.codec = codec
        self.h = lib.oinv_new(codec)

    def __del__(self):
        if getattr(self, "h", None):
            lib.oinv_free(self.h)
            self.h = None

    def add(self, doc, freq=1, mask=1, offsets=b""):
        ob = np.frombuffer(bytes(offsets), dtype=np.uint8) if len(offsets) else np.zeros(1, dtype=np.uint8)
        return lib.oinv_add(self.h, doc, freq, mask, _p(ob), len(offsets))

    def add_wide(self, doc, freq=1, mask=1, offsets=b""):
        """mask: a Python int of up to 128 bits"""
        ob = np.frombuffer(bytes(offsets), dtype=np.uint8) if len(offsets) else np.zeros(1, dtype=np.uint8)
        return lib.oinv_add_wide(self.h, doc, freq, mask & 0xFFFFFFFFFFFFFFFF, mask >> 64, _p(ob), len(offsets))

    def decode_masks128(self):
        n = self.unique_docs
        lo, hi = np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint64)
        m = lib.oinv_decode_masks128(self.h, _p(lo), _p(hi))
        return [int(a) | (int(b) << 64) for a, b in zip(lo[:m].tolist(), hi[:m].tolist())]

    def add_many(self, docs, freqs=None):
        d = np.ascontiguousarray(docs, dtype=np.uint64)
        f = np.ascontiguousarray(freqs, dtype=np.uint32) if freqs is not None else None
        lib.oinv_add_many(self.h, _p(d), _p(f) if f is not None else None, d.size)

    @property
    def unique_docs(self):
        return lib.oinv_unique_docs(self.h)

    @property
    def num_blocks(self):
        return lib.oinv_num_blocks(self.h)

    def flatten(self):
        """-> dict(first,last,num_entries,offset,bytes): the device upload format."""
        nb = self.num_blocks
        first, last = np.zeros(nb, np.uint64), np.zeros(nb, np.uint64)
        nent, off = np.zeros(nb, np.uint32), np.zeros(nb + 1, np.uint64)
        data = np.zeros(max(lib.oinv_total_bytes(self.h), 1), np.uint8)
        lib.oinv_flatten(self.h, _p(first), _p(last), _p(nent), _p(off), _p(data))
        return dict(first=first, last=last, num_entries=nent, offset=off, bytes=data[: int(off[nb])], codec=self.codec)

    def decode_all(self):
        n = self.unique_docs
        ids, fr, mk = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        m = lib.oinv_decode_all(self.h, _p(ids), _p(fr), _p(mk))
        return ids[:m], fr[:m], mk[:m]

    def reader(self):
        return Reader(self)


class Reader:
    def __init__(self, ii):
        self.ii = ii
        self.h = lib.oreader_new(ii.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib.oreader_free(self.h)
            self.h = None

    def next(self):
        return (lib.oreader_doc(self.h), lib.oreader_freq(self.h)) if lib.oreader_next(self.h) else None

    def seek(self, target):
        return (lib.oreader_doc(self.h), lib.oreader_freq(self.h)) if lib.oreader_seek(self.h, target) else None

    def rewind(self):
        lib.oreader_rewind(self.h)

    def offsets(self):
        """absolute token positions of the current record"""
        pp = C.c_void_p()
        n = lib.oreader_offsets(self.h, C.byref(pp))
        if not n:
            return []
        buf = (C.c_uint8 * n).from_address(pp.value)
        out = np.zeros(n, np.uint32)
        m = lib.oracle_decode_offsets(buf, n, _p(out), n)
        return out[:m].tolist()


def intersect(lists, cap=None):
    """N-way AND of InvertedIndex objects -> (ids[H], freqs[N,H], masks[N,H])."""
    n = len(lists)
    cap = cap if cap is not None else max(min(l.unique_docs for l in lists), 1)
    arr = (_vp * n)(*[l.h for l in lists])
    ids = np.zeros(cap, np.uint64)
    fr, mk = np.zeros((n, cap), np.uint32), np.zeros((n, cap), np.uint32)
    h = lib.oracle_intersect(C.cast(arr, _vp), n, cap, _p(ids), _p(fr), _p(mk))
    return ids[:h].copy(), fr[:, :h].copy(), mk[:, :h].copy()


_sig("oracle_intersect_ex", _sz, _vp, _sz, _sz, C.c_long, _i, _vp, _vp, _vp, _vp)
_sig("oracle_within_range", _i, _sz, _vp, _vp, _vp, _vp, C.c_long, _i)
_sig("oracle_min_offset_delta", _i, _sz, _vp, _vp, _vp, _vp)


def intersect_ex(lists, max_slop=None, in_order=False, cap=None):
    """Intersection with max_slop / in_order (oracle_intersect_ex) -> (ids[H], freqs[N,H], slops[H])."""
    n = len(lists)
    cap = cap or max(min(l.unique_docs for l in lists), 1)
    arr = (_vp * n)(*[l.h for l in lists])
    ids, fr, sl = np.zeros(cap, np.uint64), np.zeros((n, cap), np.uint32), np.zeros(cap, np.int32)
    m = lib.oracle_intersect_ex(arr, n, cap, -1 if max_slop is None else int(max_slop), int(in_order), _p(ids), _p(fr),
                                None, _p(sl))
    return ids[:m], fr[:, :m], sl[:m]


def _prox_args(children):
    """children: list of (is_agg, [offset byte strings of the leaves])"""
    first, agg, bufs, lens = [0], [], [], []
    for is_agg, leaves in children:
        agg.append(int(is_agg))
        for b in leaves:
            bufs.append(np.frombuffer(bytes(b), np.uint8) if len(b) else np.zeros(1, np.uint8))
            lens.append(len(b))
        first.append(len(bufs))
    ptrs = (_vp * max(len(bufs), 1))(*[x.ctypes.data for x in bufs])
    return (len(children), _p(np.asarray(first, np.uint64)), _p(np.asarray(agg, np.int32)), ptrs,
            _p(np.asarray(lens + [0], np.uint32))), bufs


def within_range(children, max_slop=None, in_order=False):
    a, keep = _prox_args(children)
    return bool(lib.oracle_within_range(*a, -1 if max_slop is None else int(max_slop), int(in_order)))


def min_offset_delta(children):
    a, keep = _prox_args(children)
    return lib.oracle_min_offset_delta(*a)


def qint_encode(vals):
    v = np.asarray(vals, dtype=np.uint32)
    out = np.zeros(17, np.uint8)
    k = lib.oracle_qint_encode(_p(out), _p(v), len(v))
    return bytes(out[:k])


def qint_decode(buf, n):
    b = np.frombuffer(bytes(buf), dtype=np.uint8)
    v = np.zeros(n, np.uint32)
    k = lib.oracle_qint_decode(_p(b), len(b), _p(v), n)
    return v.tolist(), k


def varint_encode(v):
    out = np.zeros(16, np.uint8)
    k = lib.oracle_varint_encode(_p(out), v)
    return bytes(out[:k])


def varint_decode(buf):
    b = np.frombuffer(bytes(buf), dtype=np.uint8)
    v = C.c_uint64(0)
    k = lib.oracle_varint_decode(_p(b), len(b), C.byref(v))
    return v.value, k


_sig("oracle_vector_norm", _dbl, _i, _dbl)
_sig("oracle_hybrid_fuse", _sz, _i, _dbl, _dbl, _dbl, _i, _vp, _vp, _sz, _vp, _vp, _sz, _sz, _vp, _vp)
RRF, LINEAR = 0, 1


def vector_norm(metric, d):
    return lib.oracle_vector_norm(metric, d)


def hybrid_fuse(scoring, a_ids, a_scores, b_ids, b_scores, window, constant=60.0, weights=(0.5, 0.5), metric=-1):
    """FT.HYBRID merge of a search list and a vector list (oracle/scoring_oracle.c oracle_hybrid_fuse)."""
    a_ids, b_ids = np.ascontiguousarray(a_ids, np.uint64), np.ascontiguousarray(b_ids, np.uint64)
    a_scores, b_scores = np.ascontiguousarray(a_scores, np.float64), np.ascontiguousarray(b_scores, np.float64)
    ids = np.zeros(len(a_ids) + len(b_ids) + 1, np.uint64)
    sc = np.zeros(len(a_ids) + len(b_ids) + 1, np.float64)
    m = lib.oracle_hybrid_fuse(scoring, constant, weights[0], weights[1], metric, _p(a_ids), _p(a_scores), len(a_ids),
                               _p(b_ids), _p(b_scores), len(b_ids), window, _p(ids), _p(sc))
    return ids[:m].copy(), sc[:m].copy()


_sig("oracle_union", _sz, _vp, _sz, _sz, _vp, _vp, _vp)
_sig("oracle_not", _sz, _vp, _vp, C.c_uint64, _sz, _vp)


_sig("oracle_intersection_sort_key", _dbl, _sz, _i, _sz, _i)
_sig("oracle_intersection_child_order", None, _sz, _vp, _vp, _vp, _i, _i, _vp)
K_TERM, K_UNION, K_INTERSECTION = 0, 1, 2


def intersection_sort_key(num_estimated, kind=K_TERM, n_children=1, prioritize_union_children=False):
    """num_estimated x intersection_sort_weight (oracle_intersection_sort_key)"""
    return lib.oracle_intersection_sort_key(int(num_estimated), int(kind), int(n_children), int(prioritize_union_children))


def intersection_child_order(children, prioritize_union_children=False, in_order=False):
    """children: [(num_estimated, kind, n_children)] in query order -> their indices in iteration (= result) order"""
    n = len(children)
    est = np.asarray([c[0] for c in children], np.uint64)
    kind = np.asarray([c[1] for c in children], np.int32)
    nch = np.asarray([c[2] for c in children], np.uint64)
    out = np.zeros(max(n, 1), np.uint64)
    lib.oracle_intersection_child_order(n, _p(est), _p(kind), _p(nch), int(prioritize_union_children), int(in_order), _p(out))
    return out[:n].astype(int).tolist()


def union_lists(lists):
    """N-way OR of InvertedIndex objects -> (ids[H], freqs[N,H], masks[N,H]); absent = 0."""
    n = len(lists)
    cap = max(sum(l.unique_docs for l in lists), 1)
    arr = (_vp * n)(*[l.h for l in lists])
    ids = np.zeros(cap, np.uint64)
    fr, mk = np.zeros((n, cap), np.uint32), np.zeros((n, cap), np.uint32)
    h = lib.oracle_union(C.cast(arr, _vp), n, cap, _p(ids), _p(fr), _p(mk))
    return ids[:h].copy(), fr[:, :h].copy(), mk[:, :h].copy()


def not_list(child, max_doc_id, universe=None):
    """Doc ids in 1..max_doc_id (or in `universe`) that `child` does not hold."""
    cap = max(universe.unique_docs if universe is not None else max_doc_id, 1)
    ids = np.zeros(cap, np.uint64)
    h = lib.oracle_not(child.h, universe.h if universe is not None else None, max_doc_id, cap, _p(ids))
    return ids[:h].copy()
