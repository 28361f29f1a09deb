"""CPU: the HOST half of Boundary 3.  redisearch_amd/csrc/query_iterators.c is compiled against tests/mock_hits.c -- a stand-in
for the device hit lists that serves arrays prepared from the CPU oracle -- so that everything the iterator library does on
the host runs without a GPU: Read / SkipTo / Rewind / NumEstimated semantics (reference src/iterators/iterator_api.h:86-141,
rqe_iterators/src/intersection.rs:428-530), lazy block paging of the per-term records, the offsets byte ranges, and the
result trees it rebuilds through the module's constructors (oracle/ext_harness.c plays the module) -- flat AND / OR / NOT
and two-level trees -- which are then scored by a scorer extension and held to the oracle's result-tree scorers.
The device half (the hit lists themselves) is covered by tests/test_gpu_iterators.py."""
import ctypes as C
import os
import subprocess

import zlib

import numpy as np
import pytest

import oracle as O
import oracle.ext as X
from tests import intersection_cases as IC
from tests.test_gpu_iterators import ALL, HAS_FREQ, HAS_MASK, WIDE, check_drain, model_script, records, size_order
from tests.test_gpu_proximity import rand_list
from tests.test_gpu_tree import I as OP_I, T as OP_T, U as OP_U, OracleTree, rand_list as tree_rand_list

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_build")
LIB = os.path.join(BUILD, "libiter_mock.so")
MAXL = 32
_vp = C.c_void_p


class MockPostings(C.Structure):
    _fields_ = [("codec", C.c_int), ("n_entries", C.c_size_t), ("bytes", _vp), ("n_bytes", C.c_size_t)]


class MockHits(C.Structure):
    _fields_ = [("len", C.c_size_t), ("n_leaves", C.c_int), ("is_union", C.c_int), ("order", C.c_int * MAXL), ("ids", _vp),
                ("entry", _vp * MAXL), ("freq", _vp * MAXL), ("olen", _vp * MAXL),
                ("mlo", _vp * MAXL), ("mhi", _vp * MAXL), ("opos", _vp * MAXL),
                ("n_groups", C.c_int), ("group_first", C.c_int * (MAXL + 1)), ("group_op", C.c_int * MAXL),
                ("group_weight", C.c_double * MAXL), ("freed", C.c_int)]


class TermArg(C.Structure):
    _fields_ = [("postings", _vp), ("term", _vp), ("weight", C.c_double)]


class TreeQuery(C.Structure):
    _fields_ = [("root_op", C.c_int), ("n_groups", C.c_size_t), ("group_first", _vp), ("group_op", _vp), ("group_weight", _vp),
                ("lists", _vp), ("max_slop", C.c_long), ("in_order", C.c_int)]


@pytest.fixture(scope="module")
def lib():
    os.makedirs(BUILD, exist_ok=True)
    srcs = [os.path.join(ROOT, "redisearch_amd", "csrc", "query_iterators.c"), os.path.join(ROOT, "tests", "mock_hits.c")]
    prebuilt = os.environ.get("RSGPU_ITER_MOCK_LIB")        # e.g. an -fsanitize=address,undefined build (scripts/asan_iterators.sh)
    if prebuilt:
        pass
    elif not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu11", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wextra",
                               "-I" + os.path.join(ROOT, "include")] + srcs + ["-o", LIB, "-ldl",
                               "-Wl,-Bsymbolic"])           # (binds the mock's RSGPU_* inside: the real engine may be loaded too)
    L = C.CDLL(prebuilt or LIB)
    L.RSGPU_Iterators_SetResultAPI.restype, L.RSGPU_Iterators_SetResultAPI.argtypes = C.c_int, [_vp, _vp]
    L.RSGPU_Iterators_SetBlock.restype, L.RSGPU_Iterators_SetBlock.argtypes = None, [C.c_size_t]
    L.RSGPU_Iterators_LastError.restype = C.c_char_p
    L.RSGPU_NewIntersectionIterator.restype = _vp
    L.RSGPU_NewIntersectionIterator.argtypes = [_vp, C.c_size_t, C.c_int32, C.c_bool, C.c_double]
    L.RSGPU_NewUnionIterator.restype, L.RSGPU_NewUnionIterator.argtypes = _vp, [_vp, C.c_size_t, C.c_double]
    L.RSGPU_NewNotIterator.restype, L.RSGPU_NewNotIterator.argtypes = _vp, [_vp, _vp, C.c_uint64, C.c_double]
    L.RSGPU_NewTreeIterator.restype, L.RSGPU_NewTreeIterator.argtypes = _vp, [C.POINTER(TreeQuery), _vp, C.c_double]
    L.RSGPU_NewHitsIterator.restype, L.RSGPU_NewHitsIterator.argtypes = _vp, [_vp, _vp, C.c_size_t, C.c_double, C.c_bool]
    L.RSGPU_Iterator_Hits.restype, L.RSGPU_Iterator_Hits.argtypes = _vp, [_vp]
    L.mock_set_next_hits.restype, L.mock_set_next_hits.argtypes = None, [_vp]
    L.mock_counters.restype, L.mock_counters.argtypes = None, [C.POINTER(C.c_int)] * 3
    assert L.RSGPU_Iterators_SetResultAPI(None, _vp(X.handle())) == 0, L.RSGPU_Iterators_LastError()
    return L


class Mock:
    """Hit list + posting lists as the device would serve them, from the oracle's view of `lists`."""

    def __init__(self, lists, ids, slots, is_union=False, present=None, groups=None):
        """slots: list index of every child slot; present(slot_list, doc) -> bool (default: the list holds the doc)"""
        self.keep = []
        self.recs = [records(l) for l in lists]
        self.posts = []
        for l, rec in zip(lists, self.recs):
            blob, where = b"", {}
            for doc in sorted(rec):                       # the offsets blobs of a list, in doc-id order
                enc, last = b"", 0
                for p in rec[doc][2]:
                    enc += O.varint_encode(p - last)
                    last = p
                where[doc] = (len(blob), len(enc))
                blob += enc
            buf = np.frombuffer(blob + b"\0", np.uint8).copy()
            self.keep.append(buf)
            mp = MockPostings(l.codec, l.unique_docs, buf.ctypes.data, len(blob))
            self.posts.append((mp, where, {d: i for i, d in enumerate(sorted(rec))}))
        h = MockHits()
        h.len, h.n_leaves, h.is_union = len(ids), len(slots), int(is_union)
        idarr = np.asarray(ids, np.uint64)
        self.keep.append(idarr)
        h.ids = idarr.ctypes.data
        for s, li in enumerate(slots):
            h.order[s] = li
            rec, (mp, where, index) = self.recs[li], self.posts[li]
            n = max(len(ids), 1)
            e, f, ol = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
            lo, hi, op = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64)
            for j, d in enumerate(ids):
                here = d in rec and (present is None or present(li, d))
                if not here:
                    e[j] = 0xFFFFFFFF
                    continue
                e[j] = index[d]
                f[j] = rec[d][0]                           # (1 where the codec stores none, like the device column)
                m = rec[d][1] if lists[li].codec in HAS_MASK else 0
                lo[j], hi[j] = m & 0xFFFFFFFFFFFFFFFF, m >> 64
                op[j], ol[j] = where[d]
            self.keep += [e, f, ol, lo, hi, op]
            h.entry[s], h.freq[s], h.olen[s] = e.ctypes.data, f.ctypes.data, ol.ctypes.data
            h.mlo[s], h.mhi[s], h.opos[s] = lo.ctypes.data, hi.ctypes.data, op.ctypes.data
        if groups is None:
            h.n_groups = len(slots)
            for g in range(len(slots) + 1):
                h.group_first[g] = g
        else:
            h.n_groups = len(groups)
            at = 0
            for g, (op_, w, n_leaves) in enumerate(groups):
                h.group_first[g], h.group_op[g], h.group_weight[g] = at, op_, w
                at += n_leaves
            h.group_first[len(groups)] = at
        self.hits = h

    def term_args(self, terms=None, weights=None):
        arr = (TermArg * len(self.posts))()
        for i, (mp, _, _) in enumerate(self.posts):
            arr[i].postings = C.addressof(mp)
            arr[i].term = terms[i] if terms else None
            arr[i].weight = weights[i] if weights else 1.0
        self.keep.append(arr)
        return arr


def make(lib, kind, mock, block=65536, weight=1.0, terms=None, weights=None, tree=None, max_doc=0):
    lib.RSGPU_Iterators_SetBlock(block)
    lib.mock_set_next_hits(C.addressof(mock.hits))
    try:
        if kind == "and":
            it = lib.RSGPU_NewIntersectionIterator(C.cast(mock.term_args(terms, weights), _vp), len(mock.posts), -1, False, weight)
        elif kind == "or":
            it = lib.RSGPU_NewUnionIterator(C.cast(mock.term_args(terms, weights), _vp), len(mock.posts), weight)
        elif kind == "not":
            it = lib.RSGPU_NewNotIterator(C.addressof(mock.posts[0][0]), None, max_doc, weight)
        else:
            it = lib.RSGPU_NewTreeIterator(C.byref(tree), C.cast(mock.term_args(terms, weights), _vp), weight)
    finally:
        lib.RSGPU_Iterators_SetBlock(65536)
    assert it, lib.RSGPU_Iterators_LastError()
    return it


def counters(lib):
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    lib.mock_counters(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


@pytest.mark.parametrize("codec", [O.C_FULL, O.C_FREQS_ONLY, O.C_DOCIDS_ONLY, O.C_FIELDS_OFFSETS, O.C_FULL_WIDE])
@pytest.mark.parametrize("block", [5, 65536])
def test_intersection_host_logic(lib, codec, block):
    rng = np.random.default_rng(700 + codec)
    lists = [rand_list(rng, codec, n, 900, wide=codec in WIDE) for n in (700, 400, 650)]
    want = O.intersect(lists)[0].tolist()
    assert len(want) > 20
    order = size_order(lists)
    m = Mock(lists, want, order)
    it = make(lib, "and", m, block=block, weight=2.5)
    try:
        assert X.iter_script(it, [(X.OP_ESTIMATE, 0)])[0][0] == min(l.unique_docs for l in lists)
        counters(lib)
        check_drain(it, lists, order, want)
        _, rec_reads, _ = counters(lib)
        n_blocks = -(-len(want) // block)
        assert rec_reads == n_blocks * 3                       # every block paged once per child, nothing re-read
        assert X.iter_script(it, [(X.OP_READ, 0)]) == [(2, want[-1], True, 0)]
        assert X.iter_script(it, [(X.OP_REWIND, 0)]) == [(0, 0, False, 0)]
        check_drain(it, lists, order, want)
        # a SkipTo pages the one block it lands in (the drained iterator still holds the LAST block)
        counters(lib)
        assert X.iter_script(it, [(X.OP_REWIND, 0), (X.OP_SKIP, want[1])])[1] == (0, want[1], False, want[1])
        assert counters(lib)[1] == (3 if n_blocks > 1 else 0)
    finally:
        X.iter_free(it)
    assert m.hits.freed == 1                                   # the iterator owned the hit list it made


@pytest.mark.parametrize("num_children", IC.NUM_CHILDREN_CASES)
@pytest.mark.parametrize("case", range(len(IC.RESULT_SET_CASES)))
def test_reference_fixture_scripts_on_the_host(lib, num_children, case):
    """rqe_iterators/tests/integration/intersection.rs fixtures (2 / 5 / 25 children): read / skip_to / rewind scripts"""
    result = IC.RESULT_SET_CASES[case]
    lists = [IC.to_index(k) for k in IC.create_children(num_children, result)]
    m = Mock(lists, result, size_order(lists))
    it = make(lib, "and", m, block=4)
    try:
        ops = []
        for d in sorted({t for r in result for t in (r - 1, r, r + 1) if t > 0} | {max(result) + 2}):
            ops += [(X.OP_REWIND, 0), (X.OP_SKIP, d), (X.OP_READ, 0)]
        ops += [(X.OP_REWIND, 0)] + [(X.OP_READ, 0)] * (len(result) + 2)
        ops += [(X.OP_REWIND, 0), (X.OP_READ, 0), (X.OP_SKIP, result[2]), (X.OP_SKIP, result[2] + 1), (X.OP_READ, 0),
                (X.OP_SKIP, result[-1]), (X.OP_READ, 0), (X.OP_READ, 0)]
        assert X.iter_script(it, ops) == model_script(result, ops)
    finally:
        X.iter_free(it)


def test_union_and_not_host_logic(lib):
    rng = np.random.default_rng(41)
    lists = [rand_list(rng, c, n, 700) for c, n in ((O.C_FULL, 200), (O.C_FREQS_ONLY, 350), (O.C_DOCIDS_ONLY, 120))]
    want = O.union_lists(lists)[0].tolist()
    m = Mock(lists, want, [0, 1, 2], is_union=True)
    it = make(lib, "or", m, block=50, weight=0.5)
    try:
        assert X.iter_script(it, [(X.OP_ESTIMATE, 0)])[0][0] == sum(l.unique_docs for l in lists)
        check_drain(it, lists, [0, 1, 2], want, union=True)
        ops = [(X.OP_REWIND, 0), (X.OP_SKIP, want[5]), (X.OP_SKIP, want[9] + 1), (X.OP_READ, 0), (X.OP_SKIP, want[-1] + 1)]
        assert X.iter_script(it, ops) == model_script(want, ops)
    finally:
        X.iter_free(it)
    child = lists[2]
    want = O.not_list(child, 620).tolist()
    m = Mock([child], want, [0], present=lambda li, d: False)
    it = make(lib, "not", m, weight=3.0, max_doc=620)
    try:
        assert X.iter_script(it, [(X.OP_ESTIMATE, 0)])[0][0] == 620
        counters(lib)
        d = X.iter_drain(it, len(want) + 4, 1)
        assert d["ids"].tolist() == want and set(d["n_children"].tolist()) == {0} and set(d["mask"]) == {ALL}
        assert counters(lib)[1] == 0                              # a NOT never pages term records
    finally:
        X.iter_free(it)


@pytest.mark.parametrize("name,root,shape", [
    ("and_of_ors", OP_I, [(OP_U, 1.0, [0, 1, 2]), (OP_U, 0.5, [3, 4])]),
    ("or_of_ands", OP_U, [(OP_I, 1.0, [0, 1]), (OP_I, 2.0, [2, 3])]),
    ("term_and_or", OP_I, [(OP_T, 1.0, [0]), (OP_U, 1.0, [1, 2, 3]), (OP_T, 1.0, [4])]),
    ("or_of_term_and_and", OP_U, [(OP_T, 1.0, [0]), (OP_I, 0.7, [1, 2, 3])]),
])
def test_tree_results_score_like_the_oracle_result_trees(lib, name, root, shape):
    """The nested `current` the tree iterator rebuilds, scored per result by a scorer extension (the reference's compiled
    default.c where oracle/_ref has it, else the product's plugin), equals the oracle's result-tree scorers on the trees the
    reference would have built (tests/test_gpu_tree.py::OracleTree)."""
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000 + 3)
    n_lists = sum(len(g[2]) for g in shape)
    built = [tree_rand_list(rng, O.C_FULL, int(rng.integers(300, 1500)), 2500, True) for _ in range(n_lists)]
    lists, recs = [b[0] for b in built], [b[1] for b in built]
    sizes = [l.unique_docs for l in lists]
    ot = OracleTree(root, shape, recs, sizes)
    assert len(ot.docs) > 5
    group_docs = {i: gr["docs"] for gr in ot.groups for i in gr["idx"]}
    m = Mock(lists, ot.docs, ot.leaf_order, is_union=root == OP_U, present=lambda li, d: d in group_docs[li],
             groups=[(gr["op"], gr["w"], len(gr["idx"])) for gr in ot.groups])
    n_docs = 2500
    doc_len = rng.integers(5, 200, n_docs + 1).astype(np.uint32)
    doc_score = rng.choice([1.0, 0.5], n_docs + 1).astype(np.float32)
    max_freq = rng.integers(1, 40, n_docs + 1).astype(np.uint32)
    idf = [O.lib.oracle_idf(n_docs, s) for s in sizes]  # (src/ext/default.c / query.c idf restated)
    bidf = [O.lib.oracle_idf_bm25(n_docs, s) for s in sizes]
    w = [float(x) for x in rng.choice([1.0, 0.5, 2.0], n_lists)]
    avg = float(doc_len[1:].mean())
    host = X.Host()
    use_ref = X.have_ref()
    host.load_ref() if use_ref else host.load_plugin()
    # the tree as the caller describes it (groups in QUERY order; the device / the mock re-orders them)
    flat_first, at = [0], 0
    for _, _, idx in shape:
        at += len(idx)
        flat_first.append(at)
    gf, go = np.asarray(flat_first, np.uint64), np.asarray([g[0] for g in shape], np.int32)
    gw = np.asarray([g[1] for g in shape], np.float64)
    tq = TreeQuery(root, len(shape), gf.ctypes.data, go.ctypes.data, gw.ctypes.data, None, -1, 0)
    for scorer in ("BM25STD", "TFIDF", "DISMAX", "BM25"):
        terms = [X.new_term(idf[i], bidf[i], "t%d" % i) for i in range(n_lists)]
        it = make(lib, "tree", m, block=23, weight=1.5, terms=terms, weights=w, tree=tq)
        try:
            ids, sc = X.iter_score_all(it, scorer, doc_len, doc_score, max_freq, len(ot.docs) + 4, num_docs=n_docs, avg_doc_len=avg,
                                       slop=0 if use_ref else 1)
            assert ids.tolist() == ot.docs
            if use_ref:                                            # (the plugin was given a fixed slop: no reference GetSlop)
                for j in rng.choice(len(ot.docs), min(40, len(ot.docs)), replace=False):
                    d = ot.docs[j]
                    node = ot.node(d, idf, bidf, w)
                    node.c.weight = 1.5
                    want = O.score(scorer, node, float(doc_score[d]), int(max_freq[d]), int(doc_len[d]), n_docs, avg)
                    assert sc[j] == want, (scorer, d)
        finally:
            X.iter_free(it)


def test_constructor_errors_and_foreign_iterators(lib):
    assert lib.RSGPU_NewIntersectionIterator(None, 0, -1, False, 1.0) is None
    assert b"terms" in lib.RSGPU_Iterators_LastError()
    lib.mock_set_next_hits(None)                                   # the device evaluation fails: NULL, terms untouched
    rng = np.random.default_rng(1)
    lists = [rand_list(rng, O.C_FREQS_ONLY, 50, 200) for _ in range(2)]
    m = Mock(lists, [], [0, 1])
    assert lib.RSGPU_NewIntersectionIterator(C.cast(m.term_args(), _vp), 2, -1, False, 1.0) is None
    assert b"RSGPU_IntersectEx" in lib.RSGPU_Iterators_LastError()
    assert lib.RSGPU_Iterator_Hits(None) is None
    # an empty hit list is at EOF from the first read
    it = make(lib, "and", m)
    try:
        assert X.iter_script(it, [(X.OP_READ, 0), (X.OP_REWIND, 0), (X.OP_SKIP, 3)]) == [(2, 0, True, 0), (0, 0, False, 0), (2, 0, True, 0)]
        assert lib.RSGPU_Iterator_Hits(it) == C.addressof(m.hits)
    finally:
        X.iter_free(it)


def test_module_allocation_failures_neither_crash_nor_leak(lib):
    """Every RSIndexResult constructor the library calls is made to fail once (oracle/ext_harness.c fault injection): the
    constructor returns NULL with a message, the hit list it made is released exactly once, nothing is used after free
    (run under scripts/asan_iterators.sh) -- for flat AND / NOT and for a two-level tree."""
    rng = np.random.default_rng(2)
    lists = [rand_list(rng, O.C_FULL, 80, 200) for _ in range(3)]
    want = O.intersect(lists)[0].tolist()
    try:
        for n in range(4):                                          # root + 3 term records
            m = Mock(lists, want, size_order(lists))
            terms = [X.new_term(1.0, 1.0, "t%d" % i) for i in range(3)]
            lib.mock_set_next_hits(C.addressof(m.hits))
            X.fail_constructor_after(n)
            it = lib.RSGPU_NewIntersectionIterator(C.cast(m.term_args(terms), _vp), 3, -1, False, 1.0)
            assert it is None and b"allocate" in lib.RSGPU_Iterators_LastError(), n
            assert m.hits.freed == 1, n
        m = Mock([lists[0]], [1, 2, 3], [0], present=lambda li, d: False)
        lib.mock_set_next_hits(C.addressof(m.hits))
        X.fail_constructor_after(0)
        assert lib.RSGPU_NewNotIterator(C.addressof(m.posts[0][0]), None, 10, 1.0) is None and m.hits.freed == 1
        # tree: make() builds the 3 term records, then one aggregate per non-term node (the union group, the root)
        ids = sorted(set(records(lists[0])) & (set(records(lists[1])) | set(records(lists[2]))))
        gf, go = np.asarray([0, 1, 3], np.uint64), np.asarray([OP_T, OP_U], np.int32)
        tq = TreeQuery(OP_I, 2, gf.ctypes.data, go.ctypes.data, None, None, -1, 0)
        for n in range(5):
            m = Mock(lists, ids, [0, 1, 2], groups=[(OP_T, 1.0, 1), (OP_U, 1.0, 2)])
            terms = [X.new_term(1.0, 1.0, "t%d" % i) for i in range(3)]
            lib.mock_set_next_hits(C.addressof(m.hits))
            X.fail_constructor_after(n)
            it = lib.RSGPU_NewTreeIterator(C.byref(tq), C.cast(m.term_args(terms), _vp), 1.0)
            assert it is None and b"allocate" in lib.RSGPU_Iterators_LastError(), n
            assert m.hits.freed == 1, n
        X.fail_constructor_after(-1)
        m = Mock(lists, ids, [0, 1, 2], groups=[(OP_T, 1.0, 1), (OP_U, 1.0, 2)])
        lib.mock_set_next_hits(C.addressof(m.hits))
        it = lib.RSGPU_NewTreeIterator(C.byref(tq), C.cast(m.term_args(), _vp), 1.0)    # (and it works once nothing fails)
        assert it
        assert X.iter_drain(it, len(ids) + 2, 2)["ids"].tolist() == ids
        X.iter_free(it)
    finally:
        X.fail_constructor_after(-1)
