"""Boundary 2: the scorer plugin librsgpu_scorers.so, loaded the way RediSearch loads an extension
(dlopen + RS_ExtensionInit + RSExtensionCtx, reference src/extension.c:105-145) by oracle/ext_harness.c, and compared
with the REFERENCE's own scorers -- src/ext/default.c + src/index_result/index_result.c compiled in place into
oracle/_ref/libref_default_ext.so -- on the same result trees: scores bit-identical, EXPLAINSCORE text identical.
CPU only; the reference-comparison tests skip where neither /root/reference nor a prebuilt oracle/_ref exists.
"""
import math
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle as O
from oracle import ext as X

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFROOT = "/root/reference"

need_ref = pytest.mark.skipif(not X.have_ref(), reason="oracle/_ref/libref_default_ext.so not available")
SCORERS = ["TFIDF", "TFIDF.DOCNORM", "BM25", "BM25STD", "BM25STD.TANH", "BM25STD.NORM", "DISMAX", "DOCSCORE"]


@pytest.fixture(scope="module", autouse=True)
def _built():
    from redisearch_amd import build as B
    B.build_c()
    assert os.path.exists(X.PLUGIN)


# ---- the boundary itself -------------------------------------------------------------------------------------------------
def test_exports_and_undefined_symbols():
    """Exports RS_ExtensionInit (dlsym target, src/extension.c:130); the only undefined non-libc symbols are the
    module's result-tree accessors."""
    out = subprocess.check_output(["nm", "-D", X.PLUGIN], text=True)
    defined = {l.split()[-1] for l in out.splitlines() if " T " in l}
    undefined = {l.split()[-1] for l in out.splitlines() if " U " in l and "@" not in l}
    assert {"RS_ExtensionInit", "RSGPU_Scorers_SetAllocator"} <= defined
    assert undefined == {"AggregateResult_Get", "AggregateResult_GetRecordsSlice", "IndexResult_AggregateRefUnchecked",
                         "IndexResult_QueryTermRef", "QueryTerm_GetBM25_IDF", "QueryTerm_GetIDF",
                         "QueryTerm_GetStrAndLen"}


def test_registration_plain_then_prefixed_then_refused():
    """A module without the default scorers gets the plain aliases; where they are taken (duplicate alias ->
    REDISEARCH_ERR, src/extension.c:79-82) the plugin falls back to RSGPU.<alias>; a third load finds both taken."""
    h = X.Host()
    assert h.load_plugin() == X.OK
    assert h.aliases() == X.DEFAULT_ALIASES
    assert h.load_plugin() == X.OK
    assert h.aliases() == X.DEFAULT_ALIASES + ["RSGPU." + a for a in X.DEFAULT_ALIASES]
    assert h.load_plugin() == X.ERR
    assert len(h.aliases()) == 18
    with pytest.raises(KeyError):
        h.score("bm25std", X.Tree(("term", 1.0, 1, 1.0, 1.0, "a", None)), slop=1)  # lookup is case sensitive


def test_missing_init_symbol_is_reported():
    h = X.Host()
    with pytest.raises(OSError, match="no NoSuchInit"):
        h.load(X.PLUGIN, "NoSuchInit")


@need_ref
def test_inside_a_stock_module_the_aliases_are_prefixed():
    h = X.Host()
    assert h.load_ref() == X.OK                       # DefaultExtensionInit: 9 scorers + 4 expanders
    assert h.aliases() == X.DEFAULT_ALIASES and h.L.xh_num_expanders() == 4
    assert h.load_plugin() == X.OK
    assert h.aliases()[9:] == ["RSGPU." + a for a in X.DEFAULT_ALIASES]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFROOT, "src")) or shutil.which("gcc") is None,
                    reason="needs /root/reference")
def test_layout_matches_reference_headers(tmp_path):
    """include/rs_extension.h restates shared-memory structs: every sizeof/offsetof/bit-field byte/enum value/alias must
    equal what the reference's own headers give (tests/ext_layout_probe.c compiled against each)."""
    probe = os.path.join(ROOT, "tests", "ext_layout_probe.c")
    mine, ref = str(tmp_path / "mine"), str(tmp_path / "ref")
    subprocess.check_call(["gcc", "-std=gnu11", "-I" + os.path.join(ROOT, "include"), probe, "-o", mine])
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-D_GNU_SOURCE", "-DPROBE_REFERENCE", "-w",  # -O1: drops redismodule.h's unused static init
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "ref_compile_stubs"),
                           "-Isrc", "-Ideps", "-Ideps/rmalloc", "-Isrc/redisearch_rs/headers", "-Ideps/rmutil", "-I.",
                           probe, "-o", ref], cwd=REFROOT)
    a, b = subprocess.check_output([mine], text=True), subprocess.check_output([ref], text=True)
    assert a == b and a.count("\n") > 45


# ---- known answers through the plugin ----------------------------------------------------------------------------------
def _score_index_docs():
    """testScoreIndex corpus, reference tests/pytests/test_scorers.py:31-69 (see tests/test_oracle_scoring.py)."""
    N = 25
    for n in range(1, N):
        sc = float(np.float32(math.sqrt(float(N - n + 10) / float(N + 10))))
        yield dict(n=n, score=sc, f=10 * n, doc_len=22 * n, max_freq=10 * n,
                   hello=list(range(1, 2 * n, 2)), world=list(range(2, 2 * n + 1, 2)))


@pytest.mark.parametrize("scorer,expected", [
    ("TFIDF", [("doc1", 1.97), ("doc2", 1.94), ("doc3", 1.91), ("doc4", 1.88), ("doc5", 1.85)]),
    ("TFIDF.DOCNORM", [("doc1", 0.9), ("doc2", 0.88), ("doc3", 0.87), ("doc4", 0.86), ("doc5", 0.84)]),
    ("BM25", [("doc17", 0.73), ("doc18", 0.73), ("doc16", 0.72), ("doc19", 0.72), ("doc15", 0.72)]),
    ("BM25STD", [("doc1", 0.08), ("doc2", 0.08), ("doc3", 0.08), ("doc4", 0.08), ("doc5", 0.08)]),
    ("BM25STD.TANH", [("doc1", 0.02), ("doc2", 0.02), ("doc3", 0.02), ("doc4", 0.02), ("doc5", 0.02)]),
    ("DISMAX", [("doc24", 480.0), ("doc23", 460.0), ("doc22", 440.0), ("doc21", 420.0), ("doc20", 400.0)]),
    ("DOCSCORE", [("doc1", 0.99), ("doc2", 0.97), ("doc3", 0.96), ("doc4", 0.94), ("doc5", 0.93)]),
])
def test_score_index_top5_through_the_plugin(scorer, expected):
    """The seven scorers' top-5 of reference tests/pytests/test_scorers.py:31-69 ('hello world', weight-10 title)."""
    h = X.Host()
    assert h.load_plugin() == X.OK
    docs = list(_score_index_docs())
    nd = len(docs)
    avg = sum(d["doc_len"] for d in docs) / nd
    idf, bidf = O.lib.oracle_idf(nd, nd), O.lib.oracle_idf_bm25(nd, nd)
    got = []
    for d in docs:
        t = X.Tree(("intersection", 1.0, [("term", 1.0, d["f"], idf, bidf, "hello", d["hello"]),
                                           ("term", 1.0, d["f"], idf, bidf, "world", d["world"])]))
        s = h.score(scorer, t, doc_score=d["score"], max_freq=d["max_freq"], doc_len=d["doc_len"], num_docs=nd,
                    avg_doc_len=avg, slop=1)  # adjacent positions: IndexResult_MinOffsetDelta = 1
        got.append(("doc%d" % d["n"], s))
    got.sort(key=lambda x: (-x[1], int(x[0][3:])))
    for (gd, gs), (ed, es) in zip(got[:5], expected):
        assert gs == pytest.approx(es, abs=0.006), (scorer, got[:5])
    if scorer in ("DISMAX",):
        assert [g[0] for g in got[:5]] == [e[0] for e in expected]


# ---- plugin == reference default.c, bit for bit ------------------------------------------------------------------------
def _rand_leaf(rng, names):
    kind = rng.choice(["term", "term", "term", "term", "virtual", "numeric", "metric", "noterm"])
    w = float(rng.choice([1.0, 1.0, 0.5, 2.0, 0.0, 10.0]))
    f = int(rng.choice([0, 1, 1, 2, 3, 7, 40]))
    if kind == "term":
        npos = int(rng.integers(0, 5))
        pos = sorted(set(int(x) for x in rng.integers(1, 60, npos)))
        return ("term", w, f, float(rng.uniform(0.1, 9)), float(rng.uniform(0.0, 6)), str(rng.choice(names)), pos)
    if kind == "noterm":
        return ("term", w, f, None, None, None, None)
    return (kind, w, f)


def _rand_tree(rng, depth, names):
    if depth == 0 or rng.random() < 0.25:
        return _rand_leaf(rng, names)
    kind = str(rng.choice(["union", "intersection", "intersection"]))
    n = int(rng.integers(1, 5))
    return (kind, float(rng.choice([1.0, 1.0, 0.3, 3.0])), [_rand_tree(rng, depth - 1, names) for _ in range(n)])


def _both():
    h = X.Host()
    assert h.load_ref() == X.OK and h.load_plugin() == X.OK   # plugin lands on RSGPU.<alias>
    return h


@need_ref
@pytest.mark.parametrize("seed", range(6))
def test_plugin_equals_reference_on_random_trees(seed):
    """Flat and nested (up to 3 levels) trees of terms / virtual / numeric / metric / term-less records, random weights
    incl. 0, random doc metadata incl. score 0 and norm 0, min_score cut-offs; GetSlop = the reference's own
    IndexResult_MinOffsetDelta over random positions.  Scores must be the same double, explanations the same text."""
    rng = np.random.default_rng(1000 + seed)
    h = _both()
    names = ["hello", "world", "lorem", "ipsum", "x"]
    for it in range(150):
        spec = _rand_tree(rng, int(rng.integers(0, 4)), names)
        if spec[0] not in ("union", "intersection"):
            spec = ("intersection", 1.0, [spec])
        t = X.Tree(spec)
        kw = dict(doc_score=float(rng.choice([1.0, 0.5, 0.0, 0.83])), max_freq=int(rng.choice([0, 1, 5, 100])),
                  doc_len=int(rng.choice([0, 1, 17, 300, 70000])), num_docs=int(rng.integers(1, 10 ** 6)),
                  avg_doc_len=float(rng.uniform(0.5, 400)), min_score=float(rng.choice([0.0, 0.0, 0.5, 50.0])),
                  tanh_factor=int(rng.choice([1, 4, 12, 10000])))
        for sc in SCORERS:
            if sc.startswith("BM25STD") and "None" in repr(spec):
                # a Term record without a query term: the reference dereferences it unconditionally in the BM25STD
                # family (QueryTerm_GetBM25_IDF requires non-null, query_term_ffi.h:44-52); the plugin scores it 0
                assert math.isfinite(h.score("RSGPU." + sc, t, explain=True, **kw)[0])
                continue
            ref, rtxt = h.score(sc, t, explain=True, **kw)
            got, gtxt = h.score("RSGPU." + sc, t, explain=True, **kw)
            assert (got == ref) or (got != got and ref != ref), (sc, spec, kw, got, ref)
            assert gtxt == rtxt, (sc, spec, kw)
            assert h.score("RSGPU." + sc, t, **kw) == got or got != got   # without EXPLAINSCORE: same value


@need_ref
def test_hybrid_metric_trees():
    """HybridMetric(vector metric, text child): summed like any aggregate by the TF-IDF/BM25 families, DISMAX descends
    into child 1 unweighted (src/ext/default.c:442-448); pins reference tests/pytests/test_vecsim.py:1248-1341 values."""
    h = _both()
    N, avg = 100, 1.9
    other = ("term", 1.0, 1, O.lib.oracle_idf(N, 10), O.lib.oracle_idf_bm25(N, 10), "other", [1])
    t = X.Tree(("hybrid", 1.0, [("metric", 1.0, 0), ("union", 1.0, [other])]))
    kw = dict(num_docs=N, avg_doc_len=avg, doc_len=1, max_freq=1)
    assert h.score("RSGPU.TFIDF", t, **kw) == 3.0
    assert h.score("RSGPU.BM25STD", t, **kw) == pytest.approx(2.8078501570291188, rel=1e-7)
    for sc in SCORERS:
        assert h.score("RSGPU." + sc, t, explain=True, **kw) == h.score(sc, t, explain=True, **kw)
    t2 = X.Tree(("hybrid", 7.0, [("metric", 1.0, 0), ("union", 2.0, [other, other])]))
    assert h.score("RSGPU.DISMAX", t2, **kw) == h.score("DISMAX", t2, **kw) == 2.0  # hybrid weight 7 not applied


@need_ref
def test_hamming_equals_reference():
    h = _both()
    rng = np.random.default_rng(5)
    t = X.Tree(("term", 1.0, 1, 1.0, 1.0, "a", None))
    for n in (1, 7, 8, 9, 64, 100):
        a, b = rng.integers(0, 256, n, dtype=np.uint8).tobytes(), rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for pa, qb in ((a, b), (a, a), (a, b[:-1] if n > 1 else b""), (None, b)):
            r = h.score("HAMMING", t, payload=pa, qdata=qb, explain=True, slop=1)
            g = h.score("RSGPU.HAMMING", t, payload=pa, qdata=qb, explain=True, slop=1)
            assert g == r
    assert h.score("RSGPU.HAMMING", t, payload=b"\x00\xff", qdata=b"\x00\x00", slop=1) == 1.0 / 9.0


@need_ref
def test_explainscore_text_of_bm25std_matches_the_reference_pytest_shape():
    """reference tests/pytests/test_scorers.py:198-242: 'Final BM25 : words BM25 ... * document score 1.00' over
    '(Weight 1.00 * children BM25 ...)' over the per-term lines; the plugin's tree is the reference's."""
    h = _both()
    t = X.Tree(("intersection", 1.0, [("term", 1.0, 1, 0.1, 0.09, "hello", [1]), ("term", 1.0, 1, 0.1, 0.09, "world", [2])]))
    kw = dict(num_docs=3, avg_doc_len=3.0, doc_len=2, max_freq=1)
    v, txt = h.score("RSGPU.BM25STD", t, explain=True, **kw)
    assert (v, txt) == h.score("BM25STD", t, explain=True, **kw)
    lines = txt.splitlines()
    assert lines[0].startswith("Final BM25 : words BM25 ") and lines[0].endswith("* document score 1.00")
    assert lines[1].strip().startswith("(Weight 1.00 * children BM25 ")
    assert lines[2].strip().startswith("hello: (") and lines[3].strip().startswith("world: (")
    v2, txt2 = h.score("RSGPU.BM25STD.TANH", t, explain=True, tanh_factor=4, **kw)
    assert txt2.splitlines()[0].startswith("Final Normalized BM25 : tanh(stretch factor 1/4 * Final BM25 ")
    assert v2 == math.tanh((1 / 4.0) * v)


# ---- the restated oracle == the reference (pins oracle/scoring_oracle.c on the reference itself) ------------------------
def _to_oracle(spec):
    k = spec[0]
    if k == "term":
        _, w, f, idf, bidf, _name, pos = spec
        return O.Node(O.R_TERM, w, f, idf, bidf or 0.0, pos)
    if k in ("virtual", "numeric", "metric"):
        return O.Node({"virtual": O.R_VIRTUAL, "numeric": O.R_NUMERIC, "metric": O.R_METRIC}[k], spec[1], spec[2])
    kids = [_to_oracle(c) for c in spec[2]]
    tag = {"union": O.R_UNION, "intersection": O.R_INTERSECTION, "hybrid": O.R_HYBRID}[k]
    return O.Node(tag, spec[1], sum(c.c.freq for c in kids), children=kids)


@need_ref
@pytest.mark.parametrize("seed", range(4))
def test_restated_oracle_equals_reference(seed):
    """oracle/scoring_oracle.c (the restatement every GPU scoring test is checked against) vs the reference's compiled
    default.c + IndexResult_MinOffsetDelta on the same random trees: bit-identical scores, identical slop."""
    rng = np.random.default_rng(77 + seed)
    h = X.Host()
    assert h.load_ref() == X.OK
    for it in range(200):
        spec = _rand_tree(rng, int(rng.integers(1, 4)), ["a", "b", "c"])
        if spec[0] not in ("union", "intersection"):
            spec = ("intersection", 1.0, [spec])
        t, o = X.Tree(spec), _to_oracle(spec)
        flat_terms = all(c[0] == "term" for c in spec[2])
        if flat_terms:   # the oracle's slop reads positions of direct Term children (its documented domain)
            assert O.lib.oracle_slop(o.ptr) == h.ref_slop(t), spec
        kw = dict(doc_score=float(rng.choice([1.0, 0.25, 0.0])), max_freq=int(rng.choice([0, 3, 50])),
                  doc_len=int(rng.choice([0, 9, 1234])), num_docs=int(rng.integers(1, 10 ** 5)),
                  avg_doc_len=float(rng.uniform(1, 300)), min_score=float(rng.choice([0.0, 0.0, 2.0])),
                  tanh_factor=int(rng.choice([1, 4, 20])))
        for sc in ("TFIDF", "TFIDF.DOCNORM", "BM25", "BM25STD", "BM25STD.TANH", "DISMAX", "DOCSCORE"):
            if not flat_terms and sc in ("TFIDF", "TFIDF.DOCNORM", "BM25"):
                continue   # slop-dependent scorers: compared where the oracle's slop domain holds
            if sc.startswith("BM25STD") and "None" in repr(spec):
                continue   # term-less Term record: undefined in the reference's BM25STD family
            ref = h.score(sc, t, **kw)
            got = O.score(sc, o, **{k: v for k, v in kw.items()})
            if sc == "BM25STD.TANH":
                assert got == pytest.approx(ref, rel=1e-15, abs=1e-300)
            else:
                assert got == ref or (got != got and ref != ref), (sc, spec, kw, got, ref)


# ---- what a NOT child is to the scorers (round 4: the general tile kernel's `a -b`) ---------------------------------------------
@need_ref
def test_a_not_child_is_a_virtual_result_of_frequency_zero_to_the_reference_scorers():
    """`a b -c` is Intersection{a, b, Not{c}}; the Not iterator's result is VIRTUAL, frequency 0, weight = the node's
    (rqe_iterators/src/not.rs:106-118, index_result/src/core/mod.rs:103-112).  On the REFERENCE's compiled scorers
    (src/ext/default.c) and IndexResult_MinOffsetDelta (src/index_result/index_result.c): it adds nothing to any scorer's sum --
    the scores of Intersection{a, b, Virtual} are those of Intersection{a, b} whenever the two have the same slop -- and it
    counts as a child for the slop: with term offsets the pairs of children that HAVE offsets decide (same value), without
    them the answer is children - 1 (2 instead of 1).  The restated oracle agrees; this is what `hybrid_general` builds its
    ScoreParams / ProxParams from (an empty aggregate per NOT child)."""
    h = X.Host()
    assert h.load_ref() == X.OK
    rng = np.random.default_rng(5)
    for it in range(100):
        with_offsets = bool(it % 2)
        terms = []
        for name in ("a", "b"):
            pos = sorted(set(int(x) for x in rng.integers(1, 40, int(rng.integers(1, 4))))) if with_offsets else None
            terms.append(("term", float(rng.choice([1.0, 0.5, 2.0])), int(rng.integers(1, 20)), float(rng.uniform(0.1, 5)),
                          float(rng.uniform(0.1, 5)), name, pos))
        wn = float(rng.choice([1.0, 0.3, 4.0]))
        pos_spec = ("intersection", 1.5, terms)
        not_spec = ("intersection", 1.5, terms + [("virtual", wn, 0)])
        tp, tn = X.Tree(pos_spec), X.Tree(not_spec)
        sp, sn = h.ref_slop(tp), h.ref_slop(tn)
        if with_offsets:
            # decided by the children that have offsets -- unless their distance is 0 (the same position in both terms):
            # "return dist ? sqrt(dist) : num - 1" falls back to the child count there too
            assert sn == sp or (sp, sn) == (1, 2)
        else:
            assert (sp, sn) == (1, 2)                         # children - 1
        assert O.lib.oracle_slop(_to_oracle(not_spec).ptr) == sn
        kw = dict(doc_score=float(rng.choice([1.0, 0.25])), max_freq=int(rng.choice([3, 50])), doc_len=int(rng.choice([9, 1234])),
                  num_docs=int(rng.integers(10, 10 ** 5)), avg_doc_len=float(rng.uniform(1, 300)), tanh_factor=4)
        for sc in ("TFIDF", "TFIDF.DOCNORM", "BM25", "BM25STD", "BM25STD.TANH", "DISMAX", "DOCSCORE"):
            ref_n, ref_p = h.score(sc, tn, **kw), h.score(sc, tp, **kw)
            got_n = O.score(sc, _to_oracle(not_spec), **kw)
            if sc == "BM25STD.TANH":
                assert got_n == pytest.approx(ref_n, rel=1e-15, abs=1e-300)
            else:
                assert got_n == ref_n, (sc, not_spec, got_n, ref_n)
            if sc in ("TFIDF", "TFIDF.DOCNORM", "BM25") and sn != sp:
                assert ref_n == ref_p * sp / sn or ref_n == pytest.approx(ref_p * sp / sn, rel=1e-15)   # the slop divides
            else:
                assert ref_n == ref_p, (sc, ref_n, ref_p)
