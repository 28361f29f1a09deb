"""bench.py's CPU-visible contract: the flags the driver passes parse, and the cpu_baseline leg (the only place
bench.py touches oracle/) returns the object the bench line carries.  The timed GPU region itself is exercised on
the MI355X (bench.py asserts a device)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench(argv):
    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        sys.modules.pop("bench", None)
        return importlib.import_module("bench")
    finally:
        sys.argv = old


def test_driver_flags_parse_and_defaults_are_the_baseline_config():
    b = load_bench([])
    old = sys.argv
    try:
        sys.argv = ["bench.py", "--gpus", "8", "--steps", "50", "--warmup", "5"]
        a = b.parse()
        assert (a.gpus, a.steps, a.warmup) == (8, 50, 5)
        assert (a.rows, a.dim, a.k, a.metric) == (10_000_000, 768, 10, "cosine")   # BASELINE configs[1]
        sys.argv = ["bench.py"]
        a = b.parse()
        assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0
    finally:
        sys.argv = old


def test_cpu_baseline_object():
    b = load_bench([])
    r = b.cpu_baseline(32, 10, 10_000_000, 2000, "cosine", "sample")
    assert set(r) >= {"value", "unit", "cores", "kind", "sample"}
    assert r["unit"] == "queries/s" and r["cores"] == 1 and r["kind"] == "port" and r["value"] > 0
    assert r["eight_threads"]["cores"] == 8 and r["eight_threads"]["value"] > 0
    # scaled by rows: the sample rate shrinks by 2000 / 10M, and the line says it was a sample
    assert "scaled by rows to 10000000" in r["sample"] and "SAMPLE" in r["sample"]


def test_cpu_baseline_full_mode_checks_gpu_answers_against_the_oracle():
    """full mode regenerates the whole keyed corpus on the host; handed the GPU's answers it also verifies them."""
    import numpy as np
    import oracle as O
    b = load_bench([])
    rows, dim, k = 3000, 16, 5
    o = O.FlatIndex(O.F32, dim, O.COSINE)
    o.add_bulk(O.philox_rows(b.SEED, 0, rows, dim), 1)
    qs = O.philox_rows(b.SEED, b.QUERY_BASE, 3, dim)
    answers = {i: tuple(x.tolist() for x in o.topk(qs[i], k)) for i in range(3)}
    r = b.cpu_baseline(dim, k, rows, 100, "cosine", "full", answers)
    assert "FULL corpus" in r["sample"] and r["value"] > 0
    chk = r["gpu_answers_checked_against_oracle_on_full_corpus"]
    assert chk == {"queries": 3, "ids_identical": True, "max_abs_score_diff": 0.0}
    bad = dict(answers)
    bad[1] = ([999] + answers[1][0][1:], answers[1][1])
    assert b.cpu_baseline(dim, k, rows, 100, "cosine", "full", bad)["gpu_answers_checked_against_oracle_on_full_corpus"]["ids_identical"] is False


def test_bench_posting_encoder_matches_the_oracle_block_writer():
    """bench.py encodes its synthetic FreqsOnly posting lists itself (numpy: the bench's inputs never come from oracle/);
    the bytes, block headers and offsets must equal the oracle's block writer for the same (doc, freq) pairs."""
    import numpy as np
    import oracle as O
    b = load_bench([])
    rng = np.random.default_rng(5)
    for n, max_gap, max_freq in ((0, 5, 5), (1, 5, 5), (99, 3, 300), (100, 70000, 70000), (101, 300, 2), (1234, 20_000_000, 1 << 30)):
        docs = np.cumsum(rng.integers(1, max_gap + 1, n)).astype(np.uint64)
        freqs = rng.integers(1, max_freq + 1, n).astype(np.uint32)
        mine = b.encode_freqs_only(docs, freqs)
        ii = O.InvertedIndex(O.C_FREQS_ONLY)
        if n:
            ii.add_many(docs, freqs)
        ref = ii.flatten()
        assert mine["codec"] == ref["codec"] == O.C_FREQS_ONLY
        for key in ("first", "last", "num_entries", "offset"):
            assert np.array_equal(np.asarray(mine[key], np.uint64), np.asarray(ref[key], np.uint64)), (n, key)
        assert bytes(mine["bytes"]) == bytes(ref["bytes"]), n


def test_bench_full_codec_encoder_matches_the_oracle_block_writer():
    """... and its Full-codec lists (configs[4]'s second variant): qint4 [delta, freq, mask, offsets length] +
    the offsets bytes, record by record what the oracle's writer produces."""
    import numpy as np
    import oracle as O
    b = load_bench([])
    rng = np.random.default_rng(6)
    for n, max_gap, max_freq in ((1, 5, 5), (99, 3, 9), (100, 70000, 3), (101, 300, 2), (731, 20_000_000, 40)):
        docs = np.cumsum(rng.integers(1, max_gap + 1, n)).astype(np.uint64)
        freqs = rng.integers(1, max_freq + 1, n).astype(np.uint32)
        masks = rng.integers(1, 1 << int(rng.integers(1, 31)), n).astype(np.uint32)
        offs = rng.integers(1, 128, int(freqs.sum())).astype(np.uint8)
        mine = b.encode_full(docs, freqs, masks, offs)
        ii = O.InvertedIndex(O.C_FULL)
        at = 0
        for d, f, m in zip(docs.tolist(), freqs.tolist(), masks.tolist()):
            ii.add(d, f, m, offs[at:at + f].tobytes())
            at += f
        ref = ii.flatten()
        assert mine["codec"] == ref["codec"] == O.C_FULL
        for key in ("first", "last", "num_entries", "offset"):
            assert np.array_equal(np.asarray(mine[key], np.uint64), np.asarray(ref[key], np.uint64)), (n, key)
        assert bytes(mine["bytes"]) == bytes(ref["bytes"]), n


def test_bench_imports_the_oracle_only_in_its_cpu_leg():
    """bench.py may use oracle/ only as the checker / CPU baseline: every import of it sits inside cpu_baseline's helpers,
    the host-side verification and the extras' oracle check, all of which run only when the cpu_baseline leg runs."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    allowed = {"_oracle_lib", "verify_answers", "check_hybrid_with_oracle", "check_batched_with_oracle"}
    users = set()
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        for node in ast.walk(fn):
            if isinstance(node, ast.Import) and any(al.name == "oracle" for al in node.names):
                users.add(fn.name)
            if isinstance(node, ast.ImportFrom) and (node.module or "").startswith("oracle"):
                users.add(fn.name)
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    assert not any(getattr(n, "module", None) == "oracle" or any(al.name == "oracle" for al in getattr(n, "names", [])) for n in top)
    assert users <= allowed, users
