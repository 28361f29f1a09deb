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


def test_line_fits():
    """The driver's capture lost round 5's 21.8 KB line: whatever the sub-records hold, the ONE stdout line stays below 8 KB
    and still carries the contract keys, `roofline` and `cpu_baseline`; the full record goes to a side file."""
    import json
    b = load_bench([])
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_1gpu_driver_args.json")))   # a real full record (21 KB)
    assert len(json.dumps(full)) > 20_000
    line = b.compact_line(full)
    text = json.dumps(line)
    assert len(text) < 8192, len(text)
    assert "\n" not in text
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "summary"):
        assert key in line, key
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"] and line["dtype"] == "f32"
    assert line["config"]["workload"].startswith("10000000x768 fp32 FLAT COSINE top-10")
    assert line["config"]["verify"]["ok"] is True
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert line["roofline"][key] == full["roofline"][key]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"]
    assert line["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    s = line["summary"]
    assert s["cfg3_f16"]["device_ms_per_pass"] > 0 and s["cfg3_f16"]["bit_identical"] is True
    assert s["cfg5_hybrid"]["warm_p50_ms"] > 0 and s["cfg5_hybrid"]["parity_ok"] is True
    assert s["callers"]["qps_64"] > 10_000

    def only_scalars(o, depth=0):
        assert depth <= 2
        for v in o.values():
            if isinstance(v, dict):
                only_scalars(v, depth + 1)
            else:
                assert v is None or isinstance(v, (int, float, bool, str)), v
    only_scalars(s)
    # an absurd sub-record cannot push the line over the limit either
    bloated = json.loads(json.dumps(full))
    bloated["config"]["concurrent_callers"].update({"%d_threads" % (100 + i): {"qps": 1.0 + i} for i in range(2000)})
    assert len(json.dumps(b.compact_line(bloated))) < 8192
