"""bench.py's CPU-visible contract: the flags the driver passes parse, and the cpu_baseline leg (the only place
bench.py touches oracle/) returns the object the bench line carries.  The timed GPU region itself is exercised on
the MI355X (bench.py asserts a device)."""
import importlib
import sys


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
