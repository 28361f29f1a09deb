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
    r = b.cpu_baseline(32, 10, 2000, 10_000_000, budget_s=0.5)
    assert set(r) >= {"value", "unit", "cores", "kind", "sample"}
    assert r["unit"] == "queries/s" and r["cores"] == 1 and r["kind"] == "port" and r["value"] > 0
    assert r["eight_threads"]["cores"] == 8 and r["eight_threads"]["value"] > 0
    # scaled by rows: the sample rate shrinks by 2000 / 10M
    assert "scaled by rows to 10000000" in r["sample"]
