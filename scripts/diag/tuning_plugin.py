"""pytest plugin for A/B runs of engine knobs without touching the tests:
    RSGPU_TUNING=qs_ovl=1,qs_phases=4 PYTHONPATH=scripts/diag python -m pytest -p tuning_plugin tests/... -m gpu
sets every key through RSGPU_SetTuning once the library is loaded (include/rsgpu_ext.h)."""
import os


def pytest_configure(config):
    spec = os.environ.get("RSGPU_TUNING", "")
    if not spec:
        return
    import torch  # noqa: F401  (first, as in the test modules: torch's own HIP runtime must be the one in the process)
    torch.cuda.is_available()
    from redisearch_amd import vecsim as V
    lib = V.load()
    for item in spec.split(","):
        key, _, val = item.partition("=")
        lib.RSGPU_SetTuning(key.strip().encode(), int(val))
        print("tuning_plugin: %s = %d" % (key.strip(), int(val)))
