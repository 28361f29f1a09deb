"""CPU: the hand-counted LDS pipeline of the query-stationary pass, checked on the compiler's output (scripts/isa_lint_qs.py).
The fragment reads and their `s_waitcnt lgkmcnt(N)` are inline asm, so the compiler does not know which registers still
have data in flight -- a register copy or re-use it schedules in between would be a silent wrong answer on the GPU.  The lint
models the LDS return queue over the gfx950 assembly of every gemm_qs_kernel instantiation; here it must come back clean,
and must bite when one counted wait is removed."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_lint_qs", os.path.join(ROOT, "scripts", "isa_lint_qs.py"))
L = importlib.util.module_from_spec(spec)
spec.loader.exec_module(L)


@pytest.fixture(scope="module")
def asm():
    if not os.path.exists(L.HIPCC):
        pytest.skip("no hipcc")
    path = L.assemble()
    yield path
    shutil.rmtree(os.path.dirname(path), ignore_errors=True)


def test_no_register_is_touched_with_its_lds_read_in_flight(asm):
    res = L.lint(asm)
    assert len(res) >= 28                         # 3 types x 5 row widths x 2 shapes, minus the int8 width not offered
    for name, (n_reads, findings) in res.items():
        assert n_reads >= 8, name                 # the inline reads were found (KS fragments per tile)
        assert not findings, (name, findings[:3])


def test_the_lint_bites(asm):
    lines = open(asm).read().split("\n")
    waits = [i for i, l in enumerate(lines) if "s_waitcnt lgkmcnt(2)" in l]
    assert len(waits) > 100
    del lines[waits[len(waits) // 2]]
    bad = asm + ".bad"
    with open(bad, "w") as f:
        f.write("\n".join(lines))
    findings = [x for _, (_, fs) in L.lint(bad).items() for x in fs]
    assert findings and "in flight" in findings[0][1]
