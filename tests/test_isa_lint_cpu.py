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


def test_h8r_ring_registers_are_never_touched_in_flight():
    """gemm_qs_h8r_kernel (round 6) keeps two tiles in flight in registers behind inline-asm loads: hipcc believes they hold their
    values from the asm on.  scripts/isa_lint_h8r.py walks the tile loop of every instantiation in the BUILT object (twice, for the
    loop-carried loads) and the prologue: nothing but the quantiser's v_pk_fma_f16 may read such a register between its load and
    its counted wait, nothing may write or spill it."""
    import subprocess
    import sys
    obj = os.path.join(ROOT, "redisearch_amd", "lib", "obj", "gemm_qs_kernels.hip.o")
    if not os.path.exists(obj):
        pytest.skip("library not built")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_lint_h8r.py")], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("gemm_qs_h8r_kernel")]
    assert len(lines) >= 10 and all(ln.rstrip().endswith("OK") for ln in lines), p.stdout    # fp16 and fp32 rows, five widths each
    assert "gemm_qs_h8r_kernel<24,1,2,1>" in p.stdout and "registers in flight  48" in p.stdout   # 2 tiles x 6 chunks x 4 dwords
    assert "gemm_qs_h8r_kernel<24,1,1,2>" in p.stdout                                             # fp32 rows: 1 tile x 12 chunks
