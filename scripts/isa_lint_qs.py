#!/usr/bin/env python3
"""Static check of the query-stationary pass's hand-counted LDS pipeline (redisearch_amd/csrc/gemm_qs_kernels.hip).

The kernel issues its fragment reads as inline-asm `ds_read_b128` and waits for them with inline-asm `s_waitcnt lgkmcnt(N)`
(the compiler would drain the queue otherwise).  The compiler therefore does NOT know that a destination register is still
in flight after the read: nothing stops it from copying, spilling or re-using such a register before the data lands.  This
lint walks the gfx950 assembly of every gemm_qs_kernel instantiation in layout order, models the LDS return queue (in
order; `lgkmcnt(N)` leaves at most N reads outstanding -- scalar loads in the same counter only make that conservative) and
reports every instruction that touches a register with a read still in flight, plus reads still in flight at a control
transfer the walk cannot follow (unconditional branch / end of program).

    python scripts/isa_lint_qs.py [file.s]      (without a file: compiles the kernel file with hipcc -S first)
tests/test_isa_lint_cpu.py runs it on every build."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "redisearch_amd", "csrc", "gemm_qs_kernels.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def assemble(src=SRC):
    out = os.path.join(tempfile.mkdtemp(), "qs.s")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src),
                           "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", out])
    return out


def regs(text):
    """VGPR numbers an operand string mentions (v7, v[4:7]); AGPRs and SGPRs are not LDS destinations here"""
    out = set()
    for a, b in re.findall(r"(?<![a-z0-9_])v\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"(?<![a-z0-9_\[:])v(\d+)(?![\d:\]])", text))
    return out


def lint_function(name, lines):
    findings, fifo, n_reads, carried = [], [], 0, 0
    for ln, raw in lines:
        s = raw.split(";")[0].strip()
        if not s or s.endswith(":") or s.startswith("."):
            continue
        op, _, rest = s.partition(" ")
        m = re.search(r"lgkmcnt\((\d+)\)", s)
        if op == "s_waitcnt":
            if m:
                del fifo[:max(0, len(fifo) - int(m.group(1)))]
            continue
        if op == "ds_read_b128":
            dst, addr = [x.strip() for x in rest.split(",")[:2]]
            busy = set().union(*fifo) if fifo else set()
            if regs(addr.split()[0]) & busy or regs(dst) & busy:
                findings.append((ln, "read issued over registers still in flight: " + s))
            fifo.append(regs(dst))
            n_reads += 1
            continue
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            if fifo:
                carried += 1
                findings.append((ln, "%d LDS read(s) in flight at %s" % (len(fifo), op)))
            fifo = []
            continue
        if fifo and op.startswith(("v_", "global_", "buffer_", "ds_", "scratch_", "flat_")):
            hit = regs(rest) & set().union(*fifo)
            if hit:
                findings.append((ln, "touches v%s with its LDS read in flight: %s" % (sorted(hit), s)))
    return n_reads, findings


def lint(path):
    """{kernel symbol: (number of inline LDS reads seen, [(line, message)])} for every gemm_qs_kernel in the assembly"""
    out, cur, buf = {}, None, []
    with open(path) as f:
        for ln, raw in enumerate(f, 1):
            m = re.match(r"^(_ZN5rsgpu\S*gemm_qs_(?:f32_)?kernel\S*):", raw)
            if m:
                cur, buf = m.group(1), []
                continue
            if cur:
                if raw.lstrip().startswith(".section") or raw.lstrip().startswith(".amdhsa_kernel"):
                    out[cur] = lint_function(cur, buf)
                    cur = None
                else:
                    buf.append((ln, raw))
    return out


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else assemble()
    res = lint(path)
    bad = 0
    for k, (n, f) in sorted(res.items()):
        print("%-90s %3d reads  %s" % (k[:90], n, "ok" if not f else "%d finding(s)" % len(f)))
        for ln, msg in f[:10]:
            print("    line %d: %s" % (ln, msg))
        bad += len(f)
    print("%d kernels, %d findings" % (len(res), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
