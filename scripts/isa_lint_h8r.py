#!/usr/bin/env python3
"""ISA lint of gemm_qs_h8r_kernel (no GPU needed): the kernel keeps tiles in flight in registers that hipcc believes were written
by the inline-asm load at its issue.  That is only sound if, INSIDE the tile loop, nothing but (a) the asm load itself writes a ring
register and (b) nothing reads one except the quantiser's v_pk_fma_f16 / v_fma_f32 behind an s_waitcnt vmcnt -- in particular no v_mov /
v_accvgpr copy of a ring register (a copy made while the load is in flight carries stale bits).  Prints one line per kernel and
exits 1 on a violation.  tests/test_isa_lint_cpu.py runs it."""
import os
import re
import subprocess
import sys
import tempfile
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "redisearch_amd", "lib", "obj", "gemm_qs_kernels.hip.o")
LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def disassemble():
    d = tempfile.mkdtemp()
    try:
        o = shutil.copy(OBJ, d)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", o], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        co = [f for f in os.listdir(d) if "gfx950" in f][0]
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(d, co)], capture_output=True, text=True, check=True).stdout
    finally:
        shutil.rmtree(d, ignore_errors=True)


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def lint(name, body):
    """body: [(mnemonic, [operands])] of the tile loop (first s_barrier .. the backward s_branch / s_cbranch), scanned TWICE so that
    the loads issued at the end of one trip are in flight at the top of the next.  A register is IN FLIGHT from its nt load until a
    v_pk_fma_f16 reads it; anything else touching it in between is a violation."""
    flight, bad, nring = set(), [], 0
    for trip in range(2):
        for mn, ops in body:
            if not ops:
                continue
            if mn == "global_load_dwordx4" and "nt" in ops:
                dst = regs_of(ops[0])
                if dst & flight:
                    bad.append("a ring load overwrites registers still in flight %s" % sorted(dst & flight))
                flight |= dst
                nring = max(nring, len(flight))
                continue
            stores = mn.startswith(("global_store", "ds_write", "buffer_store", "scratch_store"))
            dst = set() if stores else regs_of(ops[0])
            srcs = set().union(*[regs_of(o) for o in (ops if stores else ops[1:])]) if ops else set()
            if mn in ("v_pk_fma_f16", "v_fma_f32"):   # the quantisers (fp16 / fp32 rows)
                flight -= srcs          # consumed (behind its s_waitcnt vmcnt)
                continue
            if trip == 1 and dst & flight:
                bad.append("%s writes in-flight %s" % (mn, sorted(dst & flight)))
            if trip == 1 and srcs & flight:
                bad.append("%s reads in-flight %s" % (mn, sorted(srcs & flight)))
    return nring, bad


def lint_prologue(body):
    """before the loop: a register is in flight from its nt load to the next s_waitcnt vmcnt(0) (the prologue drains there)"""
    flight, bad = set(), []
    for mn, ops in body:
        if mn == "s_waitcnt" and "vmcnt(0)" in ops:
            flight = set()
            continue
        if not ops:
            continue
        if mn == "global_load_dwordx4" and "nt" in ops:
            flight |= regs_of(ops[0])
            continue
        stores = mn.startswith(("global_store", "ds_write", "buffer_store", "scratch_store"))
        dst = set() if stores else regs_of(ops[0])
        srcs = set().union(*[regs_of(o) for o in (ops if stores else ops[1:])]) if ops else set()
        if (dst | srcs) & flight:
            bad.append("prologue: %s touches in-flight %s" % (mn, sorted((dst | srcs) & flight)))
    if flight:
        bad.append("prologue: the loop is entered with undrained registers %s" % sorted(flight)[:4])
    return bad


def main():
    text = disassemble()
    ok = True
    cur, body, seen_barrier = None, [], False
    results = []

    def flush():
        nonlocal ok
        if cur and "gemm_qs_h8r_kernel" in cur and body:
            # the tile loop: the backward branch with the longest span and its target (simm16 dwords from the next instruction)
            back = [(i, a + 4 + 4 * (int(ops[0]) - 65536)) for i, (mn, ops, a) in enumerate(body)
                    if mn.startswith(("s_branch", "s_cbranch")) and ops and ops[0].isdigit() and int(ops[0]) > 32768 and a is not None]
            if not back:
                results.append((cur, 0, ["no backward branch found"]))
                ok = False
                return
            # the tile loop: the longest backward span that starts BEHIND the prologue's barrier (block placement may add longer
            # jumps back into the prologue)
            first_barrier = next(i for i, (mn, _, _) in enumerate(body) if mn == "s_barrier")
            idx_of = {a: i for i, (_, _, a) in enumerate(body) if a is not None}
            back = [(i, t) for i, t in back if t in idx_of and idx_of[t] > first_barrier]
            if not back:
                results.append((cur, 0, ["no tile loop found behind the prologue"]))
                ok = False
                return
            end, target = max(back, key=lambda bt: body[bt[0]][2] - bt[1])
            start = idx_of[target]
            ring, bad = lint(cur, [(mn, ops) for mn, ops, _ in body[start:end + 1]])
            bad += lint_prologue([(mn, ops) for mn, ops, _ in body[:start]])
            results.append((cur, ring, bad))
            if bad or not ring:
                ok = False
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            flush()
            cur, body, seen_barrier = m.group(1), [], False
            continue
        if cur is None or "gemm_qs_h8r_kernel" not in cur:
            continue
        code = line.split("//")[0].strip()
        if not code:
            continue
        am = re.search(r"//\s*([0-9A-Fa-f]+):", line)
        addr = int(am.group(1), 16) if am else None
        parts = code.replace(",", " ").split()
        mn, ops = parts[0], parts[1:]
        body.append((mn, ops, addr))
    flush()
    for name, nring, bad in results:
        short = re.search(r"gemm_qs_h8r_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
        tag = "gemm_qs_h8r_kernel<%s,%s,%s,%s>" % short.groups() if short else name
        print("%-36s registers in flight %3d  %s" % (tag, nring, "OK" if not bad else "VIOLATIONS: " + "; ".join(sorted(set(bad))[:6])))
    if not results:
        print("no gemm_qs_h8r_kernel in the object")
        ok = False
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
