#!/usr/bin/env python3
"""Register / LDS / scratch budget of every gfx950 kernel in the built objects (redisearch_amd/lib/obj/*.hip.o), read from the
code objects' metadata -- no GPU needed.  `python scripts/kernel_resources.py` prints one line per kernel FAMILY (template
name without its arguments); `--all` one line per kernel.  tests/test_kernel_resources_cpu.py asserts the budgets the design
relies on (no spills on any default path, two waves per SIMD for the matrix-core kernels, the LDS rings inside 160 KiB)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDIR = os.path.join(ROOT, "redisearch_amd", "lib", "obj")
LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
KEYS = ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size",
        "private_segment_fixed_size", "max_flat_workgroup_size")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    clean = []
    for n in out:
        if n.startswith("_Z"):             # (c++filt does not know _Float16 template arguments: keep the identifier)
            m = re.search(r"\d+([a-z0-9_]+_kernel)I?(.*)", n)
            n = "%s<%s>" % (m.group(1), m.group(2)[:24]) if m else n
        clean.append(n.replace("rsgpu::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
                     .replace("rsgpu::", "").replace("void ", ""))
    return clean


def kernels_of(obj):
    """[{name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, lds, scratch, wg}] of the gfx950 code object bundled in `obj`"""
    d = tempfile.mkdtemp()
    try:
        o = shutil.copy(obj, d)            # (llvm-objdump writes the bundles next to its input)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", o], stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, check=True)
        co = [f for f in os.listdir(d) if "gfx950" in f]
        if not co:
            return []
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(d, co[0])], capture_output=True,
                             text=True, check=True).stdout
    finally:
        shutil.rmtree(d)
    recs = []
    for blk in re.split(r"\n  - \.agpr_count:", "\n" + txt)[1:]:
        r = {"agpr": int(blk.split()[0])}
        for k in KEYS:
            m = re.search(r"\." + k + r":\s+(\S+)", blk)
            r[k] = m.group(1) if m else "0"
        recs.append(r)
    names = demangle([r["name"] for r in recs])
    return [dict(name=re.sub(r"\((GemmArgs|QsArgs|.*)\)$", "", n) if n.endswith(")") else n, vgpr=int(r["vgpr_count"]),
                 agpr=r["agpr"], sgpr=int(r["sgpr_count"]), vgpr_spill=int(r["vgpr_spill_count"]),
                 sgpr_spill=int(r["sgpr_spill_count"]), lds=int(r["group_segment_fixed_size"]),
                 scratch=int(r["private_segment_fixed_size"]), wg=int(r["max_flat_workgroup_size"]))
            for n, r in zip(names, recs)]


def all_kernels():
    out = []
    for f in sorted(os.listdir(OBJDIR)):
        if f.endswith(".hip.o"):
            for k in kernels_of(os.path.join(OBJDIR, f)):
                k["file"] = f[:-2]
                out.append(k)
    return out


def waves_per_simd(k):
    """resident wavefronts per SIMD the register budget allows (512 unified registers per lane; hardware cap 8), also
    bounded by what ONE workgroup needs (wg/64 waves over 4 SIMDs)"""
    return min(8, 512 // max(k["vgpr"], 1))


def main():
    ks = all_kernels()
    if "--all" in sys.argv:
        for k in ks:
            print("%-18s %-88s vgpr %3d (agpr %3d) spills %d/%d lds %6d scratch %4d wg %4d" % (
                k["file"], k["name"][:88], k["vgpr"], k["agpr"], k["vgpr_spill"], k["sgpr_spill"], k["lds"], k["scratch"], k["wg"]))
        return
    fam = {}
    for k in ks:
        fam.setdefault((k["file"], re.sub(r"<.*", "", k["name"])), []).append(k)
    print("%-18s %-34s %5s %11s %9s %8s %7s %9s" % ("file", "kernel family", "count", "vgpr", "waves/SIMD", "lds max", "scratch", "spilling"))
    for (f, n), v in sorted(fam.items()):
        vg = [k["vgpr"] for k in v]
        print("%-18s %-34s %5d %5d..%-4d %4d..%-4d %8d %7d %9d" % (
            f, n[:34], len(v), min(vg), max(vg), min(map(waves_per_simd, v)), max(map(waves_per_simd, v)),
            max(k["lds"] for k in v), max(k["scratch"] for k in v), sum(1 for k in v if k["vgpr_spill"] or k["sgpr_spill"])))


if __name__ == "__main__":
    main()
