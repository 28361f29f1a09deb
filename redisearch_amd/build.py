"""Builds the MI355X engine: hand-written HIP kernels + the C-ABI shim -> redisearch_amd/lib/*.so.

    python -m redisearch_amd.build          # incremental
    python -m redisearch_amd.build --force

hipcc cross-compiles for gfx950 without a GPU present.  The shared objects stay in-tree (git-ignored)
so that they travel to the GPU box with the repository snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

COMMON = ["-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
          "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function"]

# library name -> sources
LIBS = {
    "libVectorSimilarity.so": ["scan_kernels.hip", "scan_mq_kernels.hip", "select_kernels.hip", "gemm_kernels.hip", "gemm_qs_kernels.hip", "fusion_kernels.hip", "postings_kernels.hip", "hybrid_kernels.hip", "corpus_kernels.hip", "exchange_kernels.hip",
                               "flat_index.cpp", "label_table.cpp", "grow_buffer.cpp", "batch_query.cpp", "sharded_index.cpp", "shard_comm.cpp", "vecsim_abi.cpp", "search_abi.cpp"],
}


def _deps_mtime():
    m = 0.0
    for d in (CSRC, os.path.join(ROOT, "include"), os.path.join(ROOT, "include", "VecSim")):
        for f in os.listdir(d):
            if f.endswith((".h", ".hpp", ".inc")):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


# per-source extra flags: the scorers must not contract a*b+c behind the C source's back
EXTRA = {"postings_kernels.hip": ["-ffp-contract=off"], "hybrid_kernels.hip": ["-ffp-contract=off"], "fusion_kernels.hip": ["-ffp-contract=off"]}


def _compile(src, force, hdr_m):
    obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
            and os.path.getmtime(obj) >= hdr_m):
        return obj
    cmd = [HIPCC] + COMMON + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
    subprocess.check_call(cmd)
    return obj


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_m = _deps_mtime()
    out = []
    for lib, srcs in LIBS.items():
        paths = [os.path.join(CSRC, s) for s in srcs if os.path.exists(os.path.join(CSRC, s))]
        with ThreadPoolExecutor(max_workers=min(8, len(paths))) as ex:
            objs = list(ex.map(lambda s: _compile(s, force, hdr_m), paths))
        target = os.path.join(LIBDIR, lib)
        if force or not os.path.exists(target) or any(os.path.getmtime(o) > os.path.getmtime(target) for o in objs):
            cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", target] + objs + ["-ldl"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        out.append(target)
    return out + build_c(force, verbose)


# plain-C shared objects (gcc): the scorer plugin a RediSearch module loads with EXTLOAD.  The result-tree accessors
# stay undefined in it (the module provides them), hence no -Wl,--no-undefined.
CC = os.environ.get("CC", "gcc")
C_LIBS = {
    "librsgpu_scorers.so": (["scorer_plugin.c"],
                            ["-O2", "-std=gnu11", "-fPIC", "-shared", "-fvisibility=hidden", "-ffp-contract=off", "-Wall",
                             "-Wextra", "-I" + os.path.join(ROOT, "include")], ["-ldl", "-lm"]),
    # Boundary 3: the reference's QueryIterator vtable over device hit lists.  Needs the engine (RSGPU_* symbols) next to
    # it; the RSIndexResult constructors are looked up in the process (or installed) at run time.
    "librsgpu_iterators.so": (["query_iterators.c"],
                              ["-O2", "-std=gnu11", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wextra",
                               "-I" + os.path.join(ROOT, "include")],
                              ["-L" + LIBDIR, "-lVectorSimilarity", "-Wl,-rpath,$ORIGIN", "-ldl"]),
}


def build_c(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    hdr_m = _deps_mtime()
    out = []
    for lib, (srcs, flags, libs) in C_LIBS.items():
        paths = [os.path.join(CSRC, s) for s in srcs]
        target = os.path.join(LIBDIR, lib)
        newest = max([os.path.getmtime(x) for x in paths] + [hdr_m])
        if force or not os.path.exists(target) or os.path.getmtime(target) < newest:
            cmd = [CC] + flags + paths + ["-o", target] + libs
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        out.append(target)
    return out


def lib_path(name="libVectorSimilarity.so"):
    return os.path.join(LIBDIR, name)


if __name__ == "__main__":
    for t in build(force="--force" in sys.argv, verbose=True):
        print("built", t)
