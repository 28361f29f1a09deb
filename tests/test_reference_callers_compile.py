"""Boundary 1 proof: every reference C file that includes a VecSim header compiles, unchanged, against
THIS repository's include/VecSim/*.h.

The reference's deps/VectorSimilarity submodule is empty, so `include/VecSim` is the only definition of that
interface the reference's callers can see.  `gcc -fsyntax-only` is run over each caller where it lies under
/root/reference (nothing is copied); deps/hiredis is an empty submodule too, so tests/ref_compile_stubs/hiredis
holds a few declarations for the coordinator headers to parse.  Skipped where /root/reference is absent (the
GPU box).  The bar: zero errors, and zero diagnostics of any kind that mention a VecSim identifier (an implicit
declaration of a VecSim function would be one).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
STUBS = os.path.join(ROOT, "tests", "ref_compile_stubs")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")) or shutil.which("gcc") is None,
                                reason="needs /root/reference and gcc")

INC = ["src", "deps", "deps/rmalloc", "deps/rmutil", "deps/thpool", "src/redisearch_rs/headers", ".", "src/coord",
       "src/coord/rmr", "src/buffer", "src/wildcard", "src/inverted_index", "src/ttl_table", "src/geometry", "src/trie",
       "src/value", "src/iterators", "src/util/hash"]

# files the round-1 verdict names explicitly; the discovery below must find every one of them
NAMED = ["src/spec.c", "src/debug_commands.c", "src/iterators/hybrid_reader.c", "src/vector_index.c", "src/document.c",
         "src/indexer.c", "src/disk_indexer.c", "src/field_spec.c", "src/module-init/module-init.c", "src/util/workers.c",
         "src/info/info_command.c", "src/info/field_spec_info.c", "src/info/global_stats.c", "src/info/indexes_info.c",
         "src/query.c"]

VECSIM_IDENT = re.compile(r"VecSim|VECSIM_|SVS_VAMANA|HNSW_DEFAULT|INFOFIELD_|BFParams|HNSWParams|SVSParams|"
                          r"TieredIndexParams|AlgoParams|labelType|DEFAULT_BLOCK_SIZE")
# the reference's own mirror enum (src/vector_index.h:130-143) assigned to VecSearchMode: theirs, not the header's
ALLOWED = re.compile(r"implicit conversion from 'VecSimSearchMode' to 'VecSearchMode'")


def _callers():
    out = subprocess.run(["grep", "-rlE", r'#include\s+"VecSim/', "src", "--include=*.c"], cwd=REF,
                         capture_output=True, text=True).stdout.split()
    # field_spec.c reaches the ABI through field_spec.h (VecSimIndex_Free at :49)
    out = sorted(set(out) | {"src/field_spec.c"})
    return out


def _compile(rel, extra=()):
    cmd = ["gcc", "-fsyntax-only", "-std=gnu11", "-D_GNU_SOURCE", "-DREDISMODULE_SDK_RLEC", "-Wall",
           "-I" + os.path.join(ROOT, "include"), "-I" + STUBS] + ["-I" + i for i in INC] + list(extra) + [rel]
    r = subprocess.run(cmd, cwd=REF, capture_output=True, text=True)
    return r.returncode, r.stderr


def test_discovery_covers_the_named_callers():
    found = set(_callers())
    assert len(found) >= 25
    for f in NAMED:
        assert f in found, f


@pytest.mark.parametrize("rel", _callers() if os.path.isdir(os.path.join(REF, "src")) else [])
def test_reference_caller_compiles_against_our_headers(rel):
    rc, err = _compile(rel)
    errors = [l for l in err.splitlines() if re.search(r"\berror\b", l)]
    assert rc == 0 and not errors, "\n".join(errors[:20])
    bad = [l for l in err.splitlines()
           if re.search(r"warning|error", l) and VECSIM_IDENT.search(l) and not ALLOWED.search(l)]
    assert not bad, "\n".join(bad[:20])


def test_headers_are_self_contained_c_and_cpp(tmp_path):
    """Each header parses on its own, as C11 and as C++17 (bindgen and the C++ tests include them singly)."""
    for h in ("vec_sim_common.h", "query_results.h", "info_iterator.h", "vec_sim.h", "vec_sim_debug.h"):
        src = tmp_path / ("t_" + h.replace(".h", ".c"))
        src.write_text('#include "VecSim/%s"\nint main(void){return 0;}\n' % h)
        for cc, std in (("gcc", "-std=c11"), ("g++", "-std=c++17")):
            lang = ["-x", "c++"] if cc == "g++" else []
            r = subprocess.run([cc, "-fsyntax-only", std, "-Wall", "-Wextra", "-pedantic", "-I" + os.path.join(ROOT, "include")]
                               + lang + [str(src)], capture_output=True, text=True)
            assert r.returncode == 0 and "warning" not in r.stderr, (h, cc, r.stderr)


def test_disk_context_and_svs_defaults_as_spec_c_uses_them(tmp_path):
    """reference src/spec.c:1195-1240,2912-2924: designated initialiser with .userData, the SVS_VAMANA_DEFAULT_* macros;
    src/hybrid/parse/hybrid_callbacks.c:442: VECSIM_POLICY_ADHOC_BF; values pinned by tests/pytests/test_vecsim.py:357-360."""
    src = tmp_path / "t.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "VecSim/vec_sim.h"
int main(void) {
  int x = 0;
  VecSimDiskContext c = (VecSimDiskContext){.storage = &x, .indexName = "v", .indexNameLen = 1, .userData = &x, .rerank = true};
  printf("%d %d %d %d %s %d\n", (int)SVS_VAMANA_DEFAULT_GRAPH_MAX_DEGREE, (int)SVS_VAMANA_DEFAULT_CONSTRUCTION_WINDOW_SIZE,
         (int)SVS_VAMANA_DEFAULT_LEANVEC_DIM, (int)SVS_VAMANA_DEFAULT_TRAINING_THRESHOLD, VECSIM_POLICY_ADHOC_BF,
         c.userData == c.storage);
  return 0;
}''')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=gnu11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split()
    assert out == ["32", "200", "0", "10240", "adhoc_bf", "1"]
