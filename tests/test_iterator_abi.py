"""Boundary 3: include/rs_iterator.h restates the reference's QueryIterator vtable and
redisearch_amd/lib/librsgpu_iterators.so serves it.  CPU-side checks: the library loads next to the engine and exports
every entry point the header declares, the new record-access entry points of rsgpu_search.h exist, and -- where
/root/reference is present -- the restated structs / enums are layout-identical to the reference's own headers."""
import ctypes as C
import os
import re
import shutil
import subprocess

import pytest

from redisearch_amd import build as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFROOT = "/root/reference"


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(RSGPU_[A-Za-z0-9_]+)\s*\(", txt)))


def test_iterator_library_exports_what_the_header_declares():
    B.build()
    C.CDLL(B.lib_path(), mode=C.RTLD_GLOBAL)
    lib = C.CDLL(B.lib_path("librsgpu_iterators.so"))
    names = [n for n in _declared("rs_iterator.h")]
    assert {"RSGPU_NewIntersectionIterator", "RSGPU_NewUnionIterator", "RSGPU_NewNotIterator", "RSGPU_NewHitsIterator",
            "RSGPU_Iterator_Hits", "RSGPU_Iterators_SetResultAPI", "RSGPU_Iterators_LastError",
            "RSGPU_Iterators_SetBlock"} <= set(names)
    for n in names:
        assert hasattr(lib, n), n


def test_record_access_entry_points_exist():
    lib = C.CDLL(B.lib_path())
    for n in ("RSGPU_Postings_Codec", "RSGPU_Hits_NumLeaves", "RSGPU_Hits_IsUnion", "RSGPU_Hits_LeafOrder",
              "RSGPU_Hits_ReadRange", "RSGPU_Hits_ReadRecords", "RSGPU_Postings_ReadBytes"):
        assert n in _declared("rsgpu_search.h") and hasattr(lib, n), n


def test_constructors_fail_cleanly_without_a_result_api_or_lists():
    """No GPU needed: argument checks come first, and a process without the module's constructors is reported."""
    C.CDLL(B.lib_path(), mode=C.RTLD_GLOBAL)
    lib = C.CDLL(B.lib_path("librsgpu_iterators.so"))
    lib.RSGPU_NewIntersectionIterator.restype = C.c_void_p
    lib.RSGPU_NewIntersectionIterator.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_bool, C.c_double]
    lib.RSGPU_Iterators_LastError.restype = C.c_char_p
    assert lib.RSGPU_NewIntersectionIterator(None, 0, -1, False, 1.0) is None
    assert b"terms" in lib.RSGPU_Iterators_LastError()
    lib.RSGPU_Iterator_Hits.restype, lib.RSGPU_Iterator_Hits.argtypes = C.c_void_p, [C.c_void_p]
    assert lib.RSGPU_Iterator_Hits(None) is None


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFROOT, "src")) or shutil.which("gcc") is None,
                    reason="needs /root/reference")
def test_layout_matches_reference_headers(tmp_path):
    probe = os.path.join(ROOT, "tests", "iter_layout_probe.c")
    mine, ref = str(tmp_path / "mine"), str(tmp_path / "ref")
    subprocess.check_call(["gcc", "-std=gnu11", "-I" + os.path.join(ROOT, "include"), probe, "-o", mine])
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-D_GNU_SOURCE", "-DPROBE_REFERENCE", "-w",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "ref_compile_stubs"),
                           "-Isrc", "-Isrc/iterators", "-Isrc/index_result", "-Ideps", "-Ideps/rmalloc", "-Isrc/redisearch_rs/headers",
                           "-Ideps/rmutil", "-I.", probe, "-o", ref], cwd=REFROOT)
    a, b = subprocess.check_output([mine], text=True), subprocess.check_output([ref], text=True)
    assert a == b and a.count("\n") >= 20, (a, b)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFROOT, "src")), reason="needs /root/reference")
def test_result_api_names_exist_in_the_reference_ffi():
    """Every constructor the RSGPU_ResultAPI table binds by name is declared by the reference's types_ffi.h."""
    ffi = open(os.path.join(REFROOT, "src/redisearch_rs/headers/types_ffi.h")).read()
    hdr = open(os.path.join(ROOT, "include", "rs_iterator.h")).read()
    table = re.search(r"typedef struct RSGPU_ResultAPI \{(.*?)\} RSGPU_ResultAPI;", hdr, re.S).group(1)
    names = re.findall(r"\(\*([A-Za-z_]+)\)", table)
    assert len(names) == 8
    for n in names:
        assert re.search(r"\b" + n + r"\(", ffi), n
