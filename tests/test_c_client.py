"""The drop-in boundary is usable from plain C: examples/knn_example.c includes only include/*.h, links
libVectorSimilarity.so and makes the calls src/vector_index.c / hybrid_reader.c make.  On a box without a GPU the
library must fail loudly (VecSimIndex_New returns NULL with a message), never fall back to a CPU path."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_example(tmp_path, name="knn_example"):
    from redisearch_amd import vecsim as V
    V.load()                                   # builds the library if the tree is fresh
    libdir = os.path.join(ROOT, "redisearch_amd", "lib")
    exe = str(tmp_path / name)
    subprocess.run(["gcc", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", name + ".c"), "-L" + libdir, "-lVectorSimilarity",
                    "-Wl,-rpath," + libdir, "-o", exe], check=True, timeout=120)
    return exe


def test_c_client_links_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = build_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    if torch.cuda.is_available():
        assert r.returncode == 0
    else:
        assert r.returncode == 2 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_c_client_runs(tmp_path):
    exe = build_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert [ln.split()[0] for ln in r.stdout.split("\n") if ln] == ["100", "99", "101", "98", "102"]
    assert [float(ln.split()[1]) for ln in r.stdout.split("\n") if ln] == [0.0, 4.0, 4.0, 16.0, 16.0]


def test_hybrid_tree_c_client_links_and_fails_loudly_without_gpu(tmp_path):
    """examples/hybrid_tree_example.c: the hybrid path from plain C -- raw doc-id lists, RSGPU_HybridTreeQuery over
    `hello (world|words) -spam`"""
    import torch
    exe = build_example(tmp_path, "hybrid_tree_example")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        assert r.returncode == 0, (r.stdout, r.stderr)
    else:
        assert r.returncode == 2 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_hybrid_tree_c_client_runs_and_agrees_with_the_python_binding(tmp_path):
    import numpy as np
    from redisearch_amd import search as S
    from redisearch_amd import vecsim as V
    exe = build_example(tmp_path, "hybrid_tree_example")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    lines = [ln.split() for ln in r.stdout.split("\n") if ln]
    n_docs = 200_000
    ids = np.arange(1, n_docs + 1)
    hit = (ids % 2 == 0) & ((ids % 3 == 0) | (ids % 5 == 0)) & (ids % 7 != 0)
    assert lines[0] == ["hits", str(int(hit.sum())), "path", "2"]
    assert [ln[1] for ln in lines if ln[0] == "knn"] == ["3000", "3006", "2994", "2990"]
    # the same query through the ctypes binding (lists in the FreqsOnly codec with frequency 1: what the raw codec yields)
    import oracle as O

    def mk(step):
        d = np.arange(step, n_docs + 1, step).astype(np.uint64)
        ii = O.InvertedIndex(O.C_FREQS_ONLY)
        ii.add_many(d, np.ones(d.size, np.uint32))
        return S.Postings.from_flat(ii.flatten()), d.size
    built = [mk(s_) for s_ in (2, 3, 5, 7)]
    g, sizes = [b[0] for b in built], [b[1] for b in built]
    table = S.DocTable((50 + np.arange(n_docs + 1) % 100).astype(np.uint32), np.ones(n_docs + 1, np.float32))
    bidf = [S.calculate_idf_bm25(n_docs, s_) for s_ in sizes[:3]] + [0.0]
    hq = S.HybridTreeQuery(S.OP_INTERSECT, [(S.OP_TERM, 1.0, g[:1]), (S.OP_UNION, 1.0, g[1:3]), (S.OP_NOT, 1.0, g[3:])], table=table,
                           scorer="BM25STD", idf=[0.0] * 4, bm25_idf=bidf, weight=[1.0, 1.0, 1.0, 0.0], num_docs=n_docs, avg_doc_len=99.5,
                           top_n=5)
    hq.run()
    want = hq.results()
    assert [int(ln[1]) for ln in lines if ln[0] == "top"] == want["top"][0].tolist()
    assert [float(ln[2]) for ln in lines if ln[0] == "top"] == want["top"][1].tolist()
