#!/bin/bash
# round 4, call s: NOT children, BM25STD.NORM on the tile paths, the intersection sort key -- the hybrid / tree / iterator tests
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_hybrid_general.py tests/test_gpu_tree.py tests/test_gpu_iterators.py tests/test_gpu_reference_hybrid_reader.py tests/test_gpu_hybrid_tiles.py tests/test_gpu_hybrid_query.py tests/test_gpu_bm25std_norm.py tests/test_gpu_proximity.py tests/test_gpu_over_limit.py tests/test_gpu_boolean.py -q -p no:cacheprovider --timeout 150 -rf > gpurun_out/r04s_tests.txt 2>&1; echo "tests rc=$?"
tail -12 gpurun_out/r04s_tests.txt | cut -c1-400
