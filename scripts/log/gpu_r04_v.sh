#!/bin/bash
# round 4, call v: windowed RSGPU_IntersectEx through the tile kernel: parity of its consumers, timing
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_hybrid_general.py tests/test_gpu_proximity.py tests/test_gpu_iterators.py tests/test_gpu_reference_hybrid_reader.py tests/test_gpu_search.py tests/test_gpu_intersection_kats.py tests/test_gpu_over_limit.py tests/test_gpu_docid64.py -q -p no:cacheprovider --timeout 150 -rf > gpurun_out/r04v_tests.txt 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r04v_tests.txt | cut -c1-300
SKIP_STREAM=1 timeout 300 python scripts/bench_hybrid_general.py > gpurun_out/r04v_shapes.txt 2>&1; echo "shapes rc=$?"
grep -E "^(eval_tree|phrase)" gpurun_out/r04v_shapes.txt | cut -c1-330
