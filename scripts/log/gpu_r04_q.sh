#!/bin/bash
# round 4, call q: RSGPU_EvalTree through the general tile kernel (parity with the staged evaluation, the tree / iterator tests
# that now take it), the configs[4] stream forced through the general kernel (what its generality costs)
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_hybrid_general.py tests/test_gpu_tree.py tests/test_gpu_iterators.py tests/test_gpu_reference_hybrid_reader.py tests/test_gpu_hybrid_tiles.py tests/test_gpu_proximity.py -q -p no:cacheprovider --timeout 150 -rf > gpurun_out/r04q_tests.txt 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/r04q_tests.txt | cut -c1-300
CYCLES=3 timeout 400 python scripts/bench_hybrid_general.py > gpurun_out/r04q_shapes.txt 2>&1; echo "shapes rc=$?"
tail -7 gpurun_out/r04q_shapes.txt | cut -c1-500
