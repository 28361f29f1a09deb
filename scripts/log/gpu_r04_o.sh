#!/bin/bash
# round 4, call o: the general hybrid tile kernel -- parity (new tests, all of them: no -x), the hybrid / tree / proximity tests
# next to it (the proximity device functions moved into a shared header, the two-launch host code was split into helpers),
# then the stream + the general shapes A/B
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_gpu_hybrid_general.py -q -p no:cacheprovider --timeout 120 -rf > gpurun_out/r04o_new.txt 2>&1; echo "new rc=$?"
tail -5 gpurun_out/r04o_new.txt | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_hybrid_tiles.py tests/test_gpu_hybrid_query.py tests/test_gpu_tree.py tests/test_gpu_proximity.py tests/test_gpu_over_limit.py -x -q -p no:cacheprovider --timeout 150 > gpurun_out/r04o_old.txt 2>&1; echo "old rc=$?"
tail -3 gpurun_out/r04o_old.txt | cut -c1-300
CYCLES=3 timeout 400 python scripts/bench_hybrid_general.py > gpurun_out/r04o_shapes.txt 2>&1; echo "shapes rc=$?"
tail -6 gpurun_out/r04o_shapes.txt | cut -c1-600
