#!/bin/bash
# round 4, call n: the KNN phase of the tile kernel software-pipelined: parity, A/B on the stream
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_hybrid_tiles.py tests/test_gpu_hybrid_query.py -x -q -p no:cacheprovider --timeout 150 > gpurun_out/r04n_tests.txt 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r04n_tests.txt
CYCLES=3 timeout 600 python scripts/bench_hybrid_stream.py > gpurun_out/r04n_stream.txt 2>&1; echo "stream rc=$?"
tail -6 gpurun_out/r04n_stream.txt | cut -c1-420
