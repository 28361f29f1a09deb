#!/bin/bash
# round 4, call k: split tile geometry of the hybrid query: parity, knob A/B on the stream, trace
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_hybrid_tiles.py tests/test_gpu_hybrid_query.py -x -q -p no:cacheprovider --timeout 150 > gpurun_out/r04k_tests.txt 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r04k_tests.txt
CYCLES=3 timeout 600 python scripts/bench_hybrid_stream.py > gpurun_out/r04k_stream.txt 2>&1; echo "stream rc=$?"
tail -7 gpurun_out/r04k_stream.txt | cut -c1-700
timeout 300 python scripts/hybrid_trace_stream.py 2>&1 | tail -1 | cut -c1-2200 > gpurun_out/r04k_trace.txt; cat gpurun_out/r04k_trace.txt
