#!/bin/bash
# round 4, call e: the reworked reduce kernel + completion flags: parity, trace, knob A/B on the stream
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hybrid_tiles.py tests/test_gpu_hybrid_query.py -x -q -p no:cacheprovider > gpurun_out/r04e_tests.txt 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r04e_tests.txt
timeout 600 python scripts/hybrid_trace_stream.py 2>&1 | tail -2 | cut -c1-2500 > gpurun_out/r04e_trace.txt; cat gpurun_out/r04e_trace.txt
timeout 900 python scripts/bench_hybrid_stream.py > gpurun_out/r04e_stream.txt 2>&1; echo "stream rc=$?"
tail -7 gpurun_out/r04e_stream.txt
