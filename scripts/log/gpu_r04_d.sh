#!/bin/bash
# round 4, call d: where the hybrid query's time goes on the distinct stream (tile phases by tile class, reduce phases)
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/hybrid_trace_stream.py > gpurun_out/r04d_trace.txt 2>&1; echo "trace rc=$?"
tail -9 gpurun_out/r04d_trace.txt
timeout 600 python -m pytest tests/test_gpu_hybrid_tiles.py -x -q -p no:cacheprovider 2>&1 | tail -3
