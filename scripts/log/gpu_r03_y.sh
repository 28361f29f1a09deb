#!/bin/bash
# round 3, call y: the hybrid query from n caller threads at once
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2
timeout 600 python scripts/bench_hybrid_concurrent.py > gpurun_out/r03y_hybrid_concurrent.txt 2> gpurun_out/r03y_hybrid_concurrent.err; echo "rc=$?"
tail -3 gpurun_out/r03y_hybrid_concurrent.err
cat gpurun_out/r03y_hybrid_concurrent.txt
