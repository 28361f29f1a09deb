#!/bin/bash
# round 4, call a: the fp32-native matrix-core passes -- parity first, then timing at 10M x 768 (two kernel shapes)
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2
timeout 900 python -m pytest tests/test_gpu_batch_f32_native.py -x -q -p no:cacheprovider > gpurun_out/r04a_tests.txt 2>&1; echo "tests rc=$?"
tail -25 gpurun_out/r04a_tests.txt
timeout 600 python scripts/bench_batch_f32.py > gpurun_out/r04a_bench.txt 2>&1; echo "bench rc=$?"
tail -12 gpurun_out/r04a_bench.txt
