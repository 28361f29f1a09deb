#!/bin/bash
# round 4, call j: IP through the fp32-native passes, the four-wave default, over-limit, coalescer; short timeouts
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_batch_f32_native.py tests/test_gpu_over_limit.py tests/test_gpu_coalesce_wide.py tests/test_gpu_batch_l2.py tests/test_gpu_batch_f32_shadow.py -q -p no:cacheprovider --timeout 150 > gpurun_out/r04j.txt 2>&1; echo "rc=$?"
tail -15 gpurun_out/r04j.txt
