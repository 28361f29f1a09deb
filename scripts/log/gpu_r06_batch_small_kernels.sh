#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_batch_qs.py tests/test_gpu_batch_h8.py tests/test_gpu_batch_f8.py tests/test_gpu_batch_l2.py tests/test_gpu_batch_f32_native.py tests/test_gpu_batch_f32_shadow.py tests/test_gpu_batch_i8_shadow.py tests/test_gpu_coalesce_wide.py -x -q -m gpu 2>&1 | tail -4
TAG=r06b bash scripts/diag/batch_timeline.sh 2>&1 | grep -v "^ *[0-9.]* us" | tail -5
tail -2 gpurun_out/r06b_tl_f16.log | cut -c1-600
tail -2 gpurun_out/r06b_tl_f32.log | cut -c1-600
