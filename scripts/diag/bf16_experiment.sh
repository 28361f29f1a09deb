#!/bin/bash
# BFLOAT16 indexes through the int8-shadow batched passes: apply scripts/diag/bf16_int8_shadow.patch on the
# GPU box's copy, rebuild, run the bit-identity check, then the existing FLOAT16 / FLOAT32 int8-shadow tests as regression.
set -u
mkdir -p gpurun_out
patch -p1 < scripts/diag/bf16_int8_shadow.patch > gpurun_out/bf16_patch.txt 2>&1 || { echo "patch does not apply"; exit 1; }
python -c "from redisearch_amd import build; build.build()" > gpurun_out/bf16_build.txt 2>&1 || { echo "build failed"; tail -5 gpurun_out/bf16_build.txt; exit 1; }
timeout 600 python scripts/diag/bf16_i8_shadow_check.py > gpurun_out/bf16_check.txt 2>&1; echo "bf16 check: rc=$?"; tail -12 gpurun_out/bf16_check.txt
timeout 300 python -m pytest tests/test_gpu_batch_i8_shadow.py -x -q -p no:cacheprovider > gpurun_out/bf16_regress.txt 2>&1; echo "regression: rc=$?"; tail -2 gpurun_out/bf16_regress.txt
