#!/bin/bash
# ONE gpurun call for the experiments prepared at the end of round 2 (each was compiled and statically checked, none was run):
#   1. producer-wave variant of the query-stationary int8 pass   (scripts/diag/gemm_qs_producer_waves.patch, knob qs_pw)
#   2. BFLOAT16 indexes through the int8-shadow batched passes    (scripts/diag/bf16_int8_shadow.patch, knob shadow8_bf16)
# applied together on the box's copy (they touch different kernels; the two shared lines patch with fuzz), built once.
# The overlapped tile boundary edits the same launch switch as (1): it has its own call, scripts/diag/ovl_diagnose.sh.
#   gpurun --timeout 1500 -- 'bash scripts/diag/round3_first_call.sh'
set -u
mkdir -p gpurun_out
(patch -p1 < scripts/diag/gemm_qs_producer_waves.patch && patch -p1 -F3 < scripts/diag/bf16_int8_shadow.patch) > gpurun_out/r3_patch.txt 2>&1 || { echo "patches do not apply"; cat gpurun_out/r3_patch.txt; exit 1; }
python -c "from redisearch_amd import build; build.build()" > gpurun_out/r3_build.txt 2>&1 || { echo "build failed"; tail -5 gpurun_out/r3_build.txt; exit 1; }
export PYTHONPATH=scripts/diag${PYTHONPATH:+:$PYTHONPATH}
# knobs off: the patched library must be the shipped one
timeout 400 python -m pytest tests/test_gpu_batch_i8_shadow.py tests/test_gpu_batch_qs.py -x -q -p no:cacheprovider > gpurun_out/r3_regress.txt 2>&1; echo "regression (knobs off): rc=$?"; tail -2 gpurun_out/r3_regress.txt
RSGPU_TUNING=qs_pw=1 timeout 300 python -m pytest -p tuning_plugin tests/test_gpu_batch_i8_shadow.py -x -q -p no:cacheprovider > gpurun_out/r3_pw_parity.txt 2>&1
rc=$?; echo "producer waves, parity: rc=$rc"; tail -3 gpurun_out/r3_pw_parity.txt
if [ $rc -eq 0 ]; then
  KNOB=qs_pw VALUES=0,1 I8_SHADOW=1 ROUNDS=3 REPS=8 timeout 600 python scripts/bench_batch_knob.py > gpurun_out/r3_pw_ab.json 2> gpurun_out/r3_pw_ab.err
  echo "producer waves, A/B: rc=$?"; tail -c 1200 gpurun_out/r3_pw_ab.json
fi
timeout 600 python scripts/diag/bf16_i8_shadow_check.py > gpurun_out/r3_bf16_check.txt 2>&1; echo "bf16 int8 shadow: rc=$?"; tail -10 gpurun_out/r3_bf16_check.txt
