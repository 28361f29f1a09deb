#!/bin/bash
# Producer-wave variant of the query-stationary int8 pass (scripts/diag/gemm_qs_producer_waves.patch): apply on
# the GPU box's copy of the tree, rebuild, (1) parity: the int8-shadow batched tests with the knob ON -- they demand bit-identical
# answers to single queries, (2) A/B inside one process and one index on BASELINE configs[2] with the int8 shadow.
# Not to be combined with the overlapped-boundary patch (both edit the same launch switch).   bash scripts/diag/pw_experiment.sh
set -u
mkdir -p gpurun_out
patch -p1 < scripts/diag/gemm_qs_producer_waves.patch > gpurun_out/pw_patch.txt 2>&1 || { echo "patch does not apply"; exit 1; }
python -c "from redisearch_amd import build; build.build()" > gpurun_out/pw_build.txt 2>&1 || { echo "build failed"; tail -5 gpurun_out/pw_build.txt; exit 1; }
export PYTHONPATH=scripts/diag${PYTHONPATH:+:$PYTHONPATH}
RSGPU_TUNING=qs_pw=1 timeout 300 python -m pytest -p tuning_plugin tests/test_gpu_batch_i8_shadow.py -x -q -p no:cacheprovider > gpurun_out/pw_parity.txt 2>&1
rc=$?; echo "parity with qs_pw=1: rc=$rc"; tail -3 gpurun_out/pw_parity.txt
[ $rc -eq 0 ] || exit $rc
KNOB=qs_pw VALUES=0,1 I8_SHADOW=1 ROUNDS=3 REPS=8 timeout 600 python scripts/bench_batch_knob.py > gpurun_out/pw_ab.json 2> gpurun_out/pw_ab.err
echo "A/B: rc=$?"; tail -c 1500 gpurun_out/pw_ab.json
