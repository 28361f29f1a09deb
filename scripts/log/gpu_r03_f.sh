#!/bin/bash
set -u
mkdir -p gpurun_out
T0=$(date +%s)
export PYTHONPATH=scripts/diag${PYTHONPATH:+:$PYTHONPATH}
RSGPU_TUNING=qs_pw=1 timeout 300 python -m pytest -p tuning_plugin tests/test_gpu_batch_i8_shadow.py -x -q -p no:cacheprovider > gpurun_out/r03f_pw_parity.txt 2>&1
rc=$?; echo "producer waves, parity: rc=$rc t=$(( $(date +%s) - T0 ))"; tail -5 gpurun_out/r03f_pw_parity.txt
if [ $rc -eq 0 ]; then
  KNOB=qs_pw VALUES=0,1 I8_SHADOW=1 ROUNDS=3 REPS=8 timeout 600 python scripts/bench_batch_knob.py > gpurun_out/r03f_pw_ab.json 2> gpurun_out/r03f_pw_ab.err
  echo "producer waves, A/B: rc=$? t=$(( $(date +%s) - T0 ))"; tail -c 900 gpurun_out/r03f_pw_ab.json
fi
