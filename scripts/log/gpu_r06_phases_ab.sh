#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for ph in 0 3 0 3; do
  echo "== fp16 cfg3 QS_PHASES=$ph"
  QS_PHASES=$ph REPS=12 OUT_TAG=ph${ph}_ timeout 600 python scripts/bench_batch.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('device_ms_per_batch','ms_per_batch_wall','hbm_frac_of_8TBs','pipelined_qps_wall')})"
done
