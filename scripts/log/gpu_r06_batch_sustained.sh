#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for t in "" "batch_prune=0" "gemm_qs_f8=0"; do
  echo "== fp32 cosine sustained TUNING=$t"
  TUNING=$t REPS=300 METRICS=cosine SHAPES=2 timeout 600 python scripts/bench_batch_f32.py 2>&1 | grep "^cosine" | cut -c1-200
done
echo "== fp16 cfg3 sustained"
REPS=300 timeout 600 python scripts/bench_batch.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('device_ms_per_batch','ms_per_batch_wall','hbm_frac_of_8TBs','pipelined_qps_wall')})"
