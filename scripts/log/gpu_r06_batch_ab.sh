#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_batch_qs.py tests/test_gpu_batch_h8.py tests/test_gpu_batch_f8.py tests/test_gpu_batch_l2.py tests/test_gpu_batch_f32_native.py tests/test_gpu_batch_f32_shadow.py tests/test_gpu_batch_i8_shadow.py tests/test_gpu_coalesce_wide.py tests/test_gpu_two_stage.py tests/test_gpu_coalesce.py tests/test_gpu_coalesce_types.py tests/test_gpu_select_paths.py tests/test_gpu_over_limit.py -x -q -m gpu 2>&1 | tail -3
for t in "" "batch_prune=0"; do
  echo "== fp16 cfg3 TUNING=$t"
  TUNING=$t REPS=12 OUT_TAG=prune_${t:-on}_ timeout 600 python scripts/bench_batch.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('device_ms_per_batch','ms_per_batch_wall','hbm_frac_of_8TBs','pipelined_qps_wall')})"
  echo "== fp32 cosine TUNING=$t"
  TUNING=$t REPS=8 METRICS=cosine,l2 SHAPES=2 timeout 600 python scripts/bench_batch_f32.py 2>&1 | grep "^cosine\|^l2" | cut -c1-330
done
