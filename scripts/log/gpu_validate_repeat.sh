#!/bin/bash
# Repeats a set of GPU test files N times on ONE box and keeps every failing log (round 3: re-validating the overlapped tile
# boundary of the query-stationary pass, scripts/diag/gemm_qs_overlapped_tile_boundary.patch -- see DESIGN.md 8).
#   FILES="tests/test_gpu_batch_i8_shadow.py tests/test_gpu_batch_qs.py" N=20 bash scripts/gpu_validate_repeat.sh
set -u
mkdir -p gpurun_out
FILES=${FILES:-"tests/test_gpu_batch_i8_shadow.py tests/test_gpu_batch_qs.py tests/test_gpu_batch.py tests/test_gpu_batch_f32_shadow.py"}
N=${N:-20}
(hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i unique) > gpurun_out/repeat_box_id.txt 2>&1
ok=0; bad=0
for i in $(seq 1 "$N"); do
  if timeout 120 python -m pytest $FILES -q -m gpu -x -p no:cacheprovider > gpurun_out/repeat_$i.txt 2>&1; then
    ok=$((ok + 1)); rm -f gpurun_out/repeat_$i.txt
  else
    bad=$((bad + 1)); echo "run $i failed (rc=$?)"; grep -m3 -i "fault\|Fatal Python\|Timeout\|failed" gpurun_out/repeat_$i.txt
  fi
done
echo "repeat: $ok ok, $bad failed of $N on $(tr '\n' ' ' < gpurun_out/repeat_box_id.txt)"
