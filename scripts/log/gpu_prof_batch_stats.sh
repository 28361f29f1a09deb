#!/bin/bash
# rocprofv3 kernel stats of the batched-query bench (one pass, no PMCs)
set -u
export TMPDIR=/tmp REPS=${REPS:-6}
R=$(pwd); mkdir -p gpurun_out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_batch_stats" -o b -- python "$R/scripts/bench_batch.py" > "$R/gpurun_out/prof_batch_stats.log" 2>&1)
tail -1 gpurun_out/prof_batch_stats.log | cut -c1-300
find gpurun_out/prof_batch_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "cut -c1-200 {} | head -14"
find gpurun_out -name "*kernel_trace.csv" -size +4M -delete; find gpurun_out -name "*.db" -delete
