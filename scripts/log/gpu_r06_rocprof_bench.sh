#!/bin/bash
# rocprofv3 kernel stats of the driver's bench command (twice: the first attempt of the final tree died inside the HIP runtime under
# the profiler, in the pthread-callers leg -- see profiles/README.md)
export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out
for i in a b; do
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r06_prof_stats_final5$i" -o b -- python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-callers-extra > "$R/gpurun_out/r06_prof_stats_final5$i.log" 2>&1); echo "run $i rc=$?"
  f=$(find gpurun_out/r06_prof_stats_final5$i -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_bench_kernel_stats_final5$i.csv
  grep "^{\"metric\"" gpurun_out/r06_prof_stats_final5$i.log | tail -1 > gpurun_out/r06_bench_under_rocprof_final5$i.json
  grep -c "SIGSEGV" gpurun_out/r06_prof_stats_final5$i.log
done
find gpurun_out -name "*kernel_trace.csv" -size +1M -delete; find gpurun_out -name "*.db" -delete
