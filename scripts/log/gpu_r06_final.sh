#!/bin/bash
# round 6, final tree: the GPU suite, the bench with the driver's arguments, rocprofv3 kernel stats of the same command
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r06_gpu_suite_final8.txt; cat gpurun_out/r06_gpu_suite_final8.txt
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_1gpu_final8.json 2> gpurun_out/r06_bench_final8.err; cp gpurun_out/bench_full.json gpurun_out/r06_bench_1gpu_final8_full.json; cut -c1-400 gpurun_out/r06_bench_1gpu_final8.json
export TMPDIR=/tmp; R=$(pwd)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r06_prof_stats_final8" -o b -- python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-callers-extra > "$R/gpurun_out/r06_prof_stats_final8.log" 2>&1)
f=$(find gpurun_out/r06_prof_stats_final8 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r06_bench_kernel_stats_final8.csv; cut -c1-150 "$f" | head -8
grep "^{\"metric\"" gpurun_out/r06_prof_stats_final8.log | tail -1 > gpurun_out/r06_bench_under_rocprof_final8.json
find gpurun_out -name "*kernel_trace.csv" -size +1M -delete; find gpurun_out -name "*.db" -delete
