# rocprofv3 kernel stats of the driver's bench command on the final tree (the first step of scripts/gpu_prof.sh)
set -u
export TMPDIR=/tmp
R=$(pwd); mkdir -p gpurun_out/r05y
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r05y/prof_stats" -o b -- python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-callers-extra > "$R/gpurun_out/r05y/prof_stats.log" 2>&1)
grep "^{\"metric\"" gpurun_out/r05y/prof_stats.log | tail -1 > gpurun_out/r05y/bench_under_rocprof.json
f=$(find gpurun_out/r05y/prof_stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05y/bench_kernel_stats.csv; cut -c1-200 "$f" | head -24
find gpurun_out/r05y -name "*kernel_trace.csv" -size +1M -delete; find gpurun_out/r05y -name "*.db" -delete
cut -c1-300 gpurun_out/r05y/bench_under_rocprof.json
