#!/bin/bash
# the pthread-callers legs again and again, with and without the profiler (after one SIGSEGV inside the HIP runtime under rocprofv3)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_gpu_hybrid_concurrent.py tests/test_gpu_hybrid_coalesce.py tests/test_gpu_concurrency.py -x -q -m gpu 2>&1 | tail -1
done
for i in 1 2 3; do
  CODEC=freqs_only MODES=warm CONFIGS="d:" THREADS=8,16 CYCLES=2 OUT=stress_$i.json timeout 600 python scripts/bench_hybrid_stream.py > gpurun_out/stress_plain_$i.log 2>&1; echo "plain $i rc=$? segv=$(grep -c SIGSEGV gpurun_out/stress_plain_$i.log)"
done
for i in 1 2 3 4 5 6; do
  (cd /tmp && CODEC=freqs_only MODES=warm CONFIGS="d:" THREADS=8,16 CYCLES=2 OUT=stress_p$i.json timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stress_prof_$i -o b -- python "$GRAFT_REPO_ROOT/scripts/bench_hybrid_stream.py" > "$GRAFT_REPO_ROOT/gpurun_out/stress_prof_$i.log" 2>&1); echo "rocprof $i rc=$? segv=$(grep -c SIGSEGV gpurun_out/stress_prof_$i.log)"
done
