#!/bin/bash
# First GPU session: environment facts, parity tests, bench, rocprof of the bench, scan tuning sweep.
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
{
  echo "== host"; nproc; free -g | head -2; lscpu | grep -E "Model name|Socket|Thread|Core" 
  echo "== gpu"; rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8; rocm-smi --showmeminfo vram 2>/dev/null | head -8
} > gpurun_out/env.txt 2>&1
export TMPDIR=/tmp
echo "== pytest gpu"
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.txt
echo "== bench"
timeout 900 python bench.py --steps 200 --warmup 10 2>&1 | tail -5 | tee gpurun_out/bench_1gpu.txt
echo "== tune"
timeout 600 python scripts/tune_scan.py 2>&1 | tail -45 | tee gpurun_out/tune_scan.txt
echo "== rocprof"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_bench" -o bench -- python "$OLDPWD/bench.py" --steps 60 --warmup 5 --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof_bench.log" 2>&1)
ls -R gpurun_out/prof_bench | head -20
find gpurun_out/prof_bench -name "*kernel_stats*" | head -2 | xargs -I{} head -15 {}
