#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_search.py -x -v -p no:cacheprovider > gpurun_out/pytest_search_full.txt 2>&1
echo "search rc=$?"
grep -nE "PASSED|FAILED|ERROR|fault|Fatal|Abort|Error" gpurun_out/pytest_search_full.txt | head -60
timeout 900 python -m pytest tests/test_gpu_flat.py -x -q -p no:cacheprovider > gpurun_out/pytest_flat_full.txt 2>&1
echo "flat rc=$?"
tail -5 gpurun_out/pytest_flat_full.txt
for i in 1 2 3; do timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value'],2), round(d['roofline']['achieved'],1), round(d['roofline']['avg_kernel_ms'],4), d['config']['p50_ms'])"; done
