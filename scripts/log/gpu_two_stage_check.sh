#!/bin/bash
# two-stage exact scan: parity tests, then the int8-shadow bench line
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_two_stage.py -x -q 2>&1 | tail -4
python bench.py --tuning shadow8=1 --no-cpu-baseline --steps 300 2>/dev/null | tail -1 > gpurun_out/bench_shadow8.json
python -c "import json; d=json.load(open('gpurun_out/bench_shadow8.json')); print('SHADOW8', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])"
python bench.py --tuning shadow8=1 --no-cpu-baseline --steps 300 --k 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SHADOW8_K100', d['value'], d['ms_per_step'])"
