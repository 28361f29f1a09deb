#!/bin/bash
# quick round-end validation on one MI355X: GPU parity suite, smoke, default bench (no profiler passes)
set -u
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_final.json
python -c "import json; d=json.load(open('gpurun_out/bench_final.json')); print('BENCH', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('two_stage_exact_scan_extra', {}).get('qps'), d['cpu_baseline']['value'])"
