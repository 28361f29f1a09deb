#!/bin/bash
# round-end validation on one MI355X: GPU parity suite, smoke, default bench, kernel stats of bench / hybrid
set -u
export TMPDIR=/tmp
R=$(pwd); mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_final.json
python -c "import json; d=json.load(open('gpurun_out/bench_final.json')); print('BENCH', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('two_stage_exact_scan_extra', {}).get('qps'), d['cpu_baseline']['value'])"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bench_stats" -o b -- python "$R/bench.py" --no-two-stage-extra > "$R/gpurun_out/prof_bench_stats.log" 2>&1)
find gpurun_out/prof_bench_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "cut -c1-150 {} | head -6"
(cd /tmp && N_DOCS=50000000 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_hybrid_stats" -o h -- python "$R/tests/bench_hybrid.py" > "$R/gpurun_out/prof_hybrid_stats.log" 2>&1)
grep -c HYBRID_OK gpurun_out/prof_hybrid_stats.log
find gpurun_out -name "*kernel_trace.csv" -size +4M -delete; find gpurun_out -name "*.db" -delete
