#!/bin/bash
# rocprofv3 kernel stats of the driver's bench command WITHOUT the concurrent-callers extra: that extra's control leg runs
# eight uncoalesced scans at once (the same scan_kernel instantiation at an eighth of the HBM each), which would average
# into the headline kernel's figure.  The bench line of the same process is kept next to the stats.
set -u
export TMPDIR=/tmp
R=$(pwd); mkdir -p gpurun_out
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r03_prof_stats2" -o b -- python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-callers-extra > "$R/gpurun_out/r03_prof_stats2.log" 2>&1)
grep "^{\"metric\"" gpurun_out/r03_prof_stats2.log | tail -1 > gpurun_out/r03_bench_under_rocprof_no_callers.json
f=$(find gpurun_out/r03_prof_stats2 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r03_bench_kernel_stats.csv; cut -c1-150 "$f" | head -6
python3 -c "
import json; d=json.load(open('gpurun_out/r03_bench_under_rocprof_no_callers.json')); print(d['roofline'])"
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete; find gpurun_out -name "*.db" -delete
