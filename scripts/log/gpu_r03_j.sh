#!/bin/bash
# round 3, call j: the L2 form of the batched matrix-core pass -- parity, then its timing next to the IP pass
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_batch_l2.py -x -q -p no:cacheprovider > gpurun_out/r03j_l2.txt 2>&1; echo "l2 rc=$?"
tail -15 gpurun_out/r03j_l2.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r03j_prof" -o l2 -- python "$R/scripts/bench_batch_l2.py" > "$R/gpurun_out/r03j_prof.log" 2>&1); echo "prof rc=$?"
find gpurun_out/r03j_prof -name "*kernel_stats.csv" | head -1 | xargs -r head -25 | cut -c1-200
find gpurun_out/r03j_prof -name "*kernel_trace.csv" -delete; find gpurun_out/r03j_prof -name "*.db" -delete
timeout 600 python scripts/bench_batch_l2.py > gpurun_out/r03j_bench_l2.json 2> gpurun_out/r03j_bench_l2.err; echo "bench rc=$?"
cat gpurun_out/r03j_bench_l2.json; tail -5 gpurun_out/r03j_bench_l2.err
