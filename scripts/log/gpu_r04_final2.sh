#!/bin/bash
# round 4, final tree: the whole GPU suite + smoke + the driver's bench command + rocprofv3 kernel stats of the general tile
# kernel's shapes (per-shape durations)
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 700 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 300 > gpurun_out/r04_final2_tests.txt 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r04_final2_tests.txt | cut -c1-300
timeout 200 python __graft_entry__.py --smoke > gpurun_out/r04_final2_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r04_final2_smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_final2_bench.json 2> gpurun_out/r04_final2_bench.err; echo "bench rc=$?"
cut -c1-260 gpurun_out/r04_final2_bench.json
timeout 300 bash scripts/gpu_r04_r.sh > gpurun_out/r04_final2_prof.txt 2>&1; echo "prof rc=$?"
tail -3 gpurun_out/r04_final2_prof.txt | cut -c1-1200
