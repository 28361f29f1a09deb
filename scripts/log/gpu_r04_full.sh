#!/bin/bash
# round 4: the whole GPU suite + smoke + the driver's bench command + the round's profiles, in one call
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 300 > gpurun_out/r04_full_tests.txt 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/r04_full_tests.txt
timeout 200 python __graft_entry__.py --smoke > gpurun_out/r04_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r04_smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_1gpu_driver_args.json 2> gpurun_out/r04_bench.err; echo "bench rc=$?"
tail -2 gpurun_out/r04_bench.err
cut -c1-400 gpurun_out/r04_bench_1gpu_driver_args.json
timeout 1500 bash scripts/gpu_prof_r04.sh > gpurun_out/r04_prof.txt 2>&1; echo "prof rc=$?"
tail -30 gpurun_out/r04_prof.txt | cut -c1-900
