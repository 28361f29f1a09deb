#!/bin/bash
# round 4, call p: the whole GPU suite + smoke + the driver's bench command on the tree with the general hybrid tile kernel
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 300 > gpurun_out/r04p_tests.txt 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/r04p_tests.txt | cut -c1-300
timeout 200 python __graft_entry__.py --smoke > gpurun_out/r04p_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r04p_smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04p_bench.json 2> gpurun_out/r04p_bench.err; echo "bench rc=$?"
tail -2 gpurun_out/r04p_bench.err
cut -c1-300 gpurun_out/r04p_bench.json
