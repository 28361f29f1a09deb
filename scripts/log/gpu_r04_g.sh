#!/bin/bash
# round 4, call g: new tests first (shard comm, over-limit, queued timeouts), then the whole GPU suite
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_shard_comm.py tests/test_gpu_over_limit.py tests/test_gpu_coalesce.py tests/test_gpu_coalesce_wide.py -q -p no:cacheprovider > gpurun_out/r04g_new.txt 2>&1; echo "new rc=$?"
tail -25 gpurun_out/r04g_new.txt
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r04g_all.txt 2>&1; echo "all rc=$?"
tail -8 gpurun_out/r04g_all.txt
