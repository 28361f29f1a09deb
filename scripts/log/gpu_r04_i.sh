#!/bin/bash
# round 4, call i: the in-process RCCL exchange from the caller's thread (child process, deadline) + the tests fixed after call h
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_shard_comm.py -q -p no:cacheprovider > gpurun_out/r04i_comm.txt 2>&1; echo "comm rc=$?"
tail -15 gpurun_out/r04i_comm.txt
timeout 400 python -m pytest tests/test_gpu_over_limit.py tests/test_gpu_coalesce.py tests/test_gpu_sharded.py -q -p no:cacheprovider --timeout 120 > gpurun_out/r04i_a.txt 2>&1; echo "a rc=$?"
tail -15 gpurun_out/r04i_a.txt
