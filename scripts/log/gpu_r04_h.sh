#!/bin/bash
# round 4, call h: the in-process RCCL exchange (child process, deadline), then the test files the aborted run never reached
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 240 python -m pytest tests/test_gpu_shard_comm.py -q -p no:cacheprovider -x > gpurun_out/r04h_comm.txt 2>&1; echo "comm rc=$?"
tail -12 gpurun_out/r04h_comm.txt
timeout 600 python -m pytest tests/test_gpu_over_limit.py tests/test_gpu_coalesce.py tests/test_gpu_coalesce_wide.py tests/test_gpu_sharded.py -q -p no:cacheprovider --timeout 120 > gpurun_out/r04h_a.txt 2>&1; echo "a rc=$?"
tail -12 gpurun_out/r04h_a.txt
timeout 600 python -m pytest tests/test_gpu_topk_source_kats.py tests/test_gpu_tree.py tests/test_gpu_two_stage.py tests/test_gpu_types.py tests/test_gpu_select_paths.py tests/test_gpu_search.py tests/test_gpu_reference_hybrid_reader.py -q -p no:cacheprovider --timeout 120 > gpurun_out/r04h_b.txt 2>&1; echo "b rc=$?"
tail -6 gpurun_out/r04h_b.txt
