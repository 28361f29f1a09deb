#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_proximity.py tests/test_gpu_growth.py tests/test_gpu_search.py tests/test_gpu_hybrid_query.py tests/test_gpu_intersection_kats.py tests/test_gpu_boolean.py tests/test_gpu_bm25std_norm.py -x -q -m gpu > gpurun_out/r02d_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r02d_tests.txt
tail -40 gpurun_out/r02d_tests.txt
