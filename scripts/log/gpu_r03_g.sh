#!/bin/bash
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_tree.py -x -q > gpurun_out/r03g_tree.txt 2>&1; echo "tree rc=$? t=$(( $(date +%s) - T0 ))"; tail -25 gpurun_out/r03g_tree.txt
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_boolean.py tests/test_gpu_proximity.py tests/test_gpu_iterators.py tests/test_gpu_intersection_kats.py tests/test_gpu_hybrid_query.py tests/test_gpu_bm25std_norm.py tests/test_gpu_docid64.py -x -q > gpurun_out/r03g_search.txt 2>&1; echo "search rc=$? t=$(( $(date +%s) - T0 ))"; tail -5 gpurun_out/r03g_search.txt
