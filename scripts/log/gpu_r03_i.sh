#!/bin/bash
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_hybrid_query.py tests/test_gpu_docid64.py tests/test_gpu_search.py tests/test_gpu_bm25std_norm.py -x -q > gpurun_out/r03i_tests.txt 2>&1; echo "tests rc=$? t=$(( $(date +%s) - T0 ))"; tail -12 gpurun_out/r03i_tests.txt
for v in 1 0; do
  RSGPU_TUNING=hybrid_one_pass=$v timeout 300 python tests/hybrid_fused_prof.py 2>&1 | tail -1
done
