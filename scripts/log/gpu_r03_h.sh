#!/bin/bash
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_iterators.py tests/test_gpu_reference_hybrid_reader.py tests/test_gpu_tree.py -x -q > gpurun_out/r03h_tests.txt 2>&1; echo "tests rc=$? t=$(( $(date +%s) - T0 ))"; tail -8 gpurun_out/r03h_tests.txt
timeout 600 python tests/bench_iterator.py > gpurun_out/r03h_iter.json 2> gpurun_out/r03h_iter.err; echo "bench rc=$? t=$(( $(date +%s) - T0 ))"; cat gpurun_out/r03h_iter.json; tail -3 gpurun_out/r03h_iter.err
