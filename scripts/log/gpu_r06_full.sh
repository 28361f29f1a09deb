#!/bin/bash
# round 6: the GPU suite, the bench with the driver's arguments, rocprofv3 stats + PMC passes (scripts/gpu_prof.sh)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r06_gpu_suite.txt; cat gpurun_out/r06_gpu_suite.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_1gpu_driver_args.json 2> gpurun_out/r06_bench_driver_args.err; cp gpurun_out/bench_full.json gpurun_out/r06_bench_1gpu_driver_args_full.json; cut -c1-600 gpurun_out/r06_bench_1gpu_driver_args.json
TAG=r06 timeout 1500 bash scripts/gpu_prof.sh 2>&1 | tail -30
