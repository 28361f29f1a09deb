#!/bin/bash
# fp32 batched queries over the fp16 shadow: parity tests, then the 10M x 768 bench
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_batch_f32_shadow.py -x -q 2>&1 | tail -15
F32_SHADOW=1 REPS=8 QUERIES_PER_CALL=1280 timeout 600 python scripts/bench_batch.py 2>&1 | tail -3
