#!/bin/bash
# BASELINE configs[3] on ONE device: the per-GPU workload (10 M x 768 fp32 L2 top-10, single-query stream) and dry runs of both multi-GPU
# forms (8 shards on the one device, host merge; the rank-per-GPU form at world 1 with the RCCL exchange).  NOT a scaling curve.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python bench.py --metric l2 --no-extras --no-cpu-baseline > gpurun_out/r06_bench_1gpu_l2.json 2> gpurun_out/r06_l2.err; cut -c1-300 gpurun_out/r06_bench_1gpu_l2.json
RSGPU_BENCH_OVERSUBSCRIBE=1 timeout 600 python bench.py --gpus 8 --rows 1250000 --metric l2 --no-cpu-baseline --no-extras > gpurun_out/r06_bench_8shards_one_device_l2.json 2> gpurun_out/r06_8s.err; cut -c1-300 gpurun_out/r06_bench_8shards_one_device_l2.json
RSGPU_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --metric l2 --no-extras --no-cpu-baseline --rows 2000000 > gpurun_out/r06_bench_ranks_world1_l2.json 2> gpurun_out/r06_w1.err; cut -c1-300 gpurun_out/r06_bench_ranks_world1_l2.json
tail -2 gpurun_out/r06_8s.err gpurun_out/r06_w1.err | cut -c1-300
