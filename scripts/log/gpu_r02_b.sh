#!/bin/bash
# round 2, second GPU pass: new tests, full-size oracle parity (timed), the rewritten bench (N=1 with extras, in-process
# 2-shard run on one device, 1-rank RCCL run)
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_index_mutations.py tests/test_gpu_philox.py tests/test_gpu_sharded.py tests/test_gpu_growth.py -x -q -m gpu > gpurun_out/r02b_tests.txt 2>&1
echo "tests rc=$? t=$(( $(date +%s) - T0 ))s" >> gpurun_out/r02b_tests.txt
tail -15 gpurun_out/r02b_tests.txt
T1=$(date +%s)
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu --durations=10 > gpurun_out/r02b_fullsize.txt 2>&1
echo "fullsize rc=$? t=$(( $(date +%s) - T1 ))s" >> gpurun_out/r02b_fullsize.txt
tail -25 gpurun_out/r02b_fullsize.txt
T2=$(date +%s)
timeout 900 python bench.py > gpurun_out/r02b_bench_1gpu.json 2> gpurun_out/r02b_bench_1gpu.err
echo "bench rc=$? t=$(( $(date +%s) - T2 ))s"
tail -5 gpurun_out/r02b_bench_1gpu.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02b_bench_1gpu.json').read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ('value','unit','ms_per_step')}), json.dumps(d['roofline']), json.dumps(d.get('cpu_baseline')))
print(json.dumps(d['config'], indent=1)[:6000])
PY
RSGPU_BENCH_OVERSUBSCRIBE=1 timeout 300 python bench.py --gpus 2 --rows 2000000 --steps 100 --warmup 10 > gpurun_out/r02b_bench_inproc2.json 2> gpurun_out/r02b_bench_inproc2.err; echo "inproc2 rc=$?"; tail -2 gpurun_out/r02b_bench_inproc2.err; cat gpurun_out/r02b_bench_inproc2.json | cut -c1-1500
timeout 300 python bench.py --gpus 1 --rows 2000000 --steps 100 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/r02b_bench_1x2M.json 2>/dev/null; cut -c1-300 gpurun_out/r02b_bench_1x2M.json
RSGPU_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --rows 2000000 --steps 100 --warmup 10 > gpurun_out/r02b_bench_dist1.json 2> gpurun_out/r02b_bench_dist1.err; echo "dist1 rc=$?"; tail -2 gpurun_out/r02b_bench_dist1.err; cut -c1-600 gpurun_out/r02b_bench_dist1.json
