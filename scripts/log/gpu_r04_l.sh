#!/bin/bash
# round 4, call l: new tests (KNN branch vs oracle, full-size fp32-native batch), bench under torch.distributed.run with the C
# exchange (one rank), concurrent callers with the early wide switch
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 700 python -m pytest tests/test_gpu_hybrid_tiles.py tests/test_gpu_coalesce.py tests/test_gpu_coalesce_wide.py "tests/test_gpu_fullsize.py::test_fp32_native_matrix_core_batch_at_the_baseline_size" "tests/test_gpu_fullsize.py::test_fp32_native_l2_batch_at_the_baseline_size" -q -p no:cacheprovider --timeout 400 > gpurun_out/r04l_tests.txt 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/r04l_tests.txt
RSGPU_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04l_ranks.json 2> gpurun_out/r04l_ranks.err; echo "ranks rc=$?"
grep "^{" gpurun_out/r04l_ranks.json | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['parallelism'], json.dumps(d.get('collective')))"
tail -3 gpurun_out/r04l_ranks.err
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hybrid-extra --no-batched-extra > gpurun_out/r04l_callers.json 2> gpurun_out/r04l_callers.err; echo "callers rc=$?"
grep "^{" gpurun_out/r04l_callers.json | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); cc=d['config']['concurrent_callers']
for t in (8,16,32,64):
    r=cc['%d_threads'%t]; print(t, {k:r.get(k) for k in ('qps','p50_ms','queries_per_pass','wide_passes','wide_pass_device_ms','multi_query_passes','bit_identical_to_serial')})"
