#!/bin/bash
# round 2, first GPU pass: new tests (mutations, philox, sharded, growth) + A/B of the mapped row matrix vs hipMalloc
mkdir -p gpurun_out
python -c "import torch; print(torch.cuda.device_count(), torch.cuda.get_device_name(0))" > gpurun_out/r02a_env.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_index_mutations.py tests/test_gpu_philox.py tests/test_gpu_sharded.py tests/test_gpu_growth.py -x -q -m gpu > gpurun_out/r02a_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r02a_tests.txt
tail -30 gpurun_out/r02a_tests.txt
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-two-stage-extra > gpurun_out/r02a_bench_vmm1.json 2> gpurun_out/r02a_bench_vmm1.err
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-two-stage-extra --tuning vmm=0 > gpurun_out/r02a_bench_vmm0.json 2> gpurun_out/r02a_bench_vmm0.err
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-two-stage-extra > gpurun_out/r02a_bench_vmm1b.json 2>> gpurun_out/r02a_bench_vmm1.err
for f in gpurun_out/r02a_bench_vmm*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['achieved'], d['roofline']['avg_kernel_ms'])"; done
tail -3 gpurun_out/r02a_bench_vmm1.err
