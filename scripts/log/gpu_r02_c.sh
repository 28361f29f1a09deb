#!/bin/bash
# round 2, third GPU pass: reference hybrid_reader on the GPU lib, fused hybrid query + windowed intersect, mutations
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_reference_hybrid_reader.py tests/test_gpu_hybrid_query.py tests/test_gpu_index_mutations.py tests/test_gpu_search.py tests/test_gpu_intersection_kats.py tests/test_gpu_boolean.py tests/test_gpu_growth.py -x -q -m gpu > gpurun_out/r02c_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r02c_tests.txt
tail -40 gpurun_out/r02c_tests.txt
timeout 600 python bench.py --steps 100 --no-cpu-baseline --no-two-stage-extra --no-batched-extra > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02c_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'])
print(json.dumps(d['config'].get('hybrid'), indent=1))
PY
tail -3 gpurun_out/r02c_bench.err
