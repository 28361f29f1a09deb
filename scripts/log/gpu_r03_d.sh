#!/bin/bash
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_coalesce.py tests/test_gpu_concurrency.py tests/test_gpu_reference_hybrid_reader.py -x -q --durations=5 > gpurun_out/r03d_tests.txt 2>&1
echo "tests rc=$? t=$(( $(date +%s) - T0 ))s" >> gpurun_out/r03d_tests.txt
tail -12 gpurun_out/r03d_tests.txt
for n in 2 8; do
  RSGPU_BENCH_OVERSUBSCRIBE=1 timeout 600 python3 bench.py --gpus $n --steps 20 --warmup 5 --rows 2000000 > gpurun_out/r03d_bench_g$n.json 2> gpurun_out/r03d_bench_g$n.err
  echo "bench --gpus $n rc=$? t=$(( $(date +%s) - T0 ))s"; tail -3 gpurun_out/r03d_bench_g$n.err
  python3 - <<PY
import json
d = json.load(open("gpurun_out/r03d_bench_g$n.json"))
print(d["value"], d["unit"], d["ms_per_step"])
print("collective", json.dumps(d.get("collective")))
print("callers", json.dumps(d["config"].get("concurrent_callers")))
PY
done
timeout 300 python scripts/diag/two_stage_lumps.py > gpurun_out/r03d_lumps.json 2> gpurun_out/r03d_lumps.err
echo "lumps rc=$? t=$(( $(date +%s) - T0 ))s"; cat gpurun_out/r03d_lumps.json | head -80
