#!/bin/bash
# round 3, first GPU call: the coalescer + multi-query scan (parity), the regression suites next to what changed, then the
# driver's exact bench command twice (two-stage fallback counters, concurrent callers)
mkdir -p gpurun_out
(hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i 'unique') > gpurun_out/r03a_box.txt 2>&1
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_coalesce.py -x -q --durations=5 > gpurun_out/r03a_coalesce.txt 2>&1
echo "coalesce rc=$? t=$(( $(date +%s) - T0 ))s" >> gpurun_out/r03a_coalesce.txt
tail -15 gpurun_out/r03a_coalesce.txt
timeout 900 python -m pytest tests/test_gpu_flat.py tests/test_gpu_two_stage.py tests/test_gpu_batch_i8_shadow.py tests/test_gpu_batch_f32_shadow.py tests/test_gpu_growth.py tests/test_gpu_concurrency.py tests/test_gpu_sharded.py tests/test_gpu_batch.py -x -q > gpurun_out/r03a_regress.txt 2>&1
echo "regress rc=$? t=$(( $(date +%s) - T0 ))s" >> gpurun_out/r03a_regress.txt
tail -8 gpurun_out/r03a_regress.txt
for i in 1 2; do
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-batched-extra --no-hybrid-extra > gpurun_out/r03a_bench_$i.json 2> gpurun_out/r03a_bench_$i.err
  echo "bench $i rc=$? t=$(( $(date +%s) - T0 ))s"
  python3 - <<PY
import json
try:
    d = json.load(open("gpurun_out/r03a_bench_$i.json"))
    c = d["config"]
    print("value", d["value"], "frac", d["roofline"]["frac"])
    print("two_stage", json.dumps(c.get("two_stage_exact_scan_extra"))[:900])
    cc = c.get("concurrent_callers", {})
    print("callers", json.dumps({k: cc.get(k) for k in ("qps", "x_single_stream", "p50_ms", "bit_identical_to_serial", "kernel", "error", "eight_threads_without_coalescer")}))
    for t in (1, 2, 4, 8, 16):
        print(t, json.dumps(cc.get("%d_threads" % t)))
except Exception as e:
    print("parse failed", e)
    print(open("gpurun_out/r03a_bench_$i.err").read()[-2000:])
PY
done
