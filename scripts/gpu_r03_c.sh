#!/bin/bash
mkdir -p gpurun_out
T0=$(date +%s)
timeout 1200 python -m pytest tests/test_gpu_coalesce.py -x -q --durations=5 > gpurun_out/r03c_coalesce.txt 2>&1
echo "coalesce rc=$? t=$(( $(date +%s) - T0 ))s" >> gpurun_out/r03c_coalesce.txt
tail -8 gpurun_out/r03c_coalesce.txt
for v in "mq_ring=1 --ab mq_blocks_per_cu=2,4,8" "mq_ring=0 --ab mq_blocks_per_cu=2,4,8,16"; do
  timeout 600 python scripts/bench_mq.py --reps 10 --tuning $v > gpurun_out/r03c_mq.json 2> gpurun_out/r03c_mq.err
  echo "== $v rc=$? t=$(( $(date +%s) - T0 ))s"; tail -2 gpurun_out/r03c_mq.err
  python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03c_mq.json"))
print(d["single"]["kernel_ms"])
for name, r in d["mq"].items():
    print(name, r.get("kernel"))
    for nq in ("2", "4", "5", "8"):
        print("  nq", nq, [(x["scan_ms"], x["wall_ms"], x["same"]) for x in r[nq]])
PY
done
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-batched-extra --no-hybrid-extra > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err
echo "bench rc=$? t=$(( $(date +%s) - T0 ))s"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03c_bench.json"))
c = d["config"]
print("value", d["value"], "frac", d["roofline"]["frac"])
ts = c.get("two_stage_exact_scan_extra", {})
print("two_stage", {k: ts.get(k) for k in ("qps", "fallbacks", "p50_ms", "p95_ms", "max_ms", "error")})
cc = c.get("concurrent_callers", {})
print("callers", json.dumps({k: cc.get(k) for k in ("qps", "x_single_stream", "p50_ms", "bit_identical_to_serial", "kernel", "error", "eight_threads_without_coalescer")}))
for t in (1, 2, 4, 8, 16):
    print(t, json.dumps(cc.get("%d_threads" % t)))
PY
