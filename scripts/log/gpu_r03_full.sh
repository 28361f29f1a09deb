#!/bin/bash
# the whole GPU suite + smoke + the driver's bench command, as the driver runs them at round end
mkdir -p gpurun_out
(hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i 'unique') > gpurun_out/r03_box_id.txt 2>&1
T0=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=8 -p no:cacheprovider > gpurun_out/r03_full_tests.txt 2>&1
echo "tests rc=$? t=$(( $(date +%s) - T0 ))s" >> gpurun_out/r03_full_tests.txt
tail -22 gpurun_out/r03_full_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_driver_args.json 2> gpurun_out/r03_bench_driver_args.err
echo "bench rc=$? t=$(( $(date +%s) - T0 ))s"; tail -3 gpurun_out/r03_bench_driver_args.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench_driver_args.json"))
c = d["config"]
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), d["roofline"].get("traffic_note"))
print("verify", c.get("verify"))
ts = c.get("two_stage_exact_scan_extra", {})
print("two_stage", {k: ts.get(k) for k in ("qps", "fallbacks", "p50_ms", "p95_ms", "max_ms", "python_gc_passes_inside_the_timed_loop", "error")})
cc = c.get("concurrent_callers", {})
print("callers", {k: cc.get(k) for k in ("qps", "x_single_stream", "p50_ms", "bit_identical_to_serial", "kernel", "error")})
b = c.get("batched_mfma", {})
print("batched", {k: b.get(k) for k in ("device_ms_per_pass", "qps_device", "hbm_frac", "mfma_frac", "parity", "error")})
print("  i8", {k: (b.get("int8_shadow_extra") or {}).get(k) for k in ("device_ms_per_pass", "qps_device", "bit_identical_to_single_queries", "error")})
h = c.get("hybrid", {})
print("hybrid", {k: h.get(k) for k in ("path", "wall_ms_per_query", "wall_ms_p95", "wall_ms_min", "stage_device_ms", "parity", "error")})
print("  staged", h.get("staged_pipeline_same_process"))
print("  cold", h.get("cold"))
print("cpu", {k: d.get("cpu_baseline", {}).get(k) for k in ("value", "cores", "p50_ms", "eight_threads", "gpu_answers_checked_against_oracle_on_full_corpus")})
PY
