#!/bin/bash
# round 3, call q: the coalescers' expectation = callers answered + callers queued meanwhile (two half-size groups alternated on
# the sharded handle): tests, the two-shard bench, the single-index bench with its concurrent-callers extra
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2
timeout 900 python -m pytest tests/test_gpu_coalesce.py tests/test_gpu_coalesce_types.py tests/test_gpu_sharded.py tests/test_gpu_concurrency.py -q -p no:cacheprovider --maxfail=10 > gpurun_out/r03q_tests.txt 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r03q_tests.txt
for rep in 1 2; do
RSGPU_BENCH_OVERSUBSCRIBE=1 timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 --rows 2000000 > gpurun_out/r03q_bench_g2_$rep.json 2> gpurun_out/r03q_bench_g2.err
python3 - <<PY
import json
d = json.load(open("gpurun_out/r03q_bench_g2_$rep.json"))
print("g2", d["value"], json.dumps(d["config"].get("concurrent_callers"))[:500])
PY
done
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03q_bench.json 2> gpurun_out/r03q_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03q_bench.json"))
cc = d["config"]["concurrent_callers"]
print("value", d["value"])
for t in (1, 2, 4, 8, 16):
    r = cc["%d_threads" % t]
    print(t, "qps %.0f x %.2f per pass %.2f p50 %.2f same %s" % (r["qps"], r["x_single_stream"], r["queries_per_pass"], r["p50_ms"], r["bit_identical_to_serial"]))
sh = cc.get("with_int8_shadow_two_stage", {})
print({k: (v.get("qps"), v.get("queries_per_pass")) for k, v in sh.items() if isinstance(v, dict)})
PY
