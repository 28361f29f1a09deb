#!/bin/bash
# round 3, call q2: the two-shard bench three times (do eight callers settle into full passes?) + the coalescer tests
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_coalesce.py tests/test_gpu_sharded.py tests/test_gpu_concurrency.py -q -p no:cacheprovider --maxfail=10 > gpurun_out/r03q_tests.txt 2>&1; echo "tests rc=$?"
tail -2 gpurun_out/r03q_tests.txt
for rep in 1 2 3; do
RSGPU_BENCH_OVERSUBSCRIBE=1 timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 --rows 2000000 > gpurun_out/r03q_bench_g2_$rep.json 2> gpurun_out/r03q_bench_g2.err
python3 - <<PY
import json
d = json.load(open("gpurun_out/r03q_bench_g2_$rep.json"))
c = d["config"].get("concurrent_callers")
print("g2", round(d["value"]), "1 caller", round(c["1_threads"]["qps"]), "8 callers", round(c["8_threads"]["qps"]), "per pass", round(c["8_threads"]["queries_per_shard_pass"], 2), "p50", c["8_threads"]["p50_ms"], "x", round(c["x_one_caller"], 2))
PY
done
RSGPU_BENCH_OVERSUBSCRIBE=1 timeout 600 python3 bench.py --gpus 8 --steps 20 --warmup 5 --rows 2000000 > gpurun_out/r03q_bench_g8.json 2> gpurun_out/r03q_bench_g8.err
python3 - <<PY
import json
d = json.load(open("gpurun_out/r03q_bench_g8.json"))
c = d["config"].get("concurrent_callers")
print("g8", round(d["value"]), "1 caller", round(c["1_threads"]["qps"]), "8 callers", round(c["8_threads"]["qps"]), "per pass", round(c["8_threads"]["queries_per_shard_pass"], 2), "x", round(c["x_one_caller"], 2), json.dumps(d.get("collective"))[:200])
PY
