#!/bin/bash
# round 4, final tree: the whole GPU suite, smoke as the driver calls it, the driver's bench command, the multi-shard records
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 300 > gpurun_out/r04_final_tests.txt 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r04_final_tests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_final_bench.out 2> gpurun_out/r04_final_bench.err; echo "bench rc=$?"
grep "^{" gpurun_out/r04_final_bench.out | tail -1 > gpurun_out/r04_bench_1gpu_final.json; cut -c1-330 gpurun_out/r04_bench_1gpu_final.json
for n in 2 8; do
RSGPU_BENCH_OVERSUBSCRIBE=1 timeout 500 python3 bench.py --gpus $n --steps 20 --warmup 5 --rows 2000000 2> gpurun_out/r04_final_g$n.err | grep "^{" | tail -1 > gpurun_out/r04_bench_${n}shards_one_device.json
python3 - <<PY
import json
d = json.load(open("gpurun_out/r04_bench_${n}shards_one_device.json"))
c = d["config"].get("concurrent_callers", {})
print("g$n value", round(d["value"]), "global qps", round(d["config"]["global_qps_on_sharded_corpus"]), "p50", round(d["config"]["p50_ms"], 3), "callers", {k: (round(v["qps"]) if isinstance(v, dict) else round(v, 2)) for k, v in c.items()})
PY
done
