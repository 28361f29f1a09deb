#!/bin/bash
# round 3, last call: the whole GPU suite + smoke + the driver's bench command on the final tree; the in-process multi-shard form
# of the bench (two shards on the one device) as a smoke of the driver's --gpus N command
R="$GRAFT_REPO_ROOT"; cd "$R"
bash scripts/gpu_r03_full.sh
RSGPU_BENCH_OVERSUBSCRIBE=1 timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 --rows 2000000 > gpurun_out/r03_bench_g2_oversubscribed.json 2> gpurun_out/r03_bench_g2_oversubscribed.err
echo "bench --gpus 2 rc=$?"; tail -2 gpurun_out/r03_bench_g2_oversubscribed.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench_g2_oversubscribed.json"))
print(d["value"], d["unit"], d["ms_per_step"], d["n_gpus"])
print("collective", json.dumps(d.get("collective")))
print("callers", json.dumps(d["config"].get("concurrent_callers"))[:600])
PY
