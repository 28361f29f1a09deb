#!/bin/bash
# round 3, call o: grid cap of the sixteen-query pass
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
timeout 600 python scripts/bench_mq.py --ab mq_blocks_per_cu=3,6,9,12,8 > gpurun_out/r03o_bench_mq.json 2> gpurun_out/r03o_bench_mq.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03o_bench_mq.json"))
for name, r in d["mq"].items():
    for nq, v in r.items():
        if nq != "kernel":
            print(name, nq, [(x["scan_ms"], x["qps_wall"], x["same"]) for x in v])
PY
