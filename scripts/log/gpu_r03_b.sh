#!/bin/bash
mkdir -p gpurun_out
T0=$(date +%s)
timeout 1200 python -m pytest tests/test_gpu_coalesce.py -x -q --durations=5 > gpurun_out/r03b_coalesce.txt 2>&1
echo "coalesce rc=$? t=$(( $(date +%s) - T0 ))s" >> gpurun_out/r03b_coalesce.txt
tail -12 gpurun_out/r03b_coalesce.txt
timeout 600 python scripts/bench_mq.py --ab mq_ring=0,1 > gpurun_out/r03b_mq_ab.json 2> gpurun_out/r03b_mq_ab.err
echo "mq rc=$? t=$(( $(date +%s) - T0 ))s"; tail -3 gpurun_out/r03b_mq_ab.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03b_mq_ab.json"))
print(d["single"])
for name, r in d["mq"].items():
    print(name, r.get("kernel"))
    for nq in ("2", "3", "4", "5", "8"):
        print("  nq", nq, [(x["scan_ms"], x["wall_ms"], x["gbs"], x["same"], x["redo"]) for x in r[nq]])
PY
