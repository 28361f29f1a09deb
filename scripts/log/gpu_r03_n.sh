#!/bin/bash
# round 3, call n: the sixteen-query pass -- parity of the coalescer suite, device time per pass by number of queries,
# A/B against two passes of eight (mq16=0) and against U = 1
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_coalesce.py tests/test_gpu_sharded.py tests/test_gpu_batch.py -x -q -p no:cacheprovider > gpurun_out/r03n_tests.txt 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/r03n_tests.txt
timeout 600 python scripts/bench_mq.py --ab mq16=1,2,0 > gpurun_out/r03n_bench_mq.json 2> gpurun_out/r03n_bench_mq.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03n_bench_mq.json"))
print("single", d["single"])
for name, r in d["mq"].items():
    for nq, v in r.items():
        if nq != "kernel":
            print(name, nq, [(x["scan_ms"], x["qps_wall"], x["same"]) for x in v])
    print(name, r.get("kernel"))
PY
