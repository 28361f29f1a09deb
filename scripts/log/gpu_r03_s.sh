#!/bin/bash
# round 3, call s: the sync points of wavefronts that do not fit the staging buffer (fix) + the types of the multi-query scan
# (tests), then the multi-query scans with the next tile's rows requested ahead (mq16 = 3 / 4, mq_pf = 1 / 2), A/B in one process
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2
timeout 900 python -m pytest tests/test_gpu_coalesce_types.py tests/test_gpu_decode_qint.py tests/test_gpu_hybrid_query.py tests/test_gpu_search.py \
  -q -p no:cacheprovider --maxfail=12 > gpurun_out/r03s_tests.txt 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/r03s_tests.txt
timeout 600 python scripts/bench_mq.py --variants "mq16=1,mq_pf=0;mq16=3,mq_pf=1;mq16=4,mq_pf=2" > gpurun_out/r03s_mq_pf.json 2> gpurun_out/r03s_mq_pf.err; echo "mq rc=$?"
tail -3 gpurun_out/r03s_mq_pf.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03s_mq_pf.json"))
print("single", d["single"])
for name, r in d["mq"].items():
    print(name, r.get("kernel"))
    for nq in ("4", "8", "12", "16"):
        print("   nq", nq, [(x["scan_ms"], x["qps_wall"], x["same"]) for x in r[nq]])
PY
timeout 300 python scripts/bench_mq.py --metric l2 --variants "mq16=1,mq_pf=0;mq16=3,mq_pf=1;mq16=4,mq_pf=2" > gpurun_out/r03s_mq_pf_l2.json 2>> gpurun_out/r03s_mq_pf.err; echo "mq l2 rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03s_mq_pf_l2.json"))
print("L2 single", d["single"])
for name, r in d["mq"].items():
    print(name, r.get("kernel"))
    for nq in ("4", "8", "12", "16"):
        print("   nq", nq, [(x["scan_ms"], x["qps_wall"], x["same"]) for x in r[nq]])
PY
