#!/bin/bash
# round 3, call q: coalesced two-stage passes over the int8 shadow -- parity, then the callers bench with the shadow leg
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_coalesce.py tests/test_gpu_two_stage.py tests/test_gpu_fullsize.py -x -q -p no:cacheprovider -k "not oracle_full" > gpurun_out/r03q_tests.txt 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/r03q_tests.txt
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batched-extra --no-hybrid-extra > gpurun_out/r03q_callers.json 2> gpurun_out/r03q_callers.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03q_callers.json").read().strip().splitlines()[-1])
c = d["config"]["concurrent_callers"]
for k in ("8_threads", "16_threads"):
    r = c[k]; print(k, {x: (round(r[x], 3) if isinstance(r[x], float) else r[x]) for x in r if x in ("qps", "x_single_stream", "queries_per_pass", "p50_ms", "multi_query_scan_ms", "bit_identical_to_serial")})
print(json.dumps(c.get("with_int8_shadow_two_stage"), indent=0)[:2500])
print(json.dumps(d["config"].get("two_stage_exact_scan_extra"))[:600])
PY
