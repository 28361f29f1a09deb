#!/bin/bash
# round 4, call f: the whole GPU suite on the current tree, the pre-flight, the driver's bench command
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/r04f_tests.txt 2>&1; echo "tests rc=$?"
tail -12 gpurun_out/r04f_tests.txt
timeout 300 python bench.py --preflight > gpurun_out/r04f_preflight.json 2> gpurun_out/r04f_preflight.err; echo "preflight rc=$?"; cat gpurun_out/r04f_preflight.json; tail -3 gpurun_out/r04f_preflight.err
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04f_bench.json 2> gpurun_out/r04f_bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r04f_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04f_bench.json"))
c=d["config"]
print("value",d["value"],"frac",d["roofline"]["frac"], "p50", c["p50_ms"])
print("collective_rccl", json.dumps(c.get("collective_rccl")))
print("batched_mfma_f32", {k:c.get("batched_mfma_f32",{}).get(k) for k in ("device_ms_per_pass","qps_device","hbm_frac","bit_identical_to_single_queries","error")})
cc=c.get("concurrent_callers",{})
for t in (16,32,64):
    r=cc.get("%d_threads"%t,{})
    print(t, {k:r.get(k) for k in ("qps","p50_ms","queries_per_pass","wide_passes","wide_pass_device_ms","bit_identical_to_serial")})
h=c.get("hybrid",{})
print("hybrid", {k:h.get(k) for k in ("wall_ms_per_query","wall_ms_p95","path","full_codec_answers_equal_freqs_only","bench_wall_s","error")})
for name in ("stream_freqs_only","stream_full_codec"):
    for m in ("warm","cold"):
        r=h.get(name,{}).get(m,{})
        print(name,m,{k:r.get(k) for k in ("wall_ms_p50","wall_ms_p95","device_ms","tile_kernel_hbm_frac","decode_gbs_of_encoded_bytes")})
print("parity", json.dumps(h.get("parity"))[:600])
print("cpu", json.dumps(d.get("cpu_baseline"))[:300])
PY
