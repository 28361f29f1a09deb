#!/bin/bash
# round 4, call b: wide coalescer + fp32-native tests, the whole bench line (new sub-records), kernel stats of the f32 batch
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2
timeout 900 python -m pytest tests/test_gpu_batch_f32_native.py tests/test_gpu_coalesce_wide.py tests/test_gpu_coalesce.py -x -q -p no:cacheprovider > gpurun_out/r04b_tests.txt 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/r04b_tests.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04b_bench.json 2> gpurun_out/r04b_bench.err; echo "bench rc=$?"
tail -5 gpurun_out/r04b_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04b_bench.json"))
c=d["config"]
print("value",d["value"],"frac",d["roofline"]["frac"])
print("batched_mfma_f32",json.dumps(c.get("batched_mfma_f32"))[:900])
cc=c.get("concurrent_callers",{})
for t in (8,16,32,64):
    print(t, json.dumps(cc.get("%d_threads"%t))[:600])
h=c.get("hybrid",{})
print("hybrid", {k:h.get(k) for k in ("wall_ms_per_query","wall_ms_p95","path","full_codec_answers_equal_freqs_only","input_generation_s","bench_wall_s","error")})
print("fo warm", json.dumps(h.get("stream_freqs_only",{}).get("warm")))
print("fo cold", json.dumps(h.get("stream_freqs_only",{}).get("cold")))
print("full warm", json.dumps(h.get("stream_full_codec",{}).get("warm")))
print("full cold", json.dumps(h.get("stream_full_codec",{}).get("cold")))
r=h.get("repeat_same_query",{})
print("repeat", {k:r.get(k) for k in ("wall_ms_per_query","wall_ms_p95","stage_device_ms")}, r.get("cold",{}).get("wall_ms_per_query"))
print("parity", json.dumps(h.get("parity")))
PY
(cd /tmp && METRICS=cosine SHAPES=2 REPS=4 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r04b_prof" -o f32 -- python "$R/scripts/bench_batch_f32.py" > "$R/gpurun_out/r04b_prof.log" 2>&1); echo "prof rc=$?"
find gpurun_out/r04b_prof -name "*kernel_stats.csv" | head -1 | xargs -r head -14 | cut -c1-180
find gpurun_out/r04b_prof -name "*kernel_trace.csv" -delete; find gpurun_out/r04b_prof -name "*.db" -delete
