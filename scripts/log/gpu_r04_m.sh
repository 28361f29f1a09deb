#!/bin/bash
# round 4, call m: the in-process multi-shard bench on ONE device (records with the corrected p50), fp16 batch shape A/B,
# concurrent callers with the wide pass from five callers
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
for n in 2 8; do
RSGPU_BENCH_OVERSUBSCRIBE=1 timeout 500 python3 bench.py --gpus $n --steps 20 --warmup 5 --rows 2000000 > gpurun_out/r04m_bench_g$n.out 2> gpurun_out/r04m_bench_g$n.err; echo "g$n rc=$?"
grep "^{" gpurun_out/r04m_bench_g$n.out | tail -1 > gpurun_out/r04_bench_${n}shards_one_device.json
python3 - <<PY
import json
d = json.load(open("gpurun_out/r04_bench_${n}shards_one_device.json"))
c = d["config"].get("concurrent_callers", {})
print("g$n value", round(d["value"]), "global qps", round(d["config"]["global_qps_on_sharded_corpus"]), "p50", d["config"]["p50_ms"], "ms/step", d["ms_per_step"], "callers", {k: (round(v["qps"]) if isinstance(v, dict) else v) for k, v in c.items()}, "collective", json.dumps(d.get("collective"))[:160], "rccl", json.dumps(d["config"].get("collective_rccl"))[:200], "preflight", d["config"].get("preflight"))
PY
tail -2 gpurun_out/r04m_bench_g$n.err
done
for qs in 1 2; do GEMM_QS=$qs REPS=8 timeout 300 python scripts/bench_batch.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp16 gemm_qs=$qs', d['device_ms_per_batch'], d['qps_device'])"; done
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hybrid-extra --no-batched-extra --no-two-stage-extra > gpurun_out/r04m_callers.json 2> gpurun_out/r04m_callers.err; echo "callers rc=$?"
grep "^{" gpurun_out/r04m_callers.json | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); cc=d['config']['concurrent_callers']
for t in (2,4,8,16):
    r=cc['%d_threads'%t]; print(t, {k:r.get(k) for k in ('qps','p50_ms','queries_per_pass','wide_passes','wide_pass_device_ms','multi_query_passes','bit_identical_to_serial')})"
