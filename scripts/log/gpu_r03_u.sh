#!/bin/bash
# round 3, call u: the two-launch hybrid query after the second pass over its kernels -- parity, where the tile kernel's time goes
# (one branch at a time; the phase clock of every tile), A/B against the staged pipeline
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2
timeout 900 python -m pytest tests/test_gpu_hybrid_tiles.py tests/test_gpu_hybrid_query.py -q -p no:cacheprovider --maxfail=12 > gpurun_out/r03u_tests.txt 2>&1; echo "tests rc=$?"
tail -12 gpurun_out/r03u_tests.txt
timeout 900 python scripts/bench_hybrid_parts.py > gpurun_out/r03u_parts.txt 2> gpurun_out/r03u_parts.err; echo "rc=$?"
tail -3 gpurun_out/r03u_parts.err
python - <<'PY'
import json
for l in open("gpurun_out/r03u_parts.txt"):
    r = json.loads(l)
    if "rep" in r:
        if r["rep"] == 1: print(r["form"], "wall %.4f tile %.4f reduce %.4f" % (r["wall_p50_ms"], r["tile_ms"], r["reduce_ms"]))
    else:
        print(r["form"], "span", r["kernel_span_us"], "start", r["tile_start_us"], "dur", r["tile_duration_us"])
        print("   mean", {k: round(v, 2) for k, v in r["phase_mean_us"].items()})
        print("   p95 ", {k: round(v, 2) for k, v in r["phase_p95_us"].items()})
PY
timeout 600 python scripts/bench_hybrid_ab.py hybrid_tiles=0,1 > gpurun_out/r03u_hybrid_ab.txt 2> gpurun_out/r03u_hybrid_ab.err; echo "ab rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r03u_hybrid_ab.txt"):
    try: r = json.loads(l)
    except Exception: continue
    print(r["rep"], r["knobs"], "warm p50 %.4f min %.4f stages %s | cold p50 %.4f stages %s same %s %s" % (
        r["warm"]["p50_ms"], r["warm"]["min_ms"], {k: round(v, 4) for k, v in r["warm"]["stage_device_ms"].items()}, r["cold"]["p50_ms"],
        {k: round(v, 4) for k, v in r["cold"]["stage_device_ms"].items()}, r["warm"]["same_answers"], r["cold"]["same_answers"]))
PY
