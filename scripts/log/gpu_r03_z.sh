#!/bin/bash
# round 3, call z: configs[4] as it is (5 M vectors) and with a vector for EVERY document (50 M x 768 fp32 = 154 GB of rows: all
# 250 k hits are KNN candidates, 768 MB of rows per query) -- the KNN branch of the tile kernel when it is bound by bytes
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2
timeout 600 python -m pytest tests/test_gpu_hybrid_tiles.py -q -p no:cacheprovider -x > gpurun_out/r03z_tests.txt 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r03z_tests.txt
for nv in 5000000 50000000; do
N_VEC=$nv timeout 900 python scripts/bench_hybrid_ab.py hybrid_tiles=0,1 > gpurun_out/r03z_hybrid_$nv.txt 2> gpurun_out/r03z_hybrid_$nv.err; echo "rc=$?"
python - <<PY
import json
for l in open("gpurun_out/r03z_hybrid_$nv.txt"):
    try: r = json.loads(l)
    except Exception: continue
    print($nv, r["rep"], r["knobs"], "warm p50 %.4f min %.4f stages %s same %s" % (
        r["warm"]["p50_ms"], r["warm"]["min_ms"], {k: round(v, 4) for k, v in r["warm"]["stage_device_ms"].items()}, r["warm"]["same_answers"]))
PY
done
