#!/bin/bash
# round 3, call p: the staged probe with the tile kernel's improvements -- parity of everything that intersects, A/B of the tile sizes
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2
timeout 900 python -m pytest tests/test_gpu_hybrid_query.py tests/test_gpu_hybrid_tiles.py tests/test_gpu_search.py tests/test_gpu_intersection_kats.py tests/test_gpu_boolean.py tests/test_gpu_tree.py \
  tests/test_gpu_iterators.py tests/test_gpu_proximity.py tests/test_gpu_docid64.py tests/test_gpu_fusion.py -q -p no:cacheprovider --maxfail=10 > gpurun_out/r03p_tests.txt 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/r03p_tests.txt
timeout 600 python scripts/bench_hybrid_ab.py hybrid_tiles=0 probe_dpt=1,4 > gpurun_out/r03p_probe_ab.txt 2> gpurun_out/r03p_probe_ab.err; echo "ab rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r03p_probe_ab.txt"):
    try: r = json.loads(l)
    except Exception: continue
    print(r["rep"], r["knobs"], "warm p50 %.4f stages %s | cold p50 %.4f same %s %s" % (
        r["warm"]["p50_ms"], {k: round(v, 4) for k, v in r["warm"]["stage_device_ms"].items()}, r["cold"]["p50_ms"], r["warm"]["same_answers"], r["cold"]["same_answers"]))
PY
