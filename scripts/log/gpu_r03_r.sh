#!/bin/bash
# round 3, call r: multi-query scan for INT8 / UINT8 / FLOAT64 / multi-value indexes, wide probe tiles, paired decode --
# parity, then the hybrid query A/B'd in one process
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2
timeout 1200 python -m pytest tests/test_gpu_coalesce_types.py tests/test_gpu_hybrid_query.py tests/test_gpu_decode_qint.py tests/test_gpu_search.py \
  tests/test_gpu_coalesce.py tests/test_gpu_boolean.py tests/test_gpu_tree.py tests/test_gpu_iterators.py tests/test_gpu_proximity.py \
  tests/test_gpu_intersection_kats.py tests/test_gpu_flat.py tests/test_gpu_index_mutations.py tests/test_gpu_sharded.py tests/test_gpu_batch.py \
  -q -p no:cacheprovider --maxfail=12 > gpurun_out/r03r_tests.txt 2>&1; echo "tests rc=$?"
tail -40 gpurun_out/r03r_tests.txt
timeout 600 python scripts/bench_hybrid_ab.py probe_dpt=1,4 decode_pair=0,1 > gpurun_out/r03r_hybrid_ab.txt 2> gpurun_out/r03r_hybrid_ab.err; echo "ab rc=$?"
tail -3 gpurun_out/r03r_hybrid_ab.err
python - <<'PY'
import json
for l in open("gpurun_out/r03r_hybrid_ab.txt"):
    try: r = json.loads(l)
    except Exception: continue
    print(r["rep"], r["knobs"], "warm p50 %.4f min %.4f int %.4f | cold p50 %.4f int %.4f same %s %s" % (
        r["warm"]["p50_ms"], r["warm"]["min_ms"], r["warm"]["stage_device_ms"]["intersect_ms"], r["cold"]["p50_ms"],
        r["cold"]["stage_device_ms"]["intersect_ms"], r["warm"]["same_answers"], r["cold"]["same_answers"]))
PY
