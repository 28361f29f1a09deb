#!/bin/bash
# round 3, call t: the hybrid query in two launches (hybrid_kernels.hip) -- parity against the staged pipeline, A/B in one process,
# kernel timeline of one query
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2
timeout 900 python -m pytest tests/test_gpu_hybrid_tiles.py tests/test_gpu_hybrid_query.py tests/test_gpu_search.py \
  -q -p no:cacheprovider --maxfail=12 > gpurun_out/r03t_tests.txt 2>&1; echo "tests rc=$?"
tail -25 gpurun_out/r03t_tests.txt
timeout 600 python scripts/bench_hybrid_ab.py hybrid_tiles=0,1 > gpurun_out/r03t_hybrid_ab.txt 2> gpurun_out/r03t_hybrid_ab.err; echo "ab rc=$?"
tail -3 gpurun_out/r03t_hybrid_ab.err
python - <<'PY'
import json
for l in open("gpurun_out/r03t_hybrid_ab.txt"):
    try: r = json.loads(l)
    except Exception: continue
    print(r["rep"], r["knobs"], "warm p50 %.4f min %.4f stages %s | cold p50 %.4f stages %s same %s %s" % (
        r["warm"]["p50_ms"], r["warm"]["min_ms"], {k: round(v, 4) for k, v in r["warm"]["stage_device_ms"].items()}, r["cold"]["p50_ms"],
        {k: round(v, 4) for k, v in r["cold"]["stage_device_ms"].items()}, r["warm"]["same_answers"], r["cold"]["same_answers"]))
PY
(cd /tmp && REPS=40 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/hybrid_prof" -o h -- python "$R/tests/hybrid_fused_prof.py" > "$R/gpurun_out/hybrid_prof.log" 2>&1)
tail -1 gpurun_out/hybrid_prof.log
python - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/hybrid_prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
first = "hybrid_tile" if any("hybrid_tile" in n for n in names) else "intersect_probe"
last = max(i for i, n in enumerate(names) if first in n)
prev = max(i for i, n in enumerate(names[:last]) if first in n)
seg = rows[prev:last]
t0 = int(seg[0]["Start_Timestamp"])
with open("gpurun_out/r03t_hybrid_timeline.txt", "w") as out:
    for r in seg:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        line = "%8.1f us  +%6.1f us  %s" % (s / 1e3, (e - s) / 1e3, re.sub(r"\(anonymous namespace\)::|rsgpu::|void ", "", r["Kernel_Name"]).split("(")[0][:70])
        print(line); out.write(line + "\n")
    line = "span of one query on the device: %.1f us (next query's first kernel starts at %.1f us)" % ((max(int(r["End_Timestamp"]) for r in seg) - t0) / 1e3, (int(rows[last]["Start_Timestamp"]) - t0) / 1e3)
    print(line); out.write(line + "\n")
PY
find gpurun_out/hybrid_prof -name "*kernel_trace.csv" -size +2M -delete
