#!/bin/bash
# round 3, call l: rocprof kernel durations of the cold decode
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_decode_fifo.py tests/test_gpu_search.py tests/test_gpu_proximity.py tests/test_gpu_docid64.py tests/test_gpu_index_mutations.py tests/test_gpu_hybrid_query.py -x -q -p no:cacheprovider > gpurun_out/r03l_tests.txt 2>&1; echo "tests rc=$?"
tail -12 gpurun_out/r03l_tests.txt
timeout 300 python tests/make_decode_lists.py /tmp/lists.npz > /dev/null 2>&1; echo "lists rc=$?"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r03l_prof" -o dec -- python "$R/scripts/bench_decode.py" /tmp/lists.npz > "$R/gpurun_out/r03l_prof.log" 2>&1); echo "prof rc=$?"
find gpurun_out/r03l_prof -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | cut -c1-100,180-330
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03l_prof/**/*kernel_trace.csv", recursive=True)
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if "decode" in r["Kernel_Name"]]
    for r in rows:
        print(r["Kernel_Name"][:70], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X"), r.get("Workgroup_Size_X"))
PY
find gpurun_out/r03l_prof -name "*kernel_trace.csv" -delete
