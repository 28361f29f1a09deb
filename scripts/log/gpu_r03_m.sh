#!/bin/bash
# round 3, call m: qint decode with sync points -- parity, then kernel durations (first decode = one lane per block + sync
# points written; later decodes = eight lanes per block), A/B against decode_sync=0 and the window parser
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_decode_qint.py tests/test_gpu_search.py tests/test_gpu_proximity.py tests/test_gpu_docid64.py tests/test_gpu_index_mutations.py tests/test_gpu_hybrid_query.py tests/test_gpu_boolean.py tests/test_gpu_tree.py -x -q -p no:cacheprovider > gpurun_out/r03m_tests.txt 2>&1; echo "tests rc=$?"
tail -12 gpurun_out/r03m_tests.txt
timeout 300 python tests/make_decode_lists.py /tmp/lists.npz > /dev/null 2>&1; echo "lists rc=$?"
for F in; do
set -- $F
(cd /tmp && DECODE_FIFO=$1 DECODE_SYNC=$2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r03m_prof" -o dec -- python "$R/scripts/bench_decode.py" /tmp/lists.npz > "$R/gpurun_out/r03m_prof_$1$2.log" 2>&1)
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/r03m_prof/**/*kernel_trace.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "decode_blocks_kernel" in r["Kernel_Name"]]
print("fifo=$1 sync=$2", [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Grid_Size_X"]) for r in rows])
PY
rm -rf gpurun_out/r03m_prof
done
# the hybrid query, warm and cold, on the final decode path
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-callers-extra > gpurun_out/r03m_bench.json 2> gpurun_out/r03m_bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("gpurun_out/r03m_bench.json").read().strip().splitlines()[-1])
def find(o, key):
    if isinstance(o, dict):
        for k, v in o.items():
            if k == key: return v
            r = find(v, key)
            if r is not None: return r
    return None
h = find(d, "hybrid") or find(d, "hybrid_config5") or {}
print(json.dumps({k: h.get(k) for k in ("wall_ms_per_query", "wall_ms_p95", "wall_ms_min", "wall_ms_per_query_branches_enqueued_after_the_count", "cold", "stage_device_ms", "parity")})[:1800])
PY
