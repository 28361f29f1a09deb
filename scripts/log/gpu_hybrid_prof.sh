#!/bin/bash
# per-kernel times and launch gaps of the fused hybrid query (configs[4])
export TMPDIR=/tmp
R=$(pwd); mkdir -p gpurun_out
timeout 600 python tests/hybrid_fused_prof.py 2>&1 | tail -1 | tee gpurun_out/hybrid_fused_wall.txt
(cd /tmp && REPS=60 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/hybrid_prof" -o h -- python "$R/tests/hybrid_fused_prof.py" > "$R/gpurun_out/hybrid_prof.log" 2>&1)
tail -1 gpurun_out/hybrid_prof.log
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/hybrid_prof/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last full query: walk back from the end to the last intersect_probe
names = [r["Kernel_Name"] for r in rows]
last = max(i for i, n in enumerate(names) if "intersect_probe" in n)
prev = max(i for i, n in enumerate(names[:last]) if "intersect_probe" in n)
seg = rows[prev:last]
t0 = int(seg[0]["Start_Timestamp"])
with open("gpurun_out/hybrid_timeline.txt", "w") as out:
    for r in seg:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        line = "%8.1f us  +%6.1f us  q%s  %s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), __import__("re").sub(r"\(anonymous namespace\)::|rsgpu::|void ", "", r["Kernel_Name"]).split("(")[0][:60])
        print(line); out.write(line + "\n")
    line = "span of one query on the device: %.1f us (next query's first kernel starts at %.1f us)" % ((max(int(r["End_Timestamp"]) for r in seg) - t0) / 1e3, (int(rows[last]["Start_Timestamp"]) - t0) / 1e3)
    print(line); out.write(line + "\n")
PY
cp "$(find gpurun_out/hybrid_prof -name '*kernel_stats.csv' | head -1)" gpurun_out/hybrid_kernel_stats.csv
find gpurun_out/hybrid_prof -name "*kernel_trace.csv" -size +2M -delete
