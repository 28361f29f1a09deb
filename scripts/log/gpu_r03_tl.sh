#!/bin/bash
# round 3: kernel timeline of one two-launch hybrid query (rocprofv3 --kernel-trace), device microseconds
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
(cd /tmp && REPS=40 timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/hybrid_prof" -o h -- python "$R/tests/hybrid_fused_prof.py" > "$R/gpurun_out/hybrid_prof.log" 2>&1)
grep HYBRID_FUSED gpurun_out/hybrid_prof.log | tail -1
python - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/hybrid_prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "hybrid_tile" in n]
with open("gpurun_out/r03_hybrid_two_launch_timeline.txt", "w") as out:
    out.write("RSGPU_HybridQuery in two launches, BASELINE configs[4], three consecutive queries (rocprofv3 --kernel-trace; start, +duration, device us)\n")
    for a, b in zip(idx[-4:-1], idx[-3:]):
        seg = rows[a:b]
        t0 = int(seg[0]["Start_Timestamp"])
        for r in seg:
            s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
            line = "%8.1f us  +%6.1f us  %s" % (s / 1e3, (e - s) / 1e3, re.sub(r"\(anonymous namespace\)::|rsgpu::|void ", "", r["Kernel_Name"]).split("(")[0][:70])
            print(line); out.write(line + "\n")
        line = "   next query's tile kernel starts at %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3)
        print(line); out.write(line + "\n")
PY
find gpurun_out/hybrid_prof -name "*kernel_trace.csv" -size +1M -delete
