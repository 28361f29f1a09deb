#!/bin/bash
# the kernel timeline of ONE batched pass (cfg 3 fp16 IP, then the fp32 cosine pass): rocprofv3 --kernel-trace, last pass listed
# with start offsets, durations and gaps -> gpurun_out/${TAG}_batch_timeline_{f16,f32}.txt
set -u
TAG=${TAG:-r06}; export TMPDIR=/tmp
R=$(pwd); mkdir -p gpurun_out
(cd /tmp && REPS=3 QUERIES_PER_CALL=512 timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_tl_f16" -o b -- python "$R/scripts/bench_batch.py" > "$R/gpurun_out/${TAG}_tl_f16.log" 2>&1)
(cd /tmp && REPS=3 METRICS=cosine SHAPES=2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_tl_f32" -o b -- python "$R/scripts/bench_batch_f32.py" > "$R/gpurun_out/${TAG}_tl_f32.log" 2>&1)
python - <<'PY'
import csv, glob, os, re
TAG = os.environ.get("TAG", "r06")
for kind in ("f16", "f32"):
    rows = []
    for f in glob.glob("gpurun_out/%s_tl_%s/*kernel_trace.csv" % (TAG, kind)):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the last long gemm_qs launch ends a pass's filter; list from the select that closes the previous pass to the end of this one
    longs = [i for i, r in enumerate(rows) if "gemm_qs" in r[2] and r[1] - r[0] > 1_000_000]
    if len(longs) < 3:
        print(kind, "no passes found", len(rows)); continue
    lo = longs[-3] + 1
    hi = longs[-2]
    # extend to the kernels after the long launch up to the next pass's first gemm launch
    j = hi + 1
    while j < len(rows) and "gemm_qs" not in rows[j][2]:
        j += 1
    # start at the first gemm_qs launch after the previous long one
    i = lo
    while i < hi and "gemm_qs" not in rows[i][2]:
        i += 1
    # and include the small kernels before it that belong to this pass (after the previous pass's tail): everything since lo
    out = []
    t0 = rows[lo][0]
    prev_end = None
    for s, e, n in rows[lo:j]:
        short = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("rsgpu::", "").replace("void ", ""))[:60]
        out.append("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev_end is None else (s - prev_end) / 1e3, short))
        prev_end = e if prev_end is None else max(prev_end, e)
    total = (rows[j - 1][1] - rows[lo][0]) / 1e3
    out.append("span %.1f us over %d kernels; sum of durations %.1f us" % (total, j - lo, sum(e - s for s, e, _ in rows[lo:j]) / 1e3))
    open("gpurun_out/%s_batch_timeline_%s.txt" % (TAG, kind), "w").write("\n".join(out) + "\n")
    print(kind); print("\n".join(out))
PY
