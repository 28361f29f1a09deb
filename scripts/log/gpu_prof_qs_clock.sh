#!/bin/bash
# engine clock under the QS kernel variants: GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) / kernel duration
set -u
export TMPDIR=/tmp REPS=4
R=$(pwd); mkdir -p gpurun_out
for qs in 1 3 4; do
  (cd /tmp && GEMM_QS=$qs timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$R/gpurun_out/prof_qsclk$qs" -o b -- python "$R/scripts/bench_batch.py" > "$R/gpurun_out/prof_qsclk$qs.log" 2>&1)
done
python - <<'PY'
import csv, glob, collections
for qs in (1, 3, 4):
    dur = {}
    for f in glob.glob("gpurun_out/prof_qsclk%d/*kernel_trace.csv" % qs):
        for r in csv.DictReader(open(f)):
            if "gemm_qs" in r["Kernel_Name"]:
                dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/prof_qsclk%d/*counter_collection.csv" % qs):
        for r in csv.DictReader(open(f)):
            if "gemm_qs" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
    d = sum(dur.values()) / max(len(dur), 1)
    print("variant", qs, "avg kernel ns", d)
    for c, v in agg.items():
        avg = sum(x for _, x in v) / len(v)
        print("   ", c, "%.4g" % avg, "per-ns %.3f" % (avg / d if d else 0))
PY
find gpurun_out -name "*kernel_trace.csv" -size +4M -delete; find gpurun_out -name "*.db" -delete
