#!/bin/bash
# PMC pass over the batched bench (valid kernels only): clocks, MFMA busy, waits, LDS
set -u
export TMPDIR=/tmp REPS=4
R=$(pwd); mkdir -p gpurun_out
i=0
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$R/gpurun_out/prof_qspmc$i" -o b -- python "$R/scripts/bench_batch.py" > "$R/gpurun_out/prof_qspmc$i.log" 2>&1)
done
python - <<'PY'
import csv, glob, collections
for i in (1, 2):
    dur = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/prof_qspmc%d/*kernel_trace.csv" % i):
        for r in csv.DictReader(open(f)):
            if "gemm_qs" in r["Kernel_Name"]:
                dur["qs"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    d = sum(dur["qs"]) / max(len(dur["qs"]), 1)
    print("pass", i, "gemm_qs avg ns", d, "n", len(dur["qs"]))
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/prof_qspmc%d/*counter_collection.csv" % i):
        for r in csv.DictReader(open(f)):
            if "gemm_qs" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(agg.items()):
        print("   %-28s %.4g" % (c, sum(v) / len(v)))
PY
find gpurun_out -name "*kernel_trace.csv" -size +4M -delete; find gpurun_out -name "*.db" -delete
