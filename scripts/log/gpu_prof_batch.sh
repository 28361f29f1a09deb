#!/bin/bash
# rocprofv3 passes over the batched-query bench: kernel stats, then PMC groups (one group per run).
set -u
export TMPDIR=/tmp REPS=4
R=$(pwd); mkdir -p gpurun_out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_batch_stats" -o b -- python "$R/scripts/bench_batch.py" > "$R/gpurun_out/prof_batch_stats.log" 2>&1)
find gpurun_out/prof_batch_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "cut -c1-160 {} | head -12"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$R/gpurun_out/prof_batch_pmc$i" -o b -- python "$R/scripts/bench_batch.py" > "$R/gpurun_out/prof_batch_pmc$i.log" 2>&1)
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/prof_batch_pmc*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "gemm_qs" if "gemm_qs" in k else ("gemm" if "gemm_topk" in k else ("batch_select" if "batch_select" in k else None))
        if name: agg[name][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r.get("Grid_Size", 0) or 0)))
    for name, d in agg.items():
        for c, v in d.items():
            big = [x for x, g in v]
            print(f.split("/")[1], name, c, "n=%d max=%.4g avg=%.4g" % (len(big), max(big), sum(big) / len(big)))
PY
find gpurun_out -name "*kernel_trace.csv" -size +4M -delete; find gpurun_out -name "*.db" -delete
