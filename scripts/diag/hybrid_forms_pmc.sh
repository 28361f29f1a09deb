#!/bin/bash
export TMPDIR=/tmp CYCLES=2
R=$(pwd); mkdir -p gpurun_out
for form in both score knn; do
  (cd /tmp && RSGPU_BENCH_HYBRID_FORM=$form OUT=r06_hyb_form_$form.json timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d "$R/gpurun_out/r06_prof_form_$form" -o b -- python "$R/scripts/bench_hybrid_stream.py" > "$R/gpurun_out/r06_prof_form_$form.log" 2>&1)
  python - <<PY
import csv, glob
vals, dur = {}, []
for f in glob.glob("gpurun_out/r06_prof_form_$form/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "hybrid_tile_kernel" in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/r06_prof_form_$form/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "hybrid_tile_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("$form", "us %.1f" % (sum(dur) / max(len(dur), 1)), {k: round(sum(v) / len(v)) for k, v in vals.items()})
PY
done
