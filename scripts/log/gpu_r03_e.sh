#!/bin/bash
# round 3: the batched MFMA pass -- (1) regression with the experiment knobs off, (2) producer waves: parity + A/B in one
# process, (3) BFLOAT16 through the int8 passes, (4) PMC of the SHIPPED query-stationary kernels (fp16 and int8)
set -u
export TMPDIR=/tmp
R=$(pwd); mkdir -p gpurun_out
T0=$(date +%s)
export PYTHONPATH=scripts/diag${PYTHONPATH:+:$PYTHONPATH}
timeout 500 python -m pytest tests/test_gpu_batch_i8_shadow.py tests/test_gpu_batch_qs.py tests/test_gpu_batch.py -x -q -p no:cacheprovider > gpurun_out/r03e_regress.txt 2>&1; echo "regression (knobs off): rc=$? t=$(( $(date +%s) - T0 ))"; tail -2 gpurun_out/r03e_regress.txt
RSGPU_TUNING=qs_pw=1 timeout 300 python -m pytest -p tuning_plugin tests/test_gpu_batch_i8_shadow.py -x -q -p no:cacheprovider > gpurun_out/r03e_pw_parity.txt 2>&1
rc=$?; echo "producer waves, parity: rc=$rc t=$(( $(date +%s) - T0 ))"; tail -3 gpurun_out/r03e_pw_parity.txt
if [ $rc -eq 0 ]; then
  KNOB=qs_pw VALUES=0,1 I8_SHADOW=1 ROUNDS=3 REPS=8 timeout 600 python scripts/bench_batch_knob.py > gpurun_out/r03e_pw_ab.json 2> gpurun_out/r03e_pw_ab.err
  echo "producer waves, A/B: rc=$? t=$(( $(date +%s) - T0 ))"; tail -c 600 gpurun_out/r03e_pw_ab.json
fi
timeout 600 python scripts/diag/bf16_i8_shadow_check.py > gpurun_out/r03e_bf16_check.txt 2>&1; echo "bf16 int8 shadow: rc=$? t=$(( $(date +%s) - T0 ))"; tail -10 gpurun_out/r03e_bf16_check.txt
# (4) counters of the shipped kernels: one pass per variant, all counters in one group
for v in fp16 i8; do
  [ $v = i8 ] && export I8_SHADOW=1
  (cd /tmp && REPS=4 timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d "$R/gpurun_out/r03e_pmc_$v" -o b -- python "$R/scripts/bench_batch.py" > "$R/gpurun_out/r03e_pmc_$v.log" 2>&1)
  echo "pmc $v rc=$? t=$(( $(date +%s) - T0 ))"
done
unset I8_SHADOW
python - <<'PY'
import csv, glob, collections, json
out = {}
for v in ("fp16", "i8"):
    dur = {}
    for f in glob.glob("gpurun_out/r03e_pmc_%s/*kernel_trace.csv" % v):
        for r in csv.DictReader(open(f)):
            if "gemm_qs_kernel" in r["Kernel_Name"]:
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"][:90])
    agg = collections.defaultdict(dict)
    for f in glob.glob("gpurun_out/r03e_pmc_%s/*counter_collection.csv" % v):
        for r in csv.DictReader(open(f)):
            if "gemm_qs_kernel" in r["Kernel_Name"]:
                agg[r["Dispatch_Id"]][r["Counter_Name"]] = agg[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    # the long launches only (the last filter phase: 3/4 of the corpus)
    if not dur:
        print(v, "no gemm_qs launches found"); continue
    mx = max(d for d, _ in dur.values())
    big = [i for i, (d, _) in dur.items() if d > 0.5 * mx and i in agg]
    rec = {"kernel": dur[big[0]][1] if big else None, "launches": len(big), "avg_ns": sum(dur[i][0] for i in big) / max(len(big), 1)}
    names = sorted({c for i in big for c in agg[i]})
    for c in names:
        rec[c] = sum(agg[i].get(c, 0.0) for i in big) / max(len(big), 1)
    if rec.get("GRBM_GUI_ACTIVE") and rec["avg_ns"]:
        rec["sclk_ghz_from_GRBM_GUI_ACTIVE_over_8_xcd"] = rec["GRBM_GUI_ACTIVE"] / 8.0 / rec["avg_ns"]
    if rec.get("SQ_BUSY_CYCLES") and rec.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        rec["mfma_busy_over_sq_busy"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / rec["SQ_BUSY_CYCLES"]
    out[v] = rec
    print(v, json.dumps(rec, indent=1))
json.dump(out, open("gpurun_out/r03e_qs_pmc.json", "w"), indent=1)
PY
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete; find gpurun_out -name "*.db" -delete; find gpurun_out -name "*counter_collection.csv" -size +2M -delete
