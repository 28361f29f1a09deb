#!/bin/bash
# PMC passes over the batched matrix-core pass of BASELINE configs[2] (10M x 768 fp16 IP top-100, 256 queries per pass):
#   TAG=r06 KPAT=gemm_qs_kernel bash scripts/gpu_prof_batch.sh       [extra env for scripts/bench_batch.py: I8_SHADOW=1, TUNING=...]
# -> gpurun_out/${TAG}_batch_qs_pmc${SUFFIX}.json: per LONG launch (the last filter phase, rows [n/4, n)) duration, sclk from
# GRBM_GUI_ACTIVE, matrix-pipe busy fraction, wait / issue split, LDS activity, FETCH/WRITE traffic against the algorithmic bytes.
# Counters in their own runs, --kernel-trace only next to --pmc (MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -u
TAG=${TAG:-r06}; KPAT=${KPAT:-gemm_qs_kernel}; SUFFIX=${SUFFIX:-}; export TAG KPAT SUFFIX
export TMPDIR=/tmp REPS=${REPS:-3} QUERIES_PER_CALL=${QUERIES_PER_CALL:-512}
R=$(pwd); mkdir -p gpurun_out
run() {  # name, counters...
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_prof_bqs_$name" -o b -- python "$R/scripts/bench_batch.py" > "$R/gpurun_out/${TAG}_prof_bqs_$name.log" 2>&1)
}
run clk GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU
run lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SALU
run fetch FETCH_SIZE
run write WRITE_SIZE
python - <<'PY'
import csv, glob, json, os
TAG, KPAT, SUFFIX = os.environ["TAG"], os.environ["KPAT"], os.environ["SUFFIX"]
BPR = int(os.environ.get("BYTES_PER_ROW", 1536))
def counters(name):
    vals, kern = {}, None
    for f in glob.glob("gpurun_out/%s_prof_bqs_%s/*counter_collection.csv" % (TAG, name)):
        for r in csv.DictReader(open(f)):
            if KPAT in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                kern = r["Kernel_Name"]
    return vals, kern
def durations(name):
    d = []
    for f in glob.glob("gpurun_out/%s_prof_bqs_%s/*kernel_trace.csv" % (TAG, name)):
        for r in csv.DictReader(open(f)):
            if KPAT in r["Kernel_Name"]:
                d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return d
def big(v):
    b = [x for x in v if x > 0.5 * max(v)] if v else []
    return (sum(b) / len(b)) if b else None
clk, kern = counters("clk"); lds, _ = counters("lds"); fe, _ = counters("fetch"); wr, _ = counters("write")
du = durations("clk")
long_us = [x for x in du if x > 0.5 * max(du)] if du else []
out = {"command": "scripts/gpu_prof_batch.sh (rocprofv3 --pmc <group> --kernel-trace -- python scripts/bench_batch.py; REPS=%s)" % os.environ["REPS"],
       "kernel_name_in_trace": kern, "env": {k: os.environ[k] for k in ("I8_SHADOW", "TUNING", "GEMM_QS", "ROWS") if k in os.environ}}
if long_us:
    t_us = sum(long_us) / len(long_us)
    rows_long = int(os.environ.get("ROWS", 10_000_000)) * 3 // 4
    g, m = big(clk.get("GRBM_GUI_ACTIVE", [])), big(clk.get("SQ_VALU_MFMA_BUSY_CYCLES", []))
    out.update({"long_launches": len(long_us), "long_launch_us_avg": t_us, "rows_long_launch": rows_long,
                "algorithmic_bytes_long_launch": rows_long * BPR, "hbm_gbs_long_launch": rows_long * BPR / t_us / 1e3,
                "hbm_frac_long_launch": rows_long * BPR / t_us / 1e3 / 8000.0,
                "sclk_ghz": (g / 8 / t_us / 1e3) if g else None,
                "mfma_pipe_busy_frac": (m / 1024.0 / (g / 8)) if (g and m) else None,
                "mfma_tflops_long_launch": 2.0 * 256 * 768 * rows_long / t_us / 1e6})
    for k, v in list(clk.items()) + list(lds.items()):
        out[k + "_avg_long"] = big(v)
    w = big(clk.get("SQ_WAVE_CYCLES", []))
    if w:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if big(clk.get(k, [])) is not None:
                out[k + "_over_WAVE_CYCLES"] = big(clk[k]) / w
    f, wv = big(fe.get("FETCH_SIZE", [])), big(wr.get("WRITE_SIZE", []))
    if f:
        out.update({"FETCH_SIZE_raw_KB_avg_long": f, "WRITE_SIZE_raw_KB_avg_long": wv,
                    "traffic_over_algorithmic": (f * 2 + (wv or 0)) * 1024 / (rows_long * BPR)})
json.dump(out, open("gpurun_out/%s_batch_qs_pmc%s.json" % (TAG, SUFFIX), "w"), indent=1)
print(json.dumps(out))
PY
find gpurun_out -name "*kernel_trace.csv" -size +1M -delete; find gpurun_out -name "*.db" -delete; find gpurun_out -name "*counter_collection.csv" -size +1M -delete
