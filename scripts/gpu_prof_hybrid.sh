#!/bin/bash
# PMC passes over the two-launch hybrid query on the configs[4] distinct-query stream (scripts/bench_hybrid_stream.py): which pipe of a
# CU the tile kernel keeps busy -- VALU / LDS / VMEM instruction counts and active cycles against the kernel's cycles.
#   TAG=r06 bash scripts/gpu_prof_hybrid.sh   -> gpurun_out/${TAG}_hybrid_tile_pipes_pmc${SUFFIX}.json
# Counters in their own runs, --kernel-trace only next to --pmc (MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -u
TAG=${TAG:-r06}; SUFFIX=${SUFFIX:-}; KPAT=${KPAT:-hybrid_tile_kernel}; export TAG SUFFIX KPAT
export TMPDIR=/tmp CYCLES=${CYCLES:-2}
R=$(pwd); mkdir -p gpurun_out
run() {
  local name=$1; shift
  (cd /tmp && OUT=${TAG}_hyb_pmc_stream_$name.json timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_prof_hybp_$name" -o b -- python "$R/scripts/bench_hybrid_stream.py" > "$R/gpurun_out/${TAG}_prof_hybp_$name.log" 2>&1)
}
run clk GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU
run lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM
run act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
run wv SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64
python - <<'PY'
import csv, glob, json, os
TAG, SUFFIX, KPAT = os.environ["TAG"], os.environ["SUFFIX"], os.environ["KPAT"]
out = {"command": "scripts/gpu_prof_hybrid.sh (rocprofv3 --pmc <group> --kernel-trace -- python scripts/bench_hybrid_stream.py; CYCLES=%s)" % os.environ["CYCLES"],
       "kernel_pattern": KPAT}
for name in ("clk", "lds", "act", "wv"):
    vals, dur, kern = {}, [], None
    for f in glob.glob("gpurun_out/%s_prof_hybp_%s/*counter_collection.csv" % (TAG, name)):
        for r in csv.DictReader(open(f)):
            if KPAT in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                kern = r["Kernel_Name"]
    for f in glob.glob("gpurun_out/%s_prof_hybp_%s/*kernel_trace.csv" % (TAG, name)):
        for r in csv.DictReader(open(f)):
            if KPAT in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if kern:
        out["kernel_name_in_trace"] = kern
    for k, v in vals.items():
        out[k + "_avg"] = sum(v) / len(v)
    if dur:
        out["us_avg_under_pmc_" + name] = sum(dur) / len(dur)
        out["launches_" + name] = len(dur)
g = out.get("GRBM_GUI_ACTIVE_avg")
if g:
    cyc = g / 8.0                                    # (summed over the eight XCDs)
    out["kernel_cycles"] = cyc
    out["sclk_ghz"] = cyc / (out["us_avg_under_pmc_clk"] * 1e3)
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INSTS_SMEM"):
        if k + "_avg" in out:
            out[k + "_per_cu_cycle"] = out[k + "_avg"] / 256.0 / cyc
    for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"):
        if k + "_avg" in out and out.get("SQ_BUSY_CYCLES_avg"):
            out[k + "_over_SQ_BUSY_CYCLES"] = out[k + "_avg"] / out["SQ_BUSY_CYCLES_avg"]
    if out.get("SQ_WAVE_CYCLES_avg"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k + "_avg" in out:
                out[k + "_over_WAVE_CYCLES"] = out[k + "_avg"] / out["SQ_WAVE_CYCLES_avg"]
json.dump(out, open("gpurun_out/%s_hybrid_tile_pipes_pmc%s.json" % (TAG, SUFFIX), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
for n in clk lds act wv; do tail -3 gpurun_out/${TAG}_prof_hybp_$n.log | cut -c1-300; done
