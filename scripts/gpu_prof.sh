#!/bin/bash
# profiles of a round (TAG=r05 bash scripts/gpu_prof.sh): rocprofv3 kernel stats of the DRIVER'S bench command; the HBM-traffic PMC passes over the headline scan
# (-> profiles/${TAG}_scan_pmc_hbm_traffic.json, stamped with the kernel's source hash: bench.py refuses a stale one), over the
# two-launch hybrid query on the distinct-query stream and over the fp32-native matrix-core pass (+ its clock / MFMA-busy
# counters).  Counters in their own runs, --kernel-trace only next to --pmc.
set -u
TAG=${TAG:-r05}; export TAG
export TMPDIR=/tmp
R=$(pwd); mkdir -p gpurun_out
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof_stats" -o b -- python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-callers-extra > "$R/gpurun_out/${TAG}_prof_stats.log" 2>&1)
grep "^{\"metric\"" gpurun_out/${TAG}_prof_stats.log | tail -1 > gpurun_out/${TAG}_bench_under_rocprof.json
f=$(find gpurun_out/${TAG}_prof_stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${TAG}_bench_kernel_stats.csv; cut -c1-170 "$f" | head -14
for grp in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 500 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_prof_$grp" -o b -- python "$R/bench.py" --steps 60 --warmup 5 --no-cpu-baseline --no-extras > "$R/gpurun_out/${TAG}_prof_$grp.log" 2>&1)
  (cd /tmp && OUT=${TAG}_hybrid_stream_ab_freqs_only.json CYCLES=2 timeout 500 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_prof_hyb_$grp" -o b -- python "$R/scripts/bench_hybrid_stream.py" > "$R/gpurun_out/${TAG}_prof_hyb_$grp.log" 2>&1)
  (cd /tmp && METRICS=cosine SHAPES=2 REPS=3 timeout 500 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_prof_f32_$grp" -o b -- python "$R/scripts/bench_batch_f32.py" > "$R/gpurun_out/${TAG}_prof_f32_$grp.log" 2>&1)
done
(cd /tmp && METRICS=cosine SHAPES=2 REPS=3 timeout 500 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_prof_f32_clk" -o b -- python "$R/scripts/bench_batch_f32.py" > "$R/gpurun_out/${TAG}_prof_f32_clk.log" 2>&1)
python - <<'PY'
import csv, glob, json, os, sys
TAG = os.environ['TAG']
sys.path.insert(0, ".")
import bench
def counters(dirpat, kernel_pat, names, exclude=None):
    """{counter: [values of the kernel's launches]} and the kernel's name, durations from the kernel trace"""
    vals, kern = {n: [] for n in names}, None
    for f in glob.glob("gpurun_out/%s/*counter_collection.csv" % dirpat):
        for r in csv.DictReader(open(f)):
            if kernel_pat in r["Kernel_Name"] and not (exclude and exclude in r["Kernel_Name"]) and r["Counter_Name"] in vals:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
                kern = r["Kernel_Name"]
    return vals, kern
def durations(dirpat, kernel_pat, exclude=None):
    d = []
    for f in glob.glob("gpurun_out/%s/*kernel_trace.csv" % dirpat):
        for r in csv.DictReader(open(f)):
            if kernel_pat in r["Kernel_Name"] and not (exclude and exclude in r["Kernel_Name"]):
                d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return d
def big_avg(v):
    b = [x for x in v if x > 0.5 * max(v)] if v else []
    return (sum(b) / len(b), len(b)) if b else (None, 0)
# ---- the headline scan
alg = 10_000_000 * 768 * 4
fe, kern = counters(TAG + "_prof_FETCH_SIZE", "scan_kernel<", ["FETCH_SIZE"], "scan_mq")
wr, _ = counters(TAG + "_prof_WRITE_SIZE", "scan_kernel<", ["WRITE_SIZE"], "scan_mq")
f, nf = big_avg(fe["FETCH_SIZE"]); w, nw = big_avg(wr["WRITE_SIZE"])
print("scan", f, nf, w, nw)
if f and w:
    name = json.load(open("gpurun_out/" + TAG + "_bench_under_rocprof.json"))["roofline"]["kernel"]
    traffic = (f * 2 + w) * 1024
    out = {"command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace --output-format csv -- python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras (scripts/gpu_prof.sh)",
           "kernel": name, "kernel_name_in_trace": kern, "kernel_source_sha256_16": bench.scan_source_hash(), "rows": 10_000_000, "dim": 768,
           "algorithmic_bytes_per_launch": alg, "FETCH_SIZE_raw_KB_avg": f, "FETCH_SIZE_launches": nf, "WRITE_SIZE_raw_KB_avg": w, "WRITE_SIZE_launches": nw,
           "fetch_bytes_corrected": f * 2 * 1024, "write_bytes_reported": w * 1024, "traffic_bytes_per_launch": traffic, "traffic_over_algorithmic": traffic / alg}
    json.dump(out, open("gpurun_out/" + TAG + "_scan_pmc_hbm_traffic.json", "w"), indent=1)
    print(json.dumps(out))
# ---- the hybrid tile kernel on the distinct-query stream
fe, kern = counters(TAG + "_prof_hyb_FETCH_SIZE", "hybrid_tile_kernel", ["FETCH_SIZE"])
wr, _ = counters(TAG + "_prof_hyb_WRITE_SIZE", "hybrid_tile_kernel", ["WRITE_SIZE"])
du = durations(TAG + "_prof_hyb_FETCH_SIZE", "hybrid_tile_kernel")
red = durations(TAG + "_prof_hyb_FETCH_SIZE", "hybrid_reduce_kernel")
dec = durations(TAG + "_prof_hyb_FETCH_SIZE", "decode_")  # (decode_dense_kernel since round 6; decode_blocks_* where it does not apply)
if fe["FETCH_SIZE"] and wr["WRITE_SIZE"]:
    f = sum(fe["FETCH_SIZE"]) / len(fe["FETCH_SIZE"]); w = sum(wr["WRITE_SIZE"]) / len(wr["WRITE_SIZE"])
    try:
        ab = json.load(open("gpurun_out/" + TAG + "_hybrid_stream_ab_freqs_only.json"))
    except Exception:
        ab = {}
    alg_h = 109_722_430.0
    out = {"command": "CYCLES=2 rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace -- python scripts/bench_hybrid_stream.py (16 distinct term pairs + query vectors, round-robin; warm and cold cycles)",
           "kernel_name_in_trace": kern, "launches": len(fe["FETCH_SIZE"]), "algorithmic_bytes_per_query": alg_h,
           "FETCH_SIZE_raw_KB_avg": f, "WRITE_SIZE_raw_KB_avg": w,
           "traffic_bytes_per_query_fetch_x2": (f * 2 + w) * 1024, "traffic_over_algorithmic_fetch_x2": (f * 2 + w) * 1024 / alg_h,
           "traffic_bytes_per_query_fetch_x1": (f + w) * 1024, "traffic_over_algorithmic_fetch_x1": (f + w) * 1024 / alg_h,
           "note": "the x2 correction of FETCH_SIZE is calibrated for wide streaming reads (MI355X_MICROARCH.md HBM); this kernel mixes streamed rows / postings with 4-8 byte gathers, so both readings are given",
           "tile_kernel_us_avg_under_pmc": sum(du) / max(len(du), 1), "reduce_kernel_us_avg_under_pmc": sum(red) / max(len(red), 1),
           "decode_kernel_us_avg_under_pmc": sum(dec) / max(len(dec), 1), "stream_record_same_process": ab}
    json.dump(out, open("gpurun_out/" + TAG + "_hybrid_tiles_pmc.json", "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "stream_record_same_process"}))
# ---- the fp32-native matrix-core pass
KP32 = os.environ.get("F32_KPAT", "gemm_qs_h8r_kernel")  # (the default FLOAT32 route since round 6; gemm_qs_f32_kernel with TUNING=gemm_qs_f8=0)
fe, kern = counters(TAG + "_prof_f32_FETCH_SIZE", KP32, ["FETCH_SIZE"])
wr, _ = counters(TAG + "_prof_f32_WRITE_SIZE", KP32, ["WRITE_SIZE"])
ck, _ = counters(TAG + "_prof_f32_clk", KP32, ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"])
du = durations(TAG + "_prof_f32_clk", KP32)
f, nf = big_avg(fe["FETCH_SIZE"]); w, nw = big_avg(wr["WRITE_SIZE"])
if f and du:
    long_us = [x for x in du if x > 0.5 * max(du)]
    g, _ = big_avg(ck["GRBM_GUI_ACTIVE"]); m, _ = big_avg(ck["SQ_VALU_MFMA_BUSY_CYCLES"])
    t_us = sum(long_us) / len(long_us)
    rows_long = 7_500_000  # the last filter phase: rows [n/4, n)
    out = {"command": "METRICS=cosine SHAPES=2 REPS=3 rocprofv3 --pmc ... --kernel-trace -- python scripts/bench_batch_f32.py; the LONG launches = the last filter phase over 7.5 M rows",
           "kernel_name_in_trace": kern, "long_launches": len(long_us), "long_launch_us_avg": t_us,
           "algorithmic_bytes_long_launch": rows_long * 768 * 4, "hbm_gbs_long_launch": rows_long * 768 * 4 / t_us / 1e3,
           "FETCH_SIZE_raw_KB_avg_long": f, "WRITE_SIZE_raw_KB_avg_long": w,
           "traffic_over_algorithmic": (f * 2 + (w or 0)) * 1024 / (rows_long * 768 * 4),
           "GRBM_GUI_ACTIVE_avg_long": g, "sclk_ghz": (g / 8 / t_us / 1e3) if g else None,
           "SQ_VALU_MFMA_BUSY_CYCLES_avg_long": m, "mfma_pipe_busy_frac": (m / 1024.0 / (g / 8)) if (g and m) else None,
           "mfma_tflops_long_launch": 2.0 * 256 * 768 * rows_long / t_us / 1e6}
    json.dump(out, open("gpurun_out/" + TAG + "_batch_f32_pmc.json", "w"), indent=1)
    print(json.dumps(out))
PY
find gpurun_out -name "*kernel_trace.csv" -size +1M -delete; find gpurun_out -name "*.db" -delete; find gpurun_out -name "*counter_collection.csv" -size +1M -delete
cut -c1-300 gpurun_out/${TAG}_bench_under_rocprof.json
