#!/bin/bash
# round 3, call x (last): the whole GPU suite + smoke + the driver's bench command, the kernel stats of that command without the
# concurrent-callers extra, and the HBM traffic (PMC, separate passes) + timeline of the two hybrid kernels
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
bash scripts/gpu_r03_full.sh
bash scripts/gpu_prof_r03_stats.sh
for grp in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && REPS=30 timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$R/gpurun_out/r03_hyb_$grp" -o h -- python "$R/tests/hybrid_fused_prof.py" > "$R/gpurun_out/r03_hyb_$grp.log" 2>&1)
done
(cd /tmp && REPS=40 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/hybrid_prof" -o h -- python "$R/tests/hybrid_fused_prof.py" > "$R/gpurun_out/hybrid_prof.log" 2>&1)
python - <<'PY'
import csv, glob, json, re
out = {"workload": "BASELINE configs[4] through RSGPU_HybridQuery (tests/hybrid_fused_prof.py), two launches per query",
       "algorithmic_bytes": {"decoded postings read (4 B x 7.5 M)": 30_000_000, "per hit (frequency of list 2, doc length, doc score: 12 B x 249 792)": 2_997_504,
                             "vector rows (3 072 B x 25 k hits with a vector)": 76_800_000}}
for grp in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = {}
    for f in glob.glob("gpurun_out/r03_hyb_%s/**/*counter_collection.csv" % grp, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == grp and "hybrid_" in r["Kernel_Name"]:
                name = "tile" if "hybrid_tile" in r["Kernel_Name"] else "reduce"
                vals.setdefault(name, []).append(float(r["Counter_Value"]))
    for name, v in vals.items():
        v = v[len(v) // 2:]          # the later launches: everything warm
        out.setdefault(name, {})[grp + "_raw_KB_avg"] = sum(v) / len(v)
        out[name][grp + "_launches"] = len(v)
for name in ("tile", "reduce"):
    if name in out and "FETCH_SIZE_raw_KB_avg" in out[name] and "WRITE_SIZE_raw_KB_avg" in out[name]:
        out[name]["traffic_bytes_per_launch"] = (out[name]["FETCH_SIZE_raw_KB_avg"] * 2 + out[name]["WRITE_SIZE_raw_KB_avg"]) * 1024   # MI355X_MICROARCH.md: KB units, FETCH_SIZE counts half the bytes on gfx950
f = glob.glob("gpurun_out/hybrid_prof/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if "hybrid_" in r["Name"]:
            out.setdefault("tile" if "hybrid_tile" in r["Name"] else "reduce", {})["rocprofv3_avg_us"] = float(r["AverageNs"]) / 1e3
            out["tile" if "hybrid_tile" in r["Name"] else "reduce"]["rocprofv3_calls"] = int(r["Calls"])
alg = sum(out["algorithmic_bytes"].values())
if "tile" in out and "traffic_bytes_per_launch" in out["tile"] and "rocprofv3_avg_us" in out["tile"]:
    out["tile"]["traffic_over_algorithmic"] = out["tile"]["traffic_bytes_per_launch"] / alg
    out["tile"]["algorithmic_gbs"] = alg / out["tile"]["rocprofv3_avg_us"] / 1e3
    out["tile"]["frac_of_8000_gbs"] = alg / out["tile"]["rocprofv3_avg_us"] / 1e3 / 8000
json.dump(out, open("gpurun_out/r03_hybrid_tiles_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
grep HYBRID_FUSED gpurun_out/hybrid_prof.log | tail -1
find gpurun_out -name "*kernel_trace.csv" -size +1M -delete; find gpurun_out -name "*counter_collection.csv" -size +1M -delete
