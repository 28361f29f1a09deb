#!/bin/bash
# round 2 profiles: rocprofv3 kernel stats of the default bench command (headline + extras) and the two HBM-traffic PMC
# passes over the headline loop -> gpurun_out/r02_* (copied to profiles/ by hand)
set -u
export TMPDIR=/tmp
R=$(pwd); mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r02_bench_1gpu.json 2> gpurun_out/r02_bench_1gpu.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r02_prof_stats" -o b -- python "$R/bench.py" > "$R/gpurun_out/r02_prof_stats.log" 2>&1)
grep "^{\"metric\"" gpurun_out/r02_prof_stats.log | tail -1 > gpurun_out/r02_bench_under_rocprof.json
f=$(find gpurun_out/r02_prof_stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02_bench_kernel_stats.csv; cut -c1-200 "$f" | head -30
for grp in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$R/gpurun_out/r02_prof_$grp" -o b -- python "$R/bench.py" --steps 60 --warmup 5 --no-cpu-baseline --no-extras > "$R/gpurun_out/r02_prof_$grp.log" 2>&1)
done
python - <<'PY'
import csv, glob, json
res, kern = {}, None
for grp in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob("gpurun_out/r02_prof_%s/*counter_collection.csv" % grp):
        for r in csv.DictReader(open(f)):
            if "scan_kernel" in r["Kernel_Name"] and r["Counter_Name"] == grp:
                vals.append(float(r["Counter_Value"]))
                kern = r["Kernel_Name"]
    big = [v for v in vals if v > 0.5 * max(vals)] if vals else []
    res[grp] = (sum(big) / len(big), len(big)) if big else (None, 0)
    print(grp, res[grp])
if res["FETCH_SIZE"][0] and res["WRITE_SIZE"][0]:
    fetch, write = res["FETCH_SIZE"][0], res["WRITE_SIZE"][0]
    traffic = (fetch * 2 + write) * 1024   # MI355X_MICROARCH.md: KB units; FETCH_SIZE counts half the bytes on gfx950
    alg = 10_000_000 * 768 * 4
    line = json.loads(open("gpurun_out/r02_bench_1gpu.json").read().strip().splitlines()[-1])
    out = {"command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace --output-format csv -- python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras (scripts/gpu_prof_r02.sh)",
           "kernel": line["roofline"]["kernel"], "kernel_name_in_trace": kern, "rows": 10_000_000, "dim": 768,
           "algorithmic_bytes_per_launch": alg, "FETCH_SIZE_raw_KB_avg": fetch, "FETCH_SIZE_launches": res["FETCH_SIZE"][1],
           "WRITE_SIZE_raw_KB_avg": write, "WRITE_SIZE_launches": res["WRITE_SIZE"][1],
           "fetch_bytes_corrected": fetch * 2 * 1024, "write_bytes_reported": write * 1024,
           "traffic_bytes_per_launch": traffic, "traffic_over_algorithmic": traffic / alg}
    json.dump(out, open("gpurun_out/r02_scan_pmc_hbm_traffic.json", "w"), indent=1)
    print(json.dumps(out))
PY
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete; find gpurun_out -name "*.db" -delete; find gpurun_out -name "*counter_collection.csv" -size +2M -delete
cut -c1-400 gpurun_out/r02_bench_1gpu.json
