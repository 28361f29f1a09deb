#!/bin/bash
# rocprofv3 kernel stats + HBM-traffic PMC passes of the default bench command (profiles/r01_*)
set -u
export TMPDIR=/tmp
R=$(pwd); mkdir -p gpurun_out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bench_stats" -o b -- python "$R/bench.py" > "$R/gpurun_out/prof_bench_stats.log" 2>&1)
tail -1 gpurun_out/prof_bench_stats.log | cut -c1-200
find gpurun_out/prof_bench_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "cut -c1-180 {} | head -8"
for grp in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$R/gpurun_out/prof_bench_$grp" -o b -- python "$R/bench.py" --steps 60 --warmup 5 --no-cpu-baseline --no-two-stage-extra > "$R/gpurun_out/prof_bench_$grp.log" 2>&1)
done
python - <<'PY'
import csv, glob, json
res = {}
for grp in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob("gpurun_out/prof_bench_%s/*counter_collection.csv" % grp):
        for r in csv.DictReader(open(f)):
            if "scan_kernel" in r["Kernel_Name"] and r["Counter_Name"] == grp:
                vals.append(float(r["Counter_Value"]))
    big = [v for v in vals if v > 0.5 * max(vals)] if vals else []
    res[grp] = sum(big) / len(big) if big else None
    print(grp, "launches", len(big), "avg", res[grp])
if res["FETCH_SIZE"] and res["WRITE_SIZE"]:
    # MI355X_MICROARCH.md: counters are in KB; on gfx950 FETCH_SIZE reports half of the bytes fetched
    traffic = (res["FETCH_SIZE"] * 2 + res["WRITE_SIZE"]) * 1024
    alg = 10_000_000 * 768 * 4
    out = {"rows": 10_000_000, "dim": 768, "fetch_size_kb": res["FETCH_SIZE"], "write_size_kb": res["WRITE_SIZE"],
           "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg, "ratio": traffic / alg,
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python bench.py --steps 60`, scan_kernel launches averaged; FETCH_SIZE x2 (gfx950), KB -> bytes"}
    json.dump(out, open("gpurun_out/scan_pmc_hbm_traffic.json", "w"), indent=1)
    print(json.dumps(out))
PY
find gpurun_out -name "*kernel_trace.csv" -size +4M -delete; find gpurun_out -name "*.db" -delete
