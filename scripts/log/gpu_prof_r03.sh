#!/bin/bash
# round 3 profiles: rocprofv3 kernel stats of the DRIVER'S bench command, the two HBM-traffic PMC passes over the headline
# loop (-> profiles/r03_scan_pmc_hbm_traffic.json, stamped with the kernel's source hash: bench.py refuses a stale one),
# and the same two counters over the multi-query scan (one pass over the corpus serves eight queries).
set -u
export TMPDIR=/tmp
R=$(pwd); mkdir -p gpurun_out
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r03_prof_stats" -o b -- python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$R/gpurun_out/r03_prof_stats.log" 2>&1)
grep "^{\"metric\"" gpurun_out/r03_prof_stats.log | tail -1 > gpurun_out/r03_bench_under_rocprof.json
f=$(find gpurun_out/r03_prof_stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r03_bench_kernel_stats.csv; cut -c1-180 "$f" | head -16
for grp in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$R/gpurun_out/r03_prof_$grp" -o b -- python "$R/bench.py" --steps 60 --warmup 5 --no-cpu-baseline --no-extras > "$R/gpurun_out/r03_prof_$grp.log" 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$R/gpurun_out/r03_prof_mq_$grp" -o b -- python "$R/scripts/bench_mq.py" --reps 6 > "$R/gpurun_out/r03_prof_mq_$grp.log" 2>&1)
done
python - <<'PY'
import csv, glob, json, sys
sys.path.insert(0, ".")
import bench
alg = 10_000_000 * 768 * 4
def collect(prefix, pat):
    res, kern = {}, None
    for grp in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob("gpurun_out/%s_%s/*counter_collection.csv" % (prefix, grp)):
            for r in csv.DictReader(open(f)):
                if pat in r["Kernel_Name"] and "mq" in pat or (pat in r["Kernel_Name"] and "scan_mq" not in r["Kernel_Name"]):
                    if r["Counter_Name"] == grp:
                        vals.append(float(r["Counter_Value"]))
                        kern = r["Kernel_Name"]
        big = [v for v in vals if v > 0.5 * max(vals)] if vals else []
        res[grp] = (sum(big) / len(big), len(big)) if big else (None, 0)
    return res, kern
res, kern = collect("r03_prof", "scan_kernel<")
print("scan", res)
if res["FETCH_SIZE"][0] and res["WRITE_SIZE"][0]:
    fetch, write = res["FETCH_SIZE"][0], res["WRITE_SIZE"][0]
    traffic = (fetch * 2 + write) * 1024   # MI355X_MICROARCH.md: KB units; FETCH_SIZE counts half the bytes on gfx950
    line = json.loads(open("gpurun_out/r03_prof_FETCH_SIZE.log").read().strip().splitlines()[-1]) if False else None
    # kernel name as the library reports it: from the bench line of the stats run
    name = json.load(open("gpurun_out/r03_bench_under_rocprof.json"))["roofline"]["kernel"]
    out = {"command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace --output-format csv -- python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras (scripts/gpu_prof_r03.sh)",
           "kernel": name, "kernel_name_in_trace": kern, "kernel_source_sha256_16": bench.scan_source_hash(), "rows": 10_000_000, "dim": 768,
           "algorithmic_bytes_per_launch": alg, "FETCH_SIZE_raw_KB_avg": fetch, "FETCH_SIZE_launches": res["FETCH_SIZE"][1],
           "WRITE_SIZE_raw_KB_avg": write, "WRITE_SIZE_launches": res["WRITE_SIZE"][1],
           "fetch_bytes_corrected": fetch * 2 * 1024, "write_bytes_reported": write * 1024,
           "traffic_bytes_per_launch": traffic, "traffic_over_algorithmic": traffic / alg}
    json.dump(out, open("gpurun_out/r03_scan_pmc_hbm_traffic.json", "w"), indent=1)
    print(json.dumps(out))
res, kern = collect("r03_prof_mq", "scan_mq_kernel")
print("mq", res)
if res["FETCH_SIZE"][0] and res["WRITE_SIZE"][0]:
    fetch, write = res["FETCH_SIZE"][0], res["WRITE_SIZE"][0]
    out = {"command": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace -- python scripts/bench_mq.py --reps 6 (passes of 2..8 queries; launches above half the maximum = all of them)",
           "kernel_name_in_trace": kern, "rows": 10_000_000, "dim": 768, "algorithmic_bytes_per_pass": alg,
           "FETCH_SIZE_raw_KB_avg": fetch, "WRITE_SIZE_raw_KB_avg": write, "launches": res["FETCH_SIZE"][1],
           "traffic_bytes_per_pass": (fetch * 2 + write) * 1024, "traffic_over_one_corpus_pass": (fetch * 2 + write) * 1024 / alg,
           "note": "one pass over the corpus whatever the number of queries it serves; the writes are the queries' key arrays (4 B per row and query)"}
    json.dump(out, open("gpurun_out/r03_mq_scan_pmc_hbm_traffic.json", "w"), indent=1)
    print(json.dumps(out))
PY
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete; find gpurun_out -name "*.db" -delete; find gpurun_out -name "*counter_collection.csv" -size +2M -delete
cut -c1-300 gpurun_out/r03_bench_under_rocprof.json
