#!/bin/bash
# What the SHARED grids of the hybrid coalescer move: FETCH_SIZE / WRITE_SIZE per hybrid_tile_batch_kernel launch (eight caller
# threads on the configs[4] stream), by the number of tiles the launch carries -> gpurun_out/${TAG}_hybrid_saturation_traffic.json
set -u
TAG=${TAG:-r06}; export TAG TMPDIR=/tmp CYCLES=${CYCLES:-2} THREADS=8
R=$(pwd); mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && OUT=${TAG}_hyb_sat_$c.json timeout 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_prof_hybsat_$c" -o b -- python "$R/scripts/bench_hybrid_stream.py" > "$R/gpurun_out/${TAG}_prof_hybsat_$c.log" 2>&1)
done
python - <<'PY'
import csv, glob, json, os
TAG = os.environ["TAG"]
out = {"command": "scripts/diag/hybrid_saturation_traffic.sh (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python scripts/bench_hybrid_stream.py; THREADS=8)"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = {}
    dur = {}
    for f in glob.glob("gpurun_out/%s_prof_hybsat_%s/*kernel_trace.csv" % (TAG, c)):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for f in glob.glob("gpurun_out/%s_prof_hybsat_%s/*counter_collection.csv" % (TAG, c)):
        for r in csv.DictReader(open(f)):
            if "hybrid_tile" not in r["Kernel_Name"] or r["Counter_Name"] != c:
                continue
            key = ("batch" if "batch" in r["Kernel_Name"] else "single", int(r["Grid_Size"]) // 256)
            e = per.setdefault(key, {"launches": 0, "kb": 0.0, "us": 0.0})
            e["launches"] += 1
            e["kb"] += float(r["Counter_Value"])
            e["us"] += dur.get(r["Dispatch_Id"], 0.0)
    rows = []
    for (kind, tiles), e in sorted(per.items()):
        kb = e["kb"] / e["launches"]
        rows.append({"kernel": kind, "tiles": tiles, "launches": e["launches"], c + "_raw_KB_avg": kb, "us_avg_under_pmc": e["us"] / e["launches"],
                     "raw_KB_per_tile": kb / max(tiles, 1)})
    out[c] = rows
json.dump(out, open("gpurun_out/%s_hybrid_saturation_traffic.json" % TAG, "w"), indent=1)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in out[c]:
        print(c, r)
PY
tail -2 gpurun_out/${TAG}_prof_hybsat_FETCH_SIZE.log | cut -c1-300
