#!/bin/bash
# Second GPU session: full parity suite (FLAT + search), bench with tuned defaults, rocprofv3 kernel stats
# (CSV) and PMC pass for the scan kernel's HBM traffic, hybrid (configs[4]) stage bench.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd)
echo "== pytest gpu"
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.txt
echo "== bench"
timeout 900 python bench.py --steps 300 --warmup 20 2>&1 | tail -3 | tee gpurun_out/bench_1gpu.txt
echo "== rocprof kernel stats"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -o bench -- python "$R/bench.py" --steps 100 --warmup 10 --no-cpu-baseline > "$R/gpurun_out/rocprof_stats.log" 2>&1)
find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} head -12 {}
echo "== rocprof pmc"
(cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/prof_pmc" -o bench -- python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$R/gpurun_out/rocprof_pmc.log" 2>&1)
(cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/prof_pmc_w" -o bench -- python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$R/gpurun_out/rocprof_pmc_w.log" 2>&1)
ls gpurun_out/prof_pmc gpurun_out/prof_pmc_w 2>/dev/null | head
python - <<'PY'
import csv, glob, collections
for d in ("prof_pmc", "prof_pmc_w"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "scan_kernel" in r.get("Kernel_Name", ""):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print(d, k, "n=%d avg=%.1f min=%.1f max=%.1f" % (len(v), sum(v) / len(v), min(v), max(v)))
PY
echo "== hybrid"
timeout 1200 python tests/bench_hybrid.py 2>&1 | tail -8 | tee gpurun_out/hybrid.txt
# keep the merged artefacts small
find gpurun_out -name "*.db" -delete; find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
