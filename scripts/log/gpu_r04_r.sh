#!/bin/bash
# round 4, call r: rocprofv3 kernel stats of the general tile kernel's shapes (+ RSGPU_EvalTree alone)
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
(cd /tmp && SKIP_STREAM=1 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r04r_stats" -o g -- python "$R/scripts/bench_hybrid_general.py" > "$R/gpurun_out/r04r_shapes.txt" 2>&1); echo "prof rc=$?"
tail -6 gpurun_out/r04r_shapes.txt | cut -c1-400
f=$(find gpurun_out/r04r_stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r04_general_kernel_stats.csv; cut -c1-200 "$f" | head -24
# per-launch durations of the tile kernels by launch order (the shapes run one after the other: general, then staged)
python - <<'PY'
import csv, glob, json, statistics
# per-shape durations of the general tile kernel: the script runs its shapes one after the other, each as general (40 launches:
# a first run + a repeat of nothing -- 8 queries -- then 3 timed cycles) and staged (none of this kernel)
rows = []
for f in glob.glob("gpurun_out/r04r_stats/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for key in ("hybrid_tree_tile_kernel", "hybrid_hits_pack_kernel", "hybrid_reduce_kernel"):
            if key in n:
                rows.append((int(r["Start_Timestamp"]), key, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
tiles = [d for _, k, d in rows if k == "hybrid_tree_tile_kernel"]
packs = [d for _, k, d in rows if k == "hybrid_hits_pack_kernel"]
names = ["hit_list_wanted_freqs_only_bm25std_knn", "tfidf_docnorm_over_full_codec_slop_from_offsets_knn", "term_and_union_of_two_freqs_only_bm25std_knn",
         "two_terms_max_slop_30_full_codec_bm25std_knn"]
out = {"tree_tile_launches": len(tiles), "per_shape_tile_kernel_us": {}}
for i, nm in enumerate(names):
    ds = tiles[i * 32:(i + 1) * 32]
    if ds:
        out["per_shape_tile_kernel_us"][nm] = {"launches": len(ds), "median": statistics.median(ds), "min": min(ds), "max": max(ds)}
rest = tiles[len(names) * 32:]
if rest:
    out["eval_tree_tile_kernel_us"] = {"launches": len(rest), "median": statistics.median(rest)}
if packs:
    out["hits_pack_kernel_us"] = {"launches": len(packs), "median": statistics.median(packs), "max": max(packs)}
json.dump(out, open("gpurun_out/r04_general_kernel_durations.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf gpurun_out/r04r_stats
