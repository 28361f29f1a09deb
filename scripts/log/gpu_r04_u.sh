#!/bin/bash
# round 4, call u: the proximity cursors read a term's offsets blob from eight prefetched bytes (one round trip per candidate
# instead of a dependent byte load per varint): parity of every consumer, the general kernel's shapes again
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_proximity.py tests/test_gpu_tree.py tests/test_gpu_hybrid_general.py tests/test_gpu_iterators.py tests/test_gpu_reference_hybrid_reader.py tests/test_gpu_hybrid_tiles.py tests/test_gpu_hybrid_query.py tests/test_gpu_search.py tests/test_gpu_over_limit.py tests/test_gpu_boolean.py tests/test_gpu_intersection_kats.py -q -p no:cacheprovider --timeout 150 -rf > gpurun_out/r04u_tests.txt 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r04u_tests.txt | cut -c1-300
(cd /tmp && SKIP_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r04r_stats" -o g -- python "$R/scripts/bench_hybrid_general.py" > "$R/gpurun_out/r04u_shapes.txt" 2>&1); echo "prof rc=$?"
grep -E "^(hit_list|tfidf|term_and|two_terms|eval_tree)" gpurun_out/r04u_shapes.txt | cut -c1-330
f=$(find gpurun_out/r04r_stats -name "*kernel_stats.csv" | head -1); grep -E "prox_|hybrid_tree" "$f" | cut -d, -f1-4 | cut -c1-200
rm -rf gpurun_out/r04r_stats
