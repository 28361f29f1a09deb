#!/bin/bash
# round 3, call k: specialised qint decode (parity over every codec, cold-decode timing), L2 batched re-check
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_proximity.py tests/test_gpu_docid64.py tests/test_gpu_index_mutations.py tests/test_gpu_boolean.py tests/test_gpu_intersection_kats.py -x -q -p no:cacheprovider > gpurun_out/r03k_search.txt 2>&1; echo "search rc=$?"
tail -4 gpurun_out/r03k_search.txt
timeout 300 python tests/make_decode_lists.py /tmp/lists.npz > /dev/null 2>&1; echo "lists rc=$?"
timeout 300 python scripts/bench_decode.py /tmp/lists.npz > gpurun_out/r03k_decode.txt 2>&1; echo "decode rc=$?"
cat gpurun_out/r03k_decode.txt | cut -c1-250
cp gpurun_out/decode_bench.json gpurun_out/r03k_decode_bench.json 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_batch_l2.py tests/test_gpu_batch_i8_shadow.py tests/test_gpu_batch_f32_shadow.py -x -q -p no:cacheprovider > gpurun_out/r03k_l2.txt 2>&1; echo "l2 rc=$?"
tail -4 gpurun_out/r03k_l2.txt
timeout 600 python scripts/bench_batch_l2.py > gpurun_out/r03k_bench_l2.json 2> gpurun_out/r03k_bench_l2.err; echo "bench rc=$?"
cat gpurun_out/r03k_bench_l2.json
