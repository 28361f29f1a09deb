#!/bin/bash
# round 4, call c: hybrid parity (tiles, query, decode with offsets through the sync points), then the knob A/B on the stream
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2
timeout 1200 python -m pytest tests/test_gpu_hybrid_tiles.py tests/test_gpu_hybrid_query.py tests/test_gpu_decode_qint.py tests/test_gpu_proximity.py tests/test_gpu_search.py -x -q -p no:cacheprovider > gpurun_out/r04c_tests.txt 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/r04c_tests.txt
timeout 900 python scripts/bench_hybrid_stream.py > gpurun_out/r04c_stream.txt 2>&1; echo "stream rc=$?"
tail -8 gpurun_out/r04c_stream.txt
CODEC=full CYCLES=2 timeout 900 python scripts/bench_hybrid_stream.py > gpurun_out/r04c_stream_full.txt 2>&1; echo "stream full rc=$?"
tail -8 gpurun_out/r04c_stream_full.txt
