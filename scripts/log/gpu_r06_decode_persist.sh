#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_decode_qint.py tests/test_gpu_search.py tests/test_gpu_hybrid_query.py -x -q -m gpu 2>&1 | tail -2
for c in freqs_only; do
  CODEC=$c MODES=warm,cold CONFIGS="chunk_img:;chunk_noimg:decode_persistent_image=0;stride_img:decode_persistent_chunk=0;stride_noimg:decode_persistent_chunk=0,decode_persistent_image=0;classic:decode_persistent=0;chunk_noimg16:decode_persistent_image=0,decode_persistent_per_cu=16;chunk_noimg8:decode_persistent_image=0,decode_persistent_per_cu=8;chunk_img8:decode_persistent_per_cu=8;chunk_img4:decode_persistent_per_cu=4;classic2:decode_persistent=0" OUT=r06_decode_persistent_v4_ab_$c.json timeout 900 python scripts/bench_hybrid_stream.py 2>&1 | grep -o '^[a-z0-9_]* \|"cold_dev": {[^}]*}\|"same_answers_as_first_config": [a-z]*' | paste - - -
done
