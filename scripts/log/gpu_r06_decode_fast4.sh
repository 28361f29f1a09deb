#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_decode_dense.py tests/test_gpu_decode_qint.py tests/test_gpu_hybrid_query.py tests/test_gpu_fullsize_hybrid.py -x -q -m gpu 2>&1 | tail -3
for c in freqs_only full; do
  CODEC=$c MODES=warm,cold CONFIGS="dense:;chain:decode_dense=0;dense2:;chain2:decode_dense=0" OUT=r06_decode_dense_ab_$c.json timeout 600 python scripts/bench_hybrid_stream.py 2>&1 | grep -o '^[a-z0-9_]* \|"cold_p50": [0-9.]*\|"cold_dev": {[^}]*}\|"cold_decode_gbs": [0-9.]*\|"same_answers_as_first_config": [a-z]*' | paste - - - - -
done
