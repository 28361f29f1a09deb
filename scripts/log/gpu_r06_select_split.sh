#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_hybrid_query.py tests/test_gpu_hybrid_tiles.py tests/test_gpu_hybrid_general.py tests/test_gpu_hybrid_nested.py tests/test_gpu_hybrid_coalesce.py tests/test_gpu_fullsize_hybrid.py tests/test_gpu_hybrid_mutated.py -x -q -m gpu 2>&1 | tail -3
CODEC=freqs_only MODES=warm CONFIGS="a:;b:;c:" THREADS=8,16 OUT=r06_early_out.json timeout 900 python scripts/bench_hybrid_stream.py 2>&1 | grep -o '^[a-z0-9_]* \|"warm_p50": [0-9.]*\|"warm_dev": {[^}]*}\|"same_answers_as_first_config": [a-z]*\|"qps": [0-9.]*' | paste - - - - - -
