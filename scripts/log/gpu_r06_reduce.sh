#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_hybrid_query.py tests/test_gpu_hybrid_tiles.py tests/test_gpu_hybrid_general.py tests/test_gpu_hybrid_nested.py tests/test_gpu_hybrid_coalesce.py tests/test_gpu_fullsize_hybrid.py tests/test_gpu_hybrid_mutated.py tests/test_gpu_hybrid_concurrent.py tests/test_gpu_over_limit.py -x -q -m gpu 2>&1 | tail -2
CODEC=freqs_only MODES=warm CONFIGS="a:;b:;c:" THREADS=8,16 OUT=r06_reduce.json timeout 900 python scripts/bench_hybrid_stream.py 2>&1 | grep -o '^[a-z0-9_]* \|"warm_p50": [0-9.]*\|"warm_dev": {[^}]*}\|"same_answers_as_first_config": [a-z]*\|"qps": [0-9.]*' | paste - - - - - -
timeout 600 python scripts/hybrid_trace_stream.py 2>&1 | tail -2 | python -c "
import sys,json
for line in sys.stdin:
    try: r=json.loads(line)
    except Exception: continue
    print(r['kernel_span_us'], r['reduce_kernel_us_after_tile_kernel_end'])
"
