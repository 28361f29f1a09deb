#!/bin/bash
# The iterator library's host code (redisearch_amd/csrc/query_iterators.c) under AddressSanitizer + UBSan, driven by the CPU
# tests over the mock hit list (tests/test_iterator_host_cpu.py).  No GPU needed.
set -e
cd "$(dirname "$0")/.."
out=${TMPDIR:-/tmp}/rsgpu_asan
mkdir -p "$out"
gcc -O1 -g -std=gnu11 -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -fvisibility=hidden -Iinclude redisearch_amd/csrc/query_iterators.c tests/mock_hits.c -o "$out/libiter_mock.so" -ldl -Wl,-Bsymbolic
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 \
    RSGPU_ITER_MOCK_LIB="$out/libiter_mock.so" python -m pytest tests/test_iterator_host_cpu.py -x -q -p no:cacheprovider
