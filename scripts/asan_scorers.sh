#!/bin/bash
# The scorer plugin (redisearch_amd/csrc/scorer_plugin.c -- CPU code a RediSearch module loads with EXTLOAD) under
# AddressSanitizer + UBSan, driven by its CPU tests (random result trees vs the reference's compiled default.c).  No GPU needed.
set -e
cd "$(dirname "$0")/.."
out=${TMPDIR:-/tmp}/rsgpu_asan
mkdir -p "$out"
gcc -O1 -g -std=gnu11 -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -fvisibility=hidden -ffp-contract=off -Iinclude redisearch_amd/csrc/scorer_plugin.c -o "$out/librsgpu_scorers.so" -ldl -lm
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 \
    RSGPU_SCORERS_LIB="$out/librsgpu_scorers.so" python -m pytest tests/test_scorer_plugin.py -x -q -p no:cacheprovider \
    --deselect tests/test_scorer_plugin.py::test_exports_and_undefined_symbols
