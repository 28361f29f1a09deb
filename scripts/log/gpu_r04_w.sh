#!/bin/bash
# round 4, call w: the whole GPU suite on the final tree (after the windowed RSGPU_IntersectEx took the tile kernel)
R="$GRAFT_REPO_ROOT"; cd "$R" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 120 > gpurun_out/r04_final3_tests.txt 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r04_final3_tests.txt | cut -c1-300
