#!/bin/bash
# build first (the .so files travel with the snapshot: a stale library on the GPU box measures the previous edit), then gpurun
set -e
cd "$(dirname "$0")/.."
python -m redisearch_amd.build > /tmp/rsgpu_build.log 2>&1 || { tail -30 /tmp/rsgpu_build.log; exit 1; }
exec /usr/local/graft/bin/gpurun "$@"
