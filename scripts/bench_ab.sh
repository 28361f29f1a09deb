#!/bin/bash
# A/B of an engine knob on the headline config (same box, interleaved): bench_ab.sh key v1 v2
export TMPDIR=/tmp
KEY=${1:-filter_select}; A=${2:-1}; B=${3:-0}
for round in 1 2; do for v in $A $B; do
  timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --tuning $KEY=$v 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$KEY=$v qps=%.2f ms/step=%.4f kernel_ms=%.4f gbs=%.1f p50=%.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['achieved'], d['config']['p50_ms']))"
done; done
