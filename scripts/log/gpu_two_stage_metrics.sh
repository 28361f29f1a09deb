#!/bin/bash
# opt-in two-stage exact scan (int8 shadow) under the three metrics and at larger K: QPS of the same bench, same corpus
mkdir -p gpurun_out; : > gpurun_out/r02_two_stage_metrics.txt
for m in cosine ip l2; do
  for k in 10 100 1000; do
    for t in "shadow8=0" "shadow8=1"; do
      timeout 300 python bench.py --metric $m --k $k --tuning $t --no-extras --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$m k=$k $t', 'qps', round(d['value'], 1), 'ms', round(d['ms_per_step'], 3), 'verify', d['config']['verify']['ok'], d['roofline']['kernel'][:40])" | tee -a gpurun_out/r02_two_stage_metrics.txt
    done
  done
done
