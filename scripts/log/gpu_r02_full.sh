#!/bin/bash
# the whole GPU suite + smoke, as the driver runs them at round end
mkdir -p gpurun_out
# which box: three calls of round 2 landed on a box where EVERY process (the plain C client included) died at its first
# kernel with "Memory access fault by GPU node-2" while the same tree passed on every other box before and after
(hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i 'unique'; rocm-smi --showserial 2>/dev/null | grep -i serial) > gpurun_out/box_id.txt 2>&1
cat gpurun_out/box_id.txt
T0=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=8 > gpurun_out/r02_full_tests.txt 2>&1
echo "tests rc=$? t=$(( $(date +%s) - T0 ))s" >> gpurun_out/r02_full_tests.txt
tail -25 gpurun_out/r02_full_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
