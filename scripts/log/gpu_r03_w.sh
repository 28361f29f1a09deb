#!/bin/bash
# round 3, call w: the whole GPU suite + smoke + the driver's bench command (scripts/gpu_r03_full.sh), then the kernel stats of
# that command without the concurrent-callers extra (scripts/gpu_prof_r03_stats.sh)
R="$GRAFT_REPO_ROOT"; cd "$R"
bash scripts/gpu_r03_full.sh
bash scripts/gpu_prof_r03_stats.sh
