#!/bin/bash
# First GPU step of the overlapped-tile-boundary follow-up (profiles/r02_batch_qs_experiments.txt): apply the patch to the
# box's copy of the tree, rebuild, and run the ONE failing case with the runtime's logging on, so that the silent SIGABRT of round 2 gets a
# message.  Leaves the tree patched on the box only (the box is discarded).  Usage (from the repo root, on the GPU box):
#   bash scripts/diag/ovl_diagnose.sh            -> gpurun_out/ovl_diag_*.txt
set -u
mkdir -p gpurun_out
patch -p1 < scripts/diag/gemm_qs_overlapped_tile_boundary.patch > gpurun_out/ovl_diag_patch.txt 2>&1 || { echo "patch does not apply"; exit 1; }
python -c "from redisearch_amd import build; build.build()" > gpurun_out/ovl_diag_build.txt 2>&1 || { echo "build failed"; tail -5 gpurun_out/ovl_diag_build.txt; exit 1; }
CASE='tests/test_gpu_batch_i8_shadow.py::test_f32_index_batches_through_the_int8_rows'
# the knob defaults to 0: scripts/diag/tuning_plugin.py sets RSGPU_TUNING=key=value,... through RSGPU_SetTuning at pytest start-up
export PYTHONPATH=scripts/diag${PYTHONPATH:+:$PYTHONPATH}
run() {  # name, env...
  local name=$1; shift
  env "$@" RSGPU_TUNING=qs_ovl=1 timeout 150 python -m pytest -p tuning_plugin "$CASE" -x -q -p no:cacheprovider -k "768" > gpurun_out/ovl_diag_$name.txt 2>&1
  echo "$name: rc=$?"; grep -m5 -i "error\|fault\|abort\|exception\|passed\|failed" gpurun_out/ovl_diag_$name.txt
}
run plain
run blocking HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3
run logged AMD_LOG_LEVEL=3 HIP_LAUNCH_BLOCKING=1
# glibc's heap-corruption messages go to /dev/tty unless told otherwise: the round-2 logs show a bare SIGABRT, which is what a
# corrupted host heap looks like then (the reply sorts were not NaN-safe at the time; they are now)
run heapcheck LIBC_FATAL_STDERR_=1 MALLOC_CHECK_=3
tail -c 20000 gpurun_out/ovl_diag_logged.txt > gpurun_out/ovl_diag_logged_tail.txt; rm -f gpurun_out/ovl_diag_logged.txt
# and with the knob off on the same build: the patched library must behave like the shipped one
RSGPU_TUNING=qs_ovl=0 timeout 150 python -m pytest -p tuning_plugin "$CASE" -x -q -p no:cacheprovider > gpurun_out/ovl_diag_knob_off.txt 2>&1; echo "knob off: rc=$?"
