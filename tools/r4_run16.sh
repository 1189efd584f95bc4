#!/bin/bash
# tools/r4_run16.sh -- the rescheduled one-wave leaf: leaf parity tests, panel stamps, potrf at n = 2048 / 4096 / 8192
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_round4.py -m gpu -q -p no:cacheprovider -k "leaves or soak" > gpurun_out/r04_pytest16.log 2>&1
rc=$?; tail -5 gpurun_out/r04_pytest16.log
if [ $rc -ne 0 ]; then grep -n "^FAILED\|Error" gpurun_out/r04_pytest16.log | head; exit $rc; fi
timeout -k 10 120 python tools/panel_stamps.py 8192 25 > gpurun_out/r04_panel_stamps_n8192_step25_leaf2.log 2>&1 || exit 1
cat gpurun_out/r04_panel_stamps_n8192_step25_leaf2.log
for n in 2048 4096 8192; do timeout -k 10 200 bash tools/r4_ab_sized.sh $n 8 f64 2 "GPX_X=0" || exit 1; done
