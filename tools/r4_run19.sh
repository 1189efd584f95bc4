#!/bin/bash
# tools/r4_run19.sh -- quick parity (leaves, soak, round-4 file) on the current build, then A/B against tools/ab/libgpx_before.so
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_round4.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r04_pytest19.log 2>&1
rc=$?; tail -3 gpurun_out/r04_pytest19.log
if [ $rc -ne 0 ]; then grep -n "^FAILED\|Error" gpurun_out/r04_pytest19.log | head; exit $rc; fi
bash tools/r4_run17.sh
