#!/bin/bash
# tools/r5_run4.sh -- round 5: the new parity tests (configs 2 / 3 / 4 against the oracle), the tall-route tests, spill-free LV = 5 kernel
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_round5.py tests/test_gpu_configs.py -m gpu -q -x -p no:cacheprovider --durations=8 > gpurun_out/r05_pytest4.log 2>&1
rc=$?
tail -25 gpurun_out/r05_pytest4.log
exit $rc
