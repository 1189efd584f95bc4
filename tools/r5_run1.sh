#!/bin/bash
# tools/r5_run1.sh -- round 5, first GPU call: the tall-panel route (tests, then A/B at N = 65536 and N = 32768 fp32)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r05_pytest1.log 2>&1
rc=$?
tail -15 gpurun_out/r05_pytest1.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 bash tools/r4_ab_sized.sh 65536 32 f64 2 "GPX_POTRF_TALL_ROWS=1099511627776" "GPX_POTRF_TALL_ROWS=16384" "GPX_POTRF_TALL_ROWS=8192" 2>&1 | tee gpurun_out/r05_ab_tall_n65536.log || exit 1
timeout -k 10 300 bash tools/r4_ab_sized.sh 32768 16 f32 3 "GPX_POTRF_TALL_ROWS=1099511627776" "GPX_POTRF_TALL_ROWS=16384" "GPX_POTRF_TALL_ROWS=8192" 2>&1 | tee gpurun_out/r05_ab_tall_n32768_f32.log || exit 1
