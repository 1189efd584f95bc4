#!/bin/bash
# tools/r5_run2.sh -- round 5: tall route with the two-block inverse kernel: tests, timelines of both routes at N = 65536, A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r05_pytest2.log 2>&1
rc=$?
tail -5 gpurun_out/r05_pytest2.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 bash tools/r5_trace.sh r05_timeline_n65536_tall 65536 32 f64 || exit 1
timeout -k 10 200 bash tools/r5_trace.sh r05_timeline_n65536_resident 65536 32 f64 GPX_POTRF_TALL_ROWS=1099511627776 || exit 1
timeout -k 10 400 bash tools/r4_ab_sized.sh 65536 32 f64 2 "GPX_POTRF_TALL_ROWS=1099511627776" "GPX_POTRF_TALL_ROWS=16384" 2>&1 | tee gpurun_out/r05_ab_tall_n65536_v2.log || exit 1
timeout -k 10 300 bash tools/r4_ab_sized.sh 32768 16 f32 2 "GPX_POTRF_TALL_ROWS=1099511627776" "GPX_POTRF_TALL_ROWS=16384" 2>&1 | tee gpurun_out/r05_ab_tall_n32768_f32_v2.log || exit 1
