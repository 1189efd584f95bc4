#!/bin/bash
# tools/r4_run11.sh -- device-paced steps while many rows are left; the tests that changed
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "nested or mlii or leaves or soak" > gpurun_out/r04_pytest11.log 2>&1
rc=$?
tail -4 gpurun_out/r04_pytest11.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out / was killed: no further GPU step"; exit $rc; fi
rm -f gpurun_out/r04_ab_dev_paced.log
timeout -k 10 500 bash tools/r3_ab.sh 8192 3 "GPX_X=1" "GPX_POTRF_DEV_PACED_ROWS=3072" "GPX_POTRF_DEV_PACED_ROWS=4096" "GPX_POTRF_DEV_PACED_ROWS=5120" "GPX_POTRF_DEV_PACED_ROWS=6144" "GPX_POTRF_DEV_PACED_ROWS=0" >> gpurun_out/r04_ab_dev_paced.log 2>&1 || exit 1
timeout -k 10 300 bash tools/r3_ab.sh 4096 3 "GPX_X=1" "GPX_POTRF_DEV_PACED_ROWS=2048" "GPX_POTRF_DEV_PACED_ROWS=3072" >> gpurun_out/r04_ab_dev_paced.log 2>&1 || exit 1
timeout -k 10 400 bash tools/r3_ab.sh 12288 2 "GPX_X=1" "GPX_POTRF_DEV_PACED_ROWS=5120" "GPX_POTRF_DEV_PACED_ROWS=8192" >> gpurun_out/r04_ab_dev_paced.log 2>&1 || exit 1
timeout -k 10 400 bash tools/r3_ab.sh 16384 2 "GPX_X=1" "GPX_POTRF_DEV_PACED_ROWS=5120" "GPX_POTRF_DEV_PACED_ROWS=8192" >> gpurun_out/r04_ab_dev_paced.log 2>&1 || exit 1
cat gpurun_out/r04_ab_dev_paced.log
timeout -k 10 200 bash tools/r3_trace.sh r04_timeline_n8192_dev_paced 8192 GPX_POTRF_DEV_PACED_ROWS=4096 || exit 1
exit $rc
