#!/bin/bash
# tools/r5_run5.sh -- round 5: the full -m gpu suite on the build with the switch table, the leaf self-check and the margins; then n = 8192 / headline A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=10 > gpurun_out/r05_pytest5.log 2>&1
rc=$?
tail -25 gpurun_out/r05_pytest5.log
if [ $rc -ne 0 ]; then exit $rc; fi
DT=f64 timeout -k 10 200 bash tools/r3_ab.sh 8192 3 "GPX_X=1" "GPX_LEAF=1" 2>&1 | tee gpurun_out/r05_ab_n8192_leaf.log || exit 1
DT=f64 timeout -k 10 100 bash tools/r3_ab.sh 2048 3 "GPX_X=1" 2>&1 | tee -a gpurun_out/r05_ab_n8192_leaf.log || exit 1
