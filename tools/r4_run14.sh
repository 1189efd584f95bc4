#!/bin/bash
# tools/r4_run14.sh -- the pair phase (far update at K = 512 once per two panels): tests, A/B over thresholds, timeline
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "pair_phase" > gpurun_out/r04_pytest14.log 2>&1
rc=$?
tail -4 gpurun_out/r04_pytest14.log
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|Mismatch\|Max " gpurun_out/r04_pytest14.log | head -30; fi
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out / was killed: no further GPU step"; exit $rc; fi
if [ $rc -ne 0 ]; then exit $rc; fi
rm -f gpurun_out/r04_ab_pairs.log
timeout -k 10 600 bash tools/r3_ab.sh 8192 3 "GPX_X=1" "GPX_POTRF_PAIR_ROWS=3072" "GPX_POTRF_PAIR_ROWS=4096" "GPX_POTRF_PAIR_ROWS=4608" "GPX_POTRF_PAIR_ROWS=5120" "GPX_POTRF_PAIR_ROWS=5632" "GPX_POTRF_PAIR_ROWS=6144" >> gpurun_out/r04_ab_pairs.log 2>&1 || exit 1
timeout -k 10 300 bash tools/r3_ab.sh 4096 3 "GPX_X=1" "GPX_POTRF_PAIR_ROWS=2048" "GPX_POTRF_PAIR_ROWS=3072" >> gpurun_out/r04_ab_pairs.log 2>&1 || exit 1
cat gpurun_out/r04_ab_pairs.log
timeout -k 10 200 bash tools/r3_trace.sh r04_timeline_n8192_pairs 8192 GPX_POTRF_PAIR_ROWS=4608 || exit 1
