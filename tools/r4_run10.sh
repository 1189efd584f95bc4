#!/bin/bash
# tools/r4_run10.sh -- round 4: full -m gpu suite on the current defaults, then host-spin / exclusive-CU A/B over sizes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_pytest10.log 2>&1
rc=$?
tail -5 gpurun_out/r04_pytest10.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out / was killed: no further GPU step"; exit $rc; fi
rm -f gpurun_out/r04_ab_spin_excl.log
for n in 8192 4096 2048; do
  timeout -k 10 400 bash tools/r3_ab.sh $n 3 "GPX_X=1" "GPX_POTRF_SPIN_US=0" "GPX_PANEL_EXCL_ROWS=0" "GPX_PANEL_EXCL_ROWS=0 GPX_POTRF_SPIN_US=0" >> gpurun_out/r04_ab_spin_excl.log 2>&1 || exit 1
done
for n in 12288 16384; do
  timeout -k 10 400 bash tools/r3_ab.sh $n 2 "GPX_X=1" "GPX_PANEL_EXCL_ROWS=0 GPX_POTRF_SPIN_US=0" >> gpurun_out/r04_ab_spin_excl.log 2>&1 || exit 1
done
DT=f32 timeout -k 10 300 bash tools/r3_ab.sh 8192 2 "GPX_X=1" "GPX_POTRF_SPIN_US=0" >> gpurun_out/r04_ab_spin_excl.log 2>&1 || exit 1
cat gpurun_out/r04_ab_spin_excl.log
exit $rc
