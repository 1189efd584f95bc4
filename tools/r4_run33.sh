#!/bin/bash
# tools/r4_run33.sh -- fp32: short panels on CUs of their own with the fp64 two-wave leaf (default) / LV = 5 everywhere / the fp32 leaf
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "float32 or f32 or fp32 or soak or dtype or configs" > gpurun_out/r04_pytest33.log 2>&1
rc=$?; tail -3 gpurun_out/r04_pytest33.log
if [ $rc -ne 0 ]; then grep -n "^FAILED\|Error" gpurun_out/r04_pytest33.log | head; exit $rc; fi
for n in 2048 4096 8192 16384; do timeout -k 10 400 bash tools/r4_ab_sized.sh $n 8 f32 2 "GPX_X=0" "GPX_LEAF=5" "GPX_LEAF=1" || exit 1; done
timeout -k 10 600 bash tools/r4_ab_sized.sh 32768 16 f32 2 "GPX_X=0" "GPX_LEAF=5" "GPX_LEAF=1" || exit 1
