#!/bin/bash
# tools/r4_run18.sh -- full -m gpu suite on the build with the two-wave leaf, then the hazard probe's table for profiles/
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_pytest18.log 2>&1
rc=$?
tail -5 gpurun_out/r04_pytest18.log
if [ $rc -ne 0 ]; then grep -n "^FAILED\|Error" gpurun_out/r04_pytest18.log | head -20; exit $rc; fi
timeout -k 5 60 ./tools/mfma_hazard_probe > gpurun_out/r04_mfma_hazard_probe.txt 2>&1 || exit 1
timeout -k 5 60 ./tools/leaf_probe_s5 > gpurun_out/r04_leaf_probe.txt 2>&1; timeout -k 5 60 ./tools/leaf_probe_s-1 >> gpurun_out/r04_leaf_probe.txt 2>&1
tail -3 gpurun_out/r04_leaf_probe.txt
