#!/bin/bash
# tools/r4_run31.sh -- round 4 evidence: full -m gpu suite, then the headline's bench / rocprofv3 stats / PMC passes,
# config 3 under rocprofv3, the n = 8192 timeline and panel stamps of the final build
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_pytest31.log 2>&1
rc=$?
tail -5 gpurun_out/r04_pytest31.log
if [ $rc -ne 0 ]; then grep -n "^FAILED\|Error" gpurun_out/r04_pytest31.log | head -20; fi
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out / was killed: no further GPU step"; exit $rc; fi
timeout -k 10 200 bash tools/r3_trace.sh r04_timeline_n8192_final2 8192 || exit 1
timeout -k 10 120 python tools/panel_stamps.py 8192 3 > gpurun_out/r04_panel_stamps_n8192_step3_final2.log 2>&1 || exit 1
timeout -k 10 120 python tools/panel_stamps.py 8192 25 > gpurun_out/r04_panel_stamps_n8192_step25_final2.log 2>&1 || exit 1
echo "[r4_run31] profile_f32"
timeout -k 10 400 bash tools/profile_f32.sh r04 || exit 1
echo "[r4_run31] profile_round"
timeout -k 10 1000 bash tools/profile_round.sh r04 || exit 1
cd $ROOT
exit $rc
