#!/bin/bash
# tools/r5_evidence.sh <part> -- round 5 evidence on the final build.  part 1: the full -m gpu suite; part 2: the headline's
# bench / rocprofv3 stats / PMC traffic (tools/profile_round.sh), the MFMA-busy counter pass, config 3 under rocprofv3,
# the n = 8192 timeline and panel stamps
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p gpurun_out
if [ "$1" = "1" ]; then
  timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 > gpurun_out/r05_pytest_final.log 2>&1
  rc=$?
  tail -22 gpurun_out/r05_pytest_final.log
  exit $rc
fi
timeout -k 10 200 bash tools/r3_trace.sh r05_timeline_n8192_final 8192 || exit 1
timeout -k 10 120 python tools/panel_stamps.py 8192 3 > gpurun_out/r05_panel_stamps_n8192_step3_final.log 2>&1 || exit 1
timeout -k 10 120 python tools/panel_stamps.py 8192 25 > gpurun_out/r05_panel_stamps_n8192_step25_final.log 2>&1 || exit 1
echo "[r5_evidence] profile_f32"
timeout -k 10 300 bash tools/profile_f32.sh r05 || exit 1
echo "[r5_evidence] pmc_mfma"
timeout -k 10 300 bash tools/pmc_mfma.sh r05 || exit 1
echo "[r5_evidence] profile_round"
timeout -k 10 900 bash tools/profile_round.sh r05 || exit 1
cd $ROOT
