#!/bin/bash
# tools/r6_evidence.sh <part> -- round 6 evidence on the final build.  part 1: the full -m gpu suite with durations; part 2: the
# headline's bench / rocprofv3 stats / PMC traffic (tools/profile_round.sh), the MFMA-busy counter pass, config 3 under
# rocprofv3, the n = 8192 timeline and panel stamps, the GEMM probe (two / one workgroup per CU, ablations)
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p gpurun_out
if [ "$1" = "1" ]; then
  timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/r06_pytest_final.log 2>&1
  rc=$?
  tail -26 gpurun_out/r06_pytest_final.log
  exit $rc
fi
timeout -k 10 120 python tools/r6_gemm_probe.py > gpurun_out/r06_gemm_probe_final.log 2>&1 || exit 1
timeout -k 10 200 bash tools/r3_trace.sh r06_timeline_n8192_final 8192 || exit 1
timeout -k 10 120 python tools/panel_stamps.py 8192 25 > gpurun_out/r06_panel_stamps_n8192_step25_final.log 2>&1 || exit 1
echo "[r6_evidence] profile_f32"
timeout -k 10 300 bash tools/profile_f32.sh r06 || exit 1
echo "[r6_evidence] pmc_mfma"
timeout -k 10 300 bash tools/pmc_mfma.sh r06 || exit 1
echo "[r6_evidence] profile_round"
timeout -k 10 900 bash tools/profile_round.sh r06 || exit 1
cd $ROOT
