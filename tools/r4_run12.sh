#!/bin/bash
# tools/r4_run12.sh -- 512-wide outer blocks while many rows are left, now that the panel gets onto the chip before the update
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "soak or two_part or leaves or nested or outer_block or mlii or boundaries or gp_nd" > gpurun_out/r04_pytest12.log 2>&1
rc=$?
tail -4 gpurun_out/r04_pytest12.log
if [ $rc -ne 0 ]; then grep -n "Error\|assert" gpurun_out/r04_pytest12.log | head -20; fi
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out / was killed: no further GPU step"; exit $rc; fi
rm -f gpurun_out/r04_ab_widths_gate.log
timeout -k 10 600 bash tools/r3_ab.sh 8192 3 "GPX_X=1" "GPX_POTRF_WIDTHS=1,4096,8192" "GPX_POTRF_WIDTHS=1,5120,8192" "GPX_POTRF_WIDTHS=1,6144,8192" "GPX_POTRF_WIDTHS=1,5120,8192 GPX_PANEL_EXCL_ROWS=8192" "GPX_POTRF_WIDTHS=1,4096,8192 GPX_PANEL_EXCL_ROWS=8192" "GPX_POTRF_WIDTHS=1,5120,5120" "GPX_POTRF_WIDTHS=1,5120,5120 GPX_POTRF_NESTED=1" >> gpurun_out/r04_ab_widths_gate.log 2>&1 || exit 1
cat gpurun_out/r04_ab_widths_gate.log
timeout -k 10 200 bash tools/r3_trace.sh r04_timeline_n8192_nb512_gate 8192 GPX_POTRF_WIDTHS=1,5120,8192 || exit 1
exit $rc
