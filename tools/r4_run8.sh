#!/bin/bash
# tools/r4_run8.sh -- round 4: a CU of its own for every workgroup of a short panel (one-wave leaf + LDS padding): stamps and A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/r04_leaf_steps_excl.log gpurun_out/r04_ab_excl.log
for at in 25 12 3; do
  echo "== GPX_PANEL_EXCL_ROWS=8192 launch $at" >> gpurun_out/r04_leaf_steps_excl.log
  GPX_PANEL_EXCL_ROWS=8192 timeout -k 10 120 python tools/panel_stamps.py 8192 $at 2>&1 | grep -E "leaf|steps|core|wg 1|wg 2|wg 3|^   0 |^   1 |^   2 |^   3 |^   4 |last" >> gpurun_out/r04_leaf_steps_excl.log || exit 1
done
cat gpurun_out/r04_leaf_steps_excl.log
timeout -k 10 500 bash tools/r3_ab.sh 8192 3 "GPX_X=1" "GPX_PANEL_EXCL_ROWS=2048" "GPX_PANEL_EXCL_ROWS=3072" "GPX_PANEL_EXCL_ROWS=4096" "GPX_PANEL_EXCL_ROWS=5120" "GPX_PANEL_EXCL_ROWS=6144" "GPX_PANEL_EXCL_ROWS=8192" "GPX_LEAF=4" >> gpurun_out/r04_ab_excl.log 2>&1 || exit 1
timeout -k 10 300 bash tools/r3_ab.sh 4096 3 "GPX_X=1" "GPX_PANEL_EXCL_ROWS=2048" "GPX_PANEL_EXCL_ROWS=4096" >> gpurun_out/r04_ab_excl.log 2>&1 || exit 1
timeout -k 10 300 bash tools/r3_ab.sh 12288 2 "GPX_X=1" "GPX_PANEL_EXCL_ROWS=4096" "GPX_PANEL_EXCL_ROWS=6144" >> gpurun_out/r04_ab_excl.log 2>&1 || exit 1
cat gpurun_out/r04_ab_excl.log
timeout -k 10 200 bash tools/r3_trace.sh r04_timeline_n8192_excl4096 8192 GPX_PANEL_EXCL_ROWS=4096 || exit 1
