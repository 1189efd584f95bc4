#!/bin/bash
# tools/r4_run9.sh -- exclusive CUs for short panels, with enough padding this time (48 KB: 67.6 + 48 + 49 > 160)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export GPX_PANEL_PAD_LDS=49152
rm -f gpurun_out/r04_leaf_steps_excl2.log gpurun_out/r04_ab_excl2.log
for at in 12 3; do
  echo "== GPX_PANEL_EXCL_ROWS=8192 PAD=49152 launch $at" >> gpurun_out/r04_leaf_steps_excl2.log
  GPX_PANEL_EXCL_ROWS=8192 timeout -k 10 120 python tools/panel_stamps.py 8192 $at 2>&1 | grep -E "leaf|steps|core|wg 1|wg 2|wg 3|^   0 |^   1 |^   2 |^   3 |^   4 |last" >> gpurun_out/r04_leaf_steps_excl2.log || exit 1
done
cat gpurun_out/r04_leaf_steps_excl2.log
timeout -k 10 500 bash tools/r3_ab.sh 8192 3 "GPX_X=1" "GPX_PANEL_EXCL_ROWS=3072" "GPX_PANEL_EXCL_ROWS=4096" "GPX_PANEL_EXCL_ROWS=5120" "GPX_PANEL_EXCL_ROWS=6144" "GPX_PANEL_EXCL_ROWS=8192" >> gpurun_out/r04_ab_excl2.log 2>&1 || exit 1
cat gpurun_out/r04_ab_excl2.log
timeout -k 10 200 bash tools/r3_trace.sh r04_timeline_n8192_excl8192 8192 GPX_PANEL_EXCL_ROWS=8192 || exit 1
