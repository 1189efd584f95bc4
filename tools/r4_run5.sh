#!/bin/bash
# tools/r4_run5.sh -- round 4: one-wave leaf (GPX_LEAF=4) and nested wide panels: tests, per-step stamps, A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "leaves or nested or cholesky or resident or failing_minor or gp_nd or record or riding or two_part or outer_block" > gpurun_out/r04_pytest5.log 2>&1
rc=$?
tail -5 gpurun_out/r04_pytest5.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out / was killed: no further GPU step"; exit $rc; fi
rm -f gpurun_out/r04_leaf_steps_v4.log
for at in 25 3; do
  echo "== GPX_LEAF=4 launch $at" >> gpurun_out/r04_leaf_steps_v4.log
  GPX_LEAF=4 timeout -k 10 120 python tools/panel_stamps.py 8192 $at 2>&1 | grep -E "leaf|steps|core|wg 1|^   0 |^   1 |^   2 |^   3 |last" >> gpurun_out/r04_leaf_steps_v4.log || exit 1
done
cat gpurun_out/r04_leaf_steps_v4.log
rm -f gpurun_out/r04_ab_leaf_v4.log
for n in 8192 4096 2048; do
  timeout -k 10 300 bash tools/r3_ab.sh $n 3 "GPX_LEAF=1" "GPX_LEAF=4" "GPX_LEAF=4 GPX_POTRF_WIDTHS=1,5120,5120" "GPX_LEAF=4 GPX_POTRF_WIDTHS=1,4096,4096" "GPX_LEAF=4 GPX_POTRF_WIDTHS=1,3072,3072" "GPX_LEAF=1 GPX_POTRF_WIDTHS=1,5120,5120" >> gpurun_out/r04_ab_leaf_v4.log 2>&1 || exit 1
done
cat gpurun_out/r04_ab_leaf_v4.log
exit $rc
