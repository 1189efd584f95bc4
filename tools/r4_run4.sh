#!/bin/bash
# tools/r4_run4.sh -- per-step stamps of the three leaves (chain-bound step 25, update-bound step 3), n = 8192
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/r04_leaf_steps.log
for leaf in 1 2 3; do
  for at in 25 3; do
    echo "== GPX_LEAF=$leaf launch $at" >> gpurun_out/r04_leaf_steps.log
    GPX_LEAF=$leaf timeout -k 10 120 python tools/panel_stamps.py 8192 $at 2>&1 | grep -E "leaf|steps|core|wg 1|^   0 |^   1 " >> gpurun_out/r04_leaf_steps.log || exit 1
  done
done
cat gpurun_out/r04_leaf_steps.log
