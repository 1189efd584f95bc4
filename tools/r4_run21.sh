#!/bin/bash
# tools/r4_run21.sh -- the two-wave leaf for panels that share their CUs with the update (GPX_LEAF=4 everywhere)
cd "$(dirname "$0")/.."
for n in 8192 12288 16384; do
  timeout -k 10 400 bash tools/r4_ab_sized.sh $n 8 f64 2 "GPX_X=0" "GPX_LEAF=4" "GPX_LEAF=4 GPX_PANEL_EXCL_ROWS=0" || exit 1
done
