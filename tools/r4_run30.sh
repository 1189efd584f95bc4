#!/bin/bash
# tools/r4_run30.sh -- with the faster leaf: is the folded pre-update still the better form?  (GPX_POTRF_FOLD_ROWS = rows up to which it is used)
cd "$(dirname "$0")/.."
for n in 2048 4096 8192; do
  timeout -k 10 400 bash tools/r4_ab_sized.sh $n 8 f64 2 "GPX_POTRF_FOLD_ROWS=16384" "GPX_POTRF_FOLD_ROWS=0" "GPX_POTRF_FOLD_ROWS=2048" "GPX_POTRF_FOLD_K=128" || exit 1
done
