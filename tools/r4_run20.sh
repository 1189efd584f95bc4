#!/bin/bash
# tools/r4_run20.sh -- 128-wide panels for the chain-bound tail, with the two-wave leaf: rows128 threshold sweep
cd "$(dirname "$0")/.."
for n in 2048 4096 8192; do
  timeout -k 10 400 bash tools/r4_ab_sized.sh $n 8 f64 2 "GPX_POTRF_WIDTHS=1,8192,12288" "GPX_POTRF_WIDTHS=1024,8192,12288" "GPX_POTRF_WIDTHS=2048,8192,12288" "GPX_POTRF_WIDTHS=4096,8192,12288" || exit 1
done
