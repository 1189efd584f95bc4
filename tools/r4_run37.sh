#!/bin/bash
# tools/r4_run37.sh -- 512-wide outer blocks for the update-bound phase of n = 8192, again, with the faster leaf
cd "$(dirname "$0")/.."
timeout -k 10 600 bash tools/r4_ab_sized.sh 8192 8 f64 2 "GPX_X=0" "GPX_POTRF_WIDTHS=1,5120,12288" "GPX_POTRF_WIDTHS=1,6144,12288" "GPX_POTRF_WIDTHS=1,5120,12288 GPX_POTRF_NESTED=1" "GPX_POTRF_WIDTHS=1,4096,12288 GPX_POTRF_NESTED=1" || exit 1
