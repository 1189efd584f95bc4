#!/bin/bash
# tools/r4_run39.sh -- 8 x 8192 lock-step: two-part panels and outer widths again, with the two-wave leaf
cd "$(dirname "$0")/.."
for rep in 1 2; do
for e in "GPX_X=0" "GPX_POTRF_TWO_PART_BATCH=32768" "GPX_POTRF_TWO_PART_BATCH=49152" "GPX_POTRF_NB=512" "GPX_POTRF_NB=768"; do
  echo "== $e"; env $e timeout -k 10 200 python tools/r3_batch8.py 2>&1 | head -2 || exit 1
done
done
