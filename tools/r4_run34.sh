#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 400 bash tools/r4_ab_sized.sh 16384 8 f32 2 "GPX_LEAF4_ROWS=8192" "GPX_LEAF4_ROWS=0" "GPX_LEAF4_ROWS=16384" || exit 1
timeout -k 10 600 bash tools/r4_ab_sized.sh 32768 16 f32 2 "GPX_LEAF4_ROWS=8192" "GPX_LEAF4_ROWS=0" "GPX_LEAF4_ROWS=5120" || exit 1
