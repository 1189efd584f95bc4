#!/bin/bash
cd "$(dirname "$0")/.."
T=tests/test_gpu_configs.py::test_two_async_fits_on_two_streams_of_one_thread_do_not_share_the_panel_scratch
for e in "GPX_X=0" "GPX_LEAF4_ROWS=0" "GPX_LEAF=1" "GPX_X=0" "GPX_LEAF4_ROWS=0" "GPX_RES_STRICT=1"; do
  echo "== $e"; env $e timeout -k 10 120 python -m pytest $T -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|ACTUAL|DESIRED" 
done
