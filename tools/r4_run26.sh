#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for e in "GPX_X=0" "GPX_LEAF4_ROWS=0"; do
  echo "== $e"; env $e timeout -k 10 500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|ACTUAL|DESIRED|^FAILED"
done
