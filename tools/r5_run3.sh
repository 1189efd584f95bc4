#!/bin/bash
# tools/r5_run3.sh -- round 5: where the trailing kernel's slots are idle (stamps), and the panel alone on both routes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 200 python tools/gemm_stamps.py 32768 1024 3 > gpurun_out/r05_gemm_stamps_m32768_k1024.log 2>&1 || exit 1
tail -22 gpurun_out/r05_gemm_stamps_m32768_k1024.log
for e in "GPX_POTRF_TALL_ROWS=1099511627776" "GPX_POTRF_TALL_ROWS=16384"; do
  for n in 65536 32768 20480; do
    echo "$e"; env $e PANEL_LD=1024 timeout -k 10 100 python tools/panel_bench.py $n 256 512 1024 || exit 1
  done
done 2>&1 | tee gpurun_out/r05_panel_alone_tall_vs_resident.log
