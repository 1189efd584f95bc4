#!/bin/bash
# tools/r4_run17.sh -- A/B on ONE box: the library before / after the rescheduled leaf (tools/ab/*.so, built by hand), potrf at
# n = 2048 / 4096 / 8192 and the panel stamps of launch 25 at n = 8192
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
for v in before after; do
  cp tools/ab/libgpx_$v.so gaussian_processes_amd/libgpx.so
  for n in 2048 4096 8192; do timeout -k 10 200 bash tools/r4_ab_sized.sh $n 8 f64 1 "GPX_V=$v" || exit 1; done
done
done
for v in before after; do
  cp tools/ab/libgpx_$v.so gaussian_processes_amd/libgpx.so
  echo "== $v"; timeout -k 10 120 python tools/panel_stamps.py 8192 25 2>&1 | tail -22 || exit 1
done
cp tools/ab/libgpx_after.so gaussian_processes_amd/libgpx.so
