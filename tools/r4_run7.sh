#!/bin/bash
# tools/r4_run7.sh -- round 4: where the nested panel and the one-wave leaf matter beyond n = 8192: lock-step batches, n = 16384,
# N = 32768 fp32, N = 65536 (same box, interleaved)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/r04_ab_batch_nested.log gpurun_out/r04_ab_big.log
for e in "GPX_POTRF_NESTED=1" "GPX_POTRF_NESTED=0" "GPX_POTRF_NESTED=1 GPX_LEAF=4" "GPX_POTRF_NESTED=1" "GPX_POTRF_NESTED=0"; do
  echo "== $e" >> gpurun_out/r04_ab_batch_nested.log
  env $e timeout -k 10 200 python tools/r3_batch8.py >> gpurun_out/r04_ab_batch_nested.log 2>&1 || exit 1
done
cat gpurun_out/r04_ab_batch_nested.log
timeout -k 10 400 bash tools/r3_ab.sh 16384 3 "GPX_POTRF_NESTED=1" "GPX_POTRF_NESTED=0" "GPX_LEAF=1" >> gpurun_out/r04_ab_big.log 2>&1 || exit 1
timeout -k 10 500 bash tools/r4_ab_sized.sh 32768 16 f32 2 "GPX_POTRF_NESTED=1" "GPX_POTRF_NESTED=0" >> gpurun_out/r04_ab_big.log 2>&1 || exit 1
timeout -k 10 700 bash tools/r4_ab_sized.sh 65536 32 f64 2 "GPX_X=1" "GPX_POTRF_NESTED=0 GPX_LEAF=1" "GPX_LEAF=1" >> gpurun_out/r04_ab_big.log 2>&1 || exit 1
cat gpurun_out/r04_ab_big.log
