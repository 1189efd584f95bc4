#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/r04_ab_batch_gate.log
for e in "GPX_X=1" "GPX_POTRF_GATE_BATCH=1" "GPX_X=1" "GPX_POTRF_GATE_BATCH=1"; do
  echo "== $e" >> gpurun_out/r04_ab_batch_gate.log
  env $e timeout -k 10 200 python tools/r3_batch8.py >> gpurun_out/r04_ab_batch_gate.log 2>&1 || exit 1
done
cat gpurun_out/r04_ab_batch_gate.log
timeout -k 10 400 bash tools/r4_ab_sized.sh 32768 16 f32 2 "GPX_X=1" "GPX_POTRF_GATE_ROWS=0" "GPX_POTRF_GATE_ROWS=65536" > gpurun_out/r04_ab_gate_big.log 2>&1 || exit 1
timeout -k 10 400 bash tools/r3_ab.sh 16384 2 "GPX_X=1" "GPX_POTRF_GATE_ROWS=0" >> gpurun_out/r04_ab_gate_big.log 2>&1 || exit 1
timeout -k 10 600 bash tools/r4_ab_sized.sh 65536 32 f64 2 "GPX_X=1" "GPX_POTRF_GATE_ROWS=0" "GPX_POTRF_GATE_ROWS=65536" >> gpurun_out/r04_ab_gate_big.log 2>&1 || exit 1
cat gpurun_out/r04_ab_gate_big.log
