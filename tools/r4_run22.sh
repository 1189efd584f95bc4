#!/bin/bash
# tools/r4_run22.sh -- the two-wave leaf in the two-workgroups-per-CU instantiation (GPX_LEAF=5 default / 1 = round-3 leaf): parity of
# the batch / config tests, then lock-step batches and tall single factorisations, both leaves on one box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r04_pytest22.log 2>&1
rc=$?; tail -3 gpurun_out/r04_pytest22.log
if [ $rc -ne 0 ]; then grep -n "^FAILED\|Error" gpurun_out/r04_pytest22.log | head; exit $rc; fi
for rep in 1 2; do
  for e in "GPX_LEAF=5" "GPX_LEAF=1"; do echo "== $e"; env $e timeout -k 10 200 python tools/r3_batch8.py || exit 1; done
done
for n in 12288 16384; do timeout -k 10 400 bash tools/r4_ab_sized.sh $n 8 f64 2 "GPX_LEAF=5" "GPX_LEAF=1" || exit 1; done
timeout -k 10 600 bash tools/r4_ab_sized.sh 65536 32 f64 1 "GPX_LEAF=5" "GPX_LEAF=1" || exit 1
