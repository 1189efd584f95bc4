#!/bin/bash
# tools/r4_run3.sh -- round 4: the rolled leaf (GPX_LEAF=3): factorisation tests, stamps, interleaved A/B of the three leaves
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "leaf or cholesky or resident or soak or gp_nd or failing_minor or golden or record or riding or two_part" > gpurun_out/r04_pytest3.log 2>&1
rc=$?
tail -5 gpurun_out/r04_pytest3.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out / was killed: no further GPU step"; exit $rc; fi
timeout -k 10 120 python tools/panel_stamps.py 8192 3 > gpurun_out/r04_panel_stamps_n8192_step3_v3.log 2>&1 || exit 1
timeout -k 10 120 python tools/panel_stamps.py 8192 25 > gpurun_out/r04_panel_stamps_n8192_step25_v3.log 2>&1 || exit 1
tail -4 gpurun_out/r04_panel_stamps_n8192_step25_v3.log
rm -f gpurun_out/r04_ab_leaf_v3.log
for n in 8192 4096 2048; do
  timeout -k 10 300 bash tools/r3_ab.sh $n 3 "GPX_LEAF=1" "GPX_LEAF=2" "GPX_LEAF=3" >> gpurun_out/r04_ab_leaf_v3.log 2>&1 || exit 1
done
cat gpurun_out/r04_ab_leaf_v3.log
exit $rc
