#!/bin/bash
# tools/r4_run2.sh -- round 4, second GPU call: the whole -m gpu suite with the round-4 leaf as default, its stamps, and
# interleaved A/B of the two leaves at n = 2048 .. 12288
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/r04_golden_ratios.jsonl
GPX_GOLDEN_RATIOS=$PWD/gpurun_out/r04_golden_ratios.jsonl timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_pytest2.log 2>&1
rc=$?
tail -5 gpurun_out/r04_pytest2.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out / was killed: no further GPU step"; exit $rc; fi
timeout -k 10 120 python tools/panel_stamps.py 8192 3 > gpurun_out/r04_panel_stamps_n8192_step3_v2.log 2>&1 || exit 1
timeout -k 10 120 python tools/panel_stamps.py 8192 25 > gpurun_out/r04_panel_stamps_n8192_step25_v2.log 2>&1 || exit 1
tail -4 gpurun_out/r04_panel_stamps_n8192_step25_v2.log
for n in 8192 4096 2048 12288; do
  timeout -k 10 300 bash tools/r3_ab.sh $n 3 "GPX_LEAF_V2=0" "GPX_LEAF_V2=1" >> gpurun_out/r04_ab_leaf_v2.log 2>&1 || exit 1
done
cat gpurun_out/r04_ab_leaf_v2.log
timeout -k 10 200 bash tools/r3_trace.sh r04_timeline_n8192_leaf_v2 8192 || exit 1
exit $rc
