#!/bin/bash
# tools/r4_run1.sh -- round 4, first GPU call: the whole -m gpu suite (with the golden-record error ratios dumped), then the
# n = 8192 baseline: kernel-trace timeline, panel stamps of an update-bound and a chain-bound step, a quick bench line.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/r04_golden_ratios.jsonl
GPX_GOLDEN_RATIOS=$PWD/gpurun_out/r04_golden_ratios.jsonl timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_pytest1.log 2>&1
rc=$?
tail -5 gpurun_out/r04_pytest1.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out / was killed: no further GPU step"; exit $rc; fi
timeout -k 10 120 python tools/panel_stamps.py 8192 3 > gpurun_out/r04_panel_stamps_n8192_step3_before.log 2>&1 || exit 1
timeout -k 10 120 python tools/panel_stamps.py 8192 25 > gpurun_out/r04_panel_stamps_n8192_step25_before.log 2>&1 || exit 1
timeout -k 10 200 bash tools/r3_trace.sh r04_timeline_n8192_before 8192 || exit 1
timeout -k 10 120 python bench.py --problem-n 8192 --problem-d 8 --problem-m 1024 --steps 12 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r04_b8k_before.json 2>gpurun_out/r04_b8k_before.err || exit 1
tail -c 600 gpurun_out/r04_b8k_before.json
exit $rc
