#!/bin/bash
# tools/r4_run6.sh -- round 4: outer-block widths with the nested wide panel (512 / 1024 while many rows are left), timelines
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/r04_ab_widths.log
timeout -k 10 400 bash tools/r3_ab.sh 8192 3 "GPX_X=0" "GPX_POTRF_WIDTHS=1,4096,8192" "GPX_POTRF_WIDTHS=1,5120,8192" "GPX_POTRF_WIDTHS=1,6144,8192" "GPX_POTRF_WIDTHS=1,3072,8192" "GPX_POTRF_WIDTHS=1,6144,6144" >> gpurun_out/r04_ab_widths.log 2>&1 || exit 1
cat gpurun_out/r04_ab_widths.log
timeout -k 10 200 bash tools/r3_trace.sh r04_timeline_n8192_nested1024 8192 GPX_POTRF_WIDTHS=1,5120,5120 || exit 1
timeout -k 10 200 bash tools/r3_trace.sh r04_timeline_n8192_nb512 8192 GPX_POTRF_WIDTHS=1,4096,8192 || exit 1
timeout -k 10 200 bash tools/r3_trace.sh r04_timeline_n8192_leaf4 8192 || exit 1
timeout -k 10 300 python bench.py --problem-n 32768 --problem-d 16 --problem-m 1024 --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r04_b32k_f32_a.json 2>gpurun_out/r04_b32k_f32_a.err || exit 1
python -c "import json;j=json.loads(open('gpurun_out/r04_b32k_f32_a.json').read().strip().splitlines()[-1]);print('f32 32768', j['stages_ms'], j['roofline']['frac'])"
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r04_b64k_a.json 2>gpurun_out/r04_b64k_a.err || exit 1
python -c "import json;j=json.loads(open('gpurun_out/r04_b64k_a.json').read().strip().splitlines()[-1]);print('f64 65536', j['stages_ms'], j['roofline']['frac'], j['roofline'].get('all_trailing'))"
