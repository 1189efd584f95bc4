#!/bin/bash
# config 3 (N = 32768, d = 16, fp32) under a few schedule switches: step time and the trailing kernel's fraction
out=gpurun_out/r06_f32_sweep.log
: > $out
run() { printf "%-44s " "$*" >> $out; env "$@" python bench.py --problem-n 32768 --problem-d 16 --problem-m 1024 --dtype f32 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.2f ms  trailing %.4f (%d launches)  potrf %.2f ms' % (d['ms_per_step'], r['frac'], r['launches_per_step'], d['stages_ms']['potrf']))" >> $out 2>&1; tail -1 $out; }
run GPX_NONE=1
run GPX_NONE=1
run GPX_POTRF_TWO_PART_ROWS=8192
run GPX_POTRF_TWO_PART_ROWS=12288
run GPX_POTRF_TWO_PART_ROWS=24576
run GPX_POTRF_TWO_PART_ROWS=65536
run GPX_LEAF_MFMA_F32_ROWS=8192
run GPX_LEAF_MFMA_F32_ROWS=32768
run GPX_POTRF_NB=2048
run GPX_POTRF_NB=512
run GPX_POTRF_PAIR_ROWS=8192
run GPX_SYRK_BN64_TILES=2000
run GPX_SYRK_BN64_TILES=8000
run GPX_POTRF_GATE_ROWS=32768
run GPX_NONE=1
