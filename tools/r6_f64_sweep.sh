#!/bin/bash
# the headline (N = 65536, d = 32, fp64) under a few schedule switches on the final build: step time and the trailing kernel's fraction
out=gpurun_out/r06_f64_sweep.log
: > $out
run() { printf "%-44s " "$*" >> $out; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.2f ms  trailing %.4f (%d launches, %.3f ms)  potrf %.2f ms' % (d['ms_per_step'], r['frac'], r['launches_per_step'], r['avg_launch_ms'], d['stages_ms']['potrf']))" >> $out 2>&1; tail -1 $out; }
run GPX_NONE=1
run GPX_NONE=1
run GPX_POTRF_TWO_PART_ROWS=8192
run GPX_POTRF_TWO_PART_ROWS=32768
run GPX_SYRK_BN64_TILES=1500
run GPX_SYRK_BN64_TILES=5000
run GPX_GEMM_FINE_TILES=8192
run GPX_GEMM_FINE_TILES=32768
run GPX_POTRF_WIDTHS=8192,16384,32768
run GPX_POTRF_WIDTHS=4096,8192,12288
run GPX_POTRF_GATE_ROWS=0
run GPX_NONE=1
