#!/bin/bash
# tools/r6_soak_probe.py under the switches that change what runs beside what (one process per variant, one after the other)
out=gpurun_out/r06_soak_variants.log
: > $out
run() { echo "=== $*" >> $out; env "$@" timeout -k 10 170 python tools/r6_soak_probe.py n=8192 dtype=float32 reps=${REPS:-4000} $EXTRA >> $out 2>&1 || echo "(exit $?)" >> $out; }
run GPX_NONE=1
EXTRA="neighbour=0" run GPX_NONE=1
run GPX_POTRF_NO_LOOKAHEAD=1
run GPX_GEMM_NO_FAST=1
run GPX_FIT_OPS_AHEAD=0
run GPX_POTRF_FOLD_ROWS=0
[ -f tools/ab/libgpx_r05.so ] && EXTRA="lib=tools/ab/libgpx_r05.so" run GPX_NONE=1
grep -E "^===|fits differed|first differing column|^   row" $out
