#!/bin/bash
# config 2 (N = 8192, d = 8, fp64) and N = 4096 under the panel / pacing switches on the final build: potrf time
out=gpurun_out/r06_n8192_sweep.log
: > $out
one() { python bench.py --problem-n $1 --problem-d 8 --problem-m 1024 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-prof 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d['stages_ms']['potrf'], end='')"; }
run() { printf "%-36s n=8192 potrf " "$*" >> $out; env "$@" bash -c "$(declare -f one); one 8192" >> $out; printf " ms   n=4096 " >> $out; env "$@" bash -c "$(declare -f one); one 4096" >> $out; echo " ms" >> $out; tail -1 $out; }
run GPX_NONE=1
run GPX_NONE=1
run GPX_PANEL_EXCL_ROWS=8192
run GPX_PANEL_EXCL_ROWS=3072
run GPX_PANEL_EXCL_ROWS=0
run GPX_LEAF4_ROWS=4096
run GPX_LEAF4_ROWS=16384
run GPX_POTRF_FOLD_K=128
run GPX_POTRF_FOLD_K=512
run GPX_POTRF_FOLD_ROWS=4096
run GPX_POTRF_HOST_PACED=0
run GPX_POTRF_GATE_ROWS=0
run GPX_FIT_OPS_AHEAD=0
run GPX_RES_STRICT=0
run GPX_NONE=1
