#!/bin/bash
# tools/r6_soak_probe.py over sizes / dtypes / routes (one process each): every fit beside a neighbour must equal the first bit for bit
out=gpurun_out/r06_soak_sweep.log
: > $out
run() { echo "=== $*" >> $out; timeout -k 10 ${TMO:-150} python tools/r6_soak_probe.py "$@" >> $out 2>&1 || echo "(exit $?)" >> $out; tail -1 $out; }
run n=1990 dtype=float32 reps=4000
run n=1990 dtype=float64 reps=4000
run n=4171 dtype=float32 reps=3000
run n=4171 dtype=float64 reps=3000
run n=8191 dtype=float32 reps=3000
run n=8192 dtype=float64 reps=6000
run n=12288 dtype=float32 reps=1500
run n=12288 dtype=float64 reps=1000
run n=16384 dtype=float32 reps=1000 neighbour=5000
run n=16384 dtype=float64 reps=600 neighbour=5000
run n=20000 dtype=float32 reps=500 neighbour=8192
run n=20000 dtype=float64 reps=300 neighbour=8192
grep -E "^===|fits differed|first differing|^   row" $out > gpurun_out/r06_soak_sweep_summary.log
