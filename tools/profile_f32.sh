#!/bin/bash
# tools/profile_f32.sh <tag> -- config 3 (N = 32768, d = 16, fp32) under rocprofv3 --kernel-trace --stats: bench line + per-kernel stats
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${TAG}_f32
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --problem-n 32768 --problem-d 16 --problem-m 1024 --dtype f32 --no-cpu-baseline --no-secondary > $OUT/bench_n32768_f32.json 2> $OUT/err.txt || { tail -3 $OUT/err.txt; exit 1; }
rm -f $OUT/stats/*kernel_trace.csv
tail -c 400 $OUT/bench_n32768_f32.json
