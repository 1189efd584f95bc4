#!/bin/bash
# tools/profile_round.sh <tag>  -- on the GPU box: the round's evidence for the headline workload.
#   1. python bench.py                                   -> bench_default.json
#   2. the same under rocprofv3 --kernel-trace --stats   -> kernel stats CSV + the JSON line it printed
#   3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of a 1-step run -> counter CSVs
# Everything lands under gpurun_out/<tag>/ ; tools/pmc_summary.py turns 3. into the traffic JSON.
set -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err || exit 1
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o bench -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-prof > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err || exit 1
  echo "pmc $c done"
done
# keep only what is small: per-kernel sums of the counter CSVs (the raw per-dispatch CSVs are large)
python3 $ROOT/tools/pmc_summary.py $OUT
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
rm -f $OUT/stats/*kernel_trace.csv
ls -la $OUT $OUT/stats
