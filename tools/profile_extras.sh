#!/bin/bash
# tools/profile_extras.sh <tag> -- the round's other evidence (after tools/profile_round.sh): config 3 under rocprofv3,
# config 2 with event profiling, config 5, panel stamps, leaf stamps, standalone rates, multi-GPU rehearsals.
set -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f32 -o bench -- python3 $ROOT/bench.py --problem-n 32768 --problem-d 16 --dtype f32 --no-cpu-baseline --no-secondary > $OUT/bench_n32768_f32.json 2> $OUT/f32.err || exit 1
rm -f $OUT/stats_f32/*kernel_trace.csv
echo "f32 done"
python3 $ROOT/bench.py --problem-n 8192 --problem-d 8 --problem-m 1024 --no-cpu-baseline --no-secondary > $OUT/bench_n8192.json 2> $OUT/n8192.err || exit 1
python3 $ROOT/bench.py --problem-n 8192 --problem-d 8 --problem-m 1024 --no-cpu-baseline --no-secondary --no-prof --steps 10 --warmup 3 > $OUT/bench_n8192_noprof.json 2>> $OUT/n8192.err || exit 1
echo "n8192 done"
cd $ROOT
python3 tools/mlii_bench.py > $OUT/mlii_bench.log 2>&1 || exit 1
echo "mlii done"
( python3 tools/panel_stamps.py 8192 24; python3 tools/panel_stamps.py 8192 4 ) > $OUT/panel_stamps.log 2>&1 || exit 1
python3 tools/syrk_bench.py > $OUT/syrk_bench_standalone.log 2>&1 || exit 1
( for r in 0 256; do echo "GPX_POTRF_RES=$r"; GPX_POTRF_RES=$r python3 tools/panel_bench.py 8192 64 128 256; GPX_POTRF_RES=$r python3 tools/panel_bench.py 65536 256 512 1024; done ) > $OUT/panel_bench.log 2>&1 || exit 1
tools/res_ab.sh $OUT/res_ab > $OUT/res_ab.log 2>&1 || exit 1
echo "standalone done"
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 GPX_BENCH_FORCE_DIST=1 GPX_FORCE_COLLECTIVES=1 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_rccl_world1_rehearsal.json 2> $OUT/mg1.err || exit 1
GPX_BENCH_SINGLE_DEVICE=1 GPX_DIST_BACKEND=gloo python3 bench.py --gpus 2 --problem-n 16384 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_2rank_gloo_rehearsal_n16384.json 2> $OUT/mg2.err || exit 1
echo "rehearsals done"
ls -la $OUT
