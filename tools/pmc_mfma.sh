#!/bin/bash
# tools/pmc_mfma.sh <tag> -- on the GPU box: MFMA utilisation of the headline fit's kernels.
# One --pmc pass (SQ counters + GRBM_GUI_ACTIVE, separate from any trace domain but --kernel-trace) of a 1-step run;
# tools/pmc_mfma_summary.py reduces the per-dispatch CSV to per-kernel sums and the MFMA-busy fraction.
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${TAG}_mfma
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/pmc -o bench -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-prof \
  > $OUT/bench.json 2> $OUT/pmc.err || { tail -5 $OUT/pmc.err; exit 1; }
python3 $ROOT/tools/pmc_mfma_summary.py $OUT || exit 1
rm -rf $OUT/pmc
ls -la $OUT
