#!/bin/bash
# tools/r5_trace.sh <tag> <n> <d> <dtype> [ENV=VAL ...] -- kernel-trace timeline of the last fit of a short bench run (rocprofv3), to gpurun_out/<tag>.txt
tag=$1; n=$2; d=$3; dt=$4; shift; shift; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -o t -- python3 $ROOT/bench.py --problem-n $n --problem-d $d --problem-m 1024 --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-prof > /tmp/tr_$tag.json 2>/tmp/tr_$tag.err
f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/fit_timeline.py $f > $ROOT/gpurun_out/$tag.txt
python3 -c "import json;j=json.loads(open('/tmp/tr_$tag.json').read().strip().splitlines()[-1]);print('$tag', j['stages_ms'])"
