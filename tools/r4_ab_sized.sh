#!/bin/bash
# tools/r4_ab_sized.sh <n> <d> <dtype> <reps> "ENV..." ... -- interleaved bench runs at any size: step s, potrf ms, trailing fractions
cd "$(dirname "$0")/.."
n=$1; d=$2; dt=$3; reps=$4; shift; shift; shift; shift
for r in $(seq $reps); do
  for e in "$@"; do
    env $e python bench.py --problem-n $n --problem-d $d --problem-m 1024 --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline'] or {}; print('n=$n $dt %-44s step %.4f s potrf %.2f ms trailing128 %.4f all %.4f' % ('$e', j['value'], j['stages_ms']['potrf'], r.get('frac', 0), (r.get('all_trailing') or {}).get('frac', 0)))"
  done
done
