#!/bin/bash
# tools/r3_ab_big.sh <reps> "ENV..." ... -- interleaved repetitions of the headline bench (2 steps): step s, potrf ms, trailing frac
cd "$(dirname "$0")/.."
reps=$1; shift
for r in $(seq $reps); do
  for e in "$@"; do
    env $e python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%-50s step %.4f s potrf %.1f ms trailing frac %.4f' % ('$e', j['value'], j['stages_ms']['potrf'], j['roofline']['frac']))"
  done
done
