#!/bin/bash
# tools/r3_ab.sh <n> <reps> "ENV..." "ENV..." ...  -- interleaved repetitions of several environments (ms of stage $STAGE, default potrf, each run, then the medians)
cd "$(dirname "$0")/.."
n=$1; reps=$2; shift; shift
declare -a res
for r in $(seq $reps); do
  i=0
  for e in "$@"; do
    v=$(env $e python bench.py --problem-n $n --problem-d 8 --problem-m 1024 --dtype ${DT:-f64} --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --no-prof 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f' % j['stages_ms']['${STAGE:-potrf}'])")
    res[$i]="${res[$i]} $v"
    i=$((i+1))
  done
done
i=0
for e in "$@"; do
  python - "$n" "$e" ${res[$i]} <<'PY'
import sys, statistics
v=[float(x) for x in sys.argv[3:]]
print("n=%s %-70s median %.3f  (%s)" % (sys.argv[1], sys.argv[2], statistics.median(v), " ".join("%.3f" % x for x in v)))
PY
  i=$((i+1))
done
