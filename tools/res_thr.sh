#!/bin/bash
n=$1; shift
for thr in "$@"; do
  GPX_POTRF_RESERVE_BELOW=$thr python bench.py --problem-n $n --problem-d 8 --problem-m 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-prof > /tmp/sw.json 2>/tmp/sw.err || { tail -3 /tmp/sw.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("/tmp/sw.json").read().strip().splitlines()[-1])
print("n=$n reserve_below=$thr potrf %.3f ms fit %.3f" % (j["stages_ms"]["potrf"], j["stages_ms"]["fit_total"]))
PY
done
