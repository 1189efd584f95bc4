#!/bin/bash
# sweep of outer block / reserved CUs with the resident panel kernel: n dtype "nb list" "reserve list"
n=$1; dt=$2; nbs=$3; rsv=$4
for nb in $nbs; do for r in $rsv; do
  GPX_POTRF_NB=$nb GPX_POTRF_RESERVE_CUS=$r python bench.py --problem-n $n --problem-d 8 --problem-m 1024 --dtype $dt --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-prof > /tmp/sw.json 2>/tmp/sw.err || { tail -3 /tmp/sw.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("/tmp/sw.json").read().strip().splitlines()[-1])
print("n=$n $dt nb=$nb reserve=$r potrf %.3f ms fit %.3f" % (j["stages_ms"]["potrf"], j["stages_ms"]["fit_total"]))
PY
done; done
