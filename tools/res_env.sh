#!/bin/bash
# n dtype VAR then values of that environment variable
n=$1; dt=$2; var=$3; shift; shift; shift
for v in "$@"; do
  env $var=$v python bench.py --problem-n $n --problem-d 8 --problem-m 1024 --dtype $dt --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-prof > /tmp/sw.json 2>/tmp/sw.err || { tail -3 /tmp/sw.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("/tmp/sw.json").read().strip().splitlines()[-1])
print("n=$n $dt $var=$v potrf %.3f ms fit %.3f resid %.1e" % (j["stages_ms"]["potrf"], j["stages_ms"]["fit_total"], j["check"]["max_abs_residual_K_alpha_minus_y_over_max_y"]))
PY
done
