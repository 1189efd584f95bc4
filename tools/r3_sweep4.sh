#!/bin/bash
cd "$(dirname "$0")/.."
run() {  # n label env...
  n=$1; label=$2; shift; shift
  env "$@" python bench.py --problem-n $n --problem-d 8 --problem-m 1024 --dtype f64 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-prof > /tmp/sw.json 2>/tmp/sw.err || { echo "$label FAILED"; tail -3 /tmp/sw.err; return; }
  python - "$n $label" <<'PY'
import json, sys
j=json.loads(open("/tmp/sw.json").read().strip().splitlines()[-1])
print("n=%-52s potrf %.3f ms fit %.3f resid %.1e" % (sys.argv[1], j["stages_ms"]["potrf"], j["stages_ms"]["fit_total"], j["check"]["max_abs_residual_K_alpha_minus_y_over_max_y"]))
PY
}
run 8192 "split off" GPX_POTRF_SPLIT_CUS=0
for c in 8 16; do for a in 1024 2048 3072 4096; do run 8192 "SPLIT_CUS=$c ABOVE=$a" GPX_POTRF_SPLIT_CUS=$c GPX_POTRF_SPLIT_ABOVE=$a; done; done
run 8192 "SPLIT_CUS=8 ABOVE=2048 ATOMIC_C=0" GPX_POTRF_SPLIT_CUS=8 GPX_GEMM_ATOMIC_C=0
for n in 4096 6144 12288; do run $n "split off" GPX_POTRF_SPLIT_CUS=0; run $n "split default" X=1; done
