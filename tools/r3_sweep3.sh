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
for n in 2048 4096 6144 8192 10240 12288; do
  run $n "reserve as round 2" X=1
  run $n "RESERVE_CUS=0" GPX_POTRF_RESERVE_CUS=0
  run $n "RESERVE_BELOW=0 (pipe leaf flag on)" GPX_POTRF_RESERVE_BELOW=0
done
run 8192 "WIDTHS=1,6144,12288 RESERVE_CUS=0" GPX_POTRF_WIDTHS=1,6144,12288 GPX_POTRF_RESERVE_CUS=0
run 8192 "WIDTHS=1,6144,12288 RESERVE_CUS=0 FOLD_K=512" GPX_POTRF_WIDTHS=1,6144,12288 GPX_POTRF_RESERVE_CUS=0 GPX_POTRF_FOLD_K=512
run 8192 "WIDTHS=1,5120,12288 RESERVE_CUS=0 FOLD_K=512" GPX_POTRF_WIDTHS=1,5120,12288 GPX_POTRF_RESERVE_CUS=0 GPX_POTRF_FOLD_K=512
run 8192 "RESERVE_CUS=0 ATOMIC_C=0" GPX_POTRF_RESERVE_CUS=0 GPX_GEMM_ATOMIC_C=0
run 8192 "RESERVE_CUS=0 FINE_TILES chunks of 4 (n/a)" GPX_POTRF_RESERVE_CUS=0
