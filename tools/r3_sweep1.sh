#!/bin/bash
# round 3, sweep 1: n = 8192 fp64 -- outer-block taper thresholds x CU-reservation threshold
cd "$(dirname "$0")/.."
run() {  # label, env assignments...
  label=$1; shift
  env "$@" python bench.py --problem-n 8192 --problem-d 8 --problem-m 1024 --dtype f64 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-prof > /tmp/sw.json 2>/tmp/sw.err || { echo "$label FAILED"; tail -3 /tmp/sw.err; return; }
  python - "$label" <<'PY'
import json, sys
j=json.loads(open("/tmp/sw.json").read().strip().splitlines()[-1])
print("%-60s potrf %.3f ms fit %.3f resid %.1e" % (sys.argv[1], j["stages_ms"]["potrf"], j["stages_ms"]["fit_total"], j["check"]["max_abs_residual_K_alpha_minus_y_over_max_y"]))
PY
}
run "default" X=1
run "default(again)" X=1
for rb in 6144 5120 4096 3072 2048; do run "RESERVE_BELOW=$rb" GPX_POTRF_RESERVE_BELOW=$rb; done
for w in 1,6144,12288 1,5120,12288 1,4096,12288 1,3072,12288; do
  run "WIDTHS=$w" GPX_POTRF_WIDTHS=$w
  for rb in 5120 4096 3072; do run "WIDTHS=$w RESERVE_BELOW=$rb" GPX_POTRF_WIDTHS=$w GPX_POTRF_RESERVE_BELOW=$rb; done
done
run "RESERVE_CUS=16" GPX_POTRF_RESERVE_CUS=16
run "RESERVE_CUS=48" GPX_POTRF_RESERVE_CUS=48
run "RESERVE_CUS=64" GPX_POTRF_RESERVE_CUS=64
