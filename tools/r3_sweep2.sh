#!/bin/bash
# round 3, sweep 2: chunk granularity of the exact tile map x CU reservation threshold, n = 4096 .. 16384
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
for n in 8192 4096 12288 16384; do
  run $n "old chunks (64)" GPX_GEMM_FINE_TILES=0
  run $n "fine chunks (8)" X=1
done
for rb in 8192 6144 5120 4096 3072 2048 0; do run 8192 "fine RESERVE_BELOW=$rb" GPX_POTRF_RESERVE_BELOW=$rb; done
for rc in 16 24 40; do run 8192 "fine RESERVE_CUS=$rc" GPX_POTRF_RESERVE_CUS=$rc; done
