#!/bin/bash
# A/B of the resident panel kernel (GPX_POTRF_RES=0 restores the launch chain) over the small and mid sizes
out=${1:-gpurun_out/res_ab}
mkdir -p $out
for n in 4096 8192 16384; do
  for res in 0 256; do
    GPX_POTRF_RES=$res python bench.py --problem-n $n --problem-d 8 --problem-m 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-prof > $out/n${n}_res${res}.json 2>$out/n${n}_res${res}.err || exit 1
    python - <<PY
import json
j=json.loads(open("$out/n${n}_res${res}.json").read().strip().splitlines()[-1])
print("n=$n res=$res", j["value"], j["stages_ms"], j["check"]["max_abs_residual_K_alpha_minus_y_over_max_y"])
PY
  done
done
