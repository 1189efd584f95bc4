#!/bin/bash
# tools/r5_run6.sh -- round 5: multi-GPU: rehearsal mode, schedule autotune in the 3- / 4-rank bench lines, the thread-world / native-world tests; then rehearsals at N = 65536
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_configs.py tests/test_gpu_round4.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "rehearsal or bench or world or native or mg or rccl or config4 or panel_broadcast" > gpurun_out/r05_pytest6.log 2>&1
rc=$?
tail -15 gpurun_out/r05_pytest6.log
if [ $rc -ne 0 ]; then exit $rc; fi
for P in 8 4 2; do
  timeout -k 10 300 python tools/mg_rehearse.py 65536 32 $P 0,$((P-1)) > gpurun_out/r05_mg_rehearsal_p$P.jsonl 2> gpurun_out/r05_mg_rehearsal_p$P.err || { tail -5 gpurun_out/r05_mg_rehearsal_p$P.err; exit 1; }
  python - <<PY
import json
for ln in open("gpurun_out/r05_mg_rehearsal_p$P.jsonl"):
    j = json.loads(ln)
    print("P=$P", j["what"], "nb", j["nb"], "step %.4f s" % j["rank_step_s"], j["per_step_ms"], j["check"])
PY
done
