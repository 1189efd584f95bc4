#!/bin/bash
# tools/r5_run8.sh -- round 5: rehearsal, owner-first on / off (P = 8 / 4 / 2, nb = 512 / 1024, both broadcast forms)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/r05_mg_rehearsal_owner_first.jsonl
for cfg in "8 512 1 0" "8 512 1 1" "8 512 0 0" "8 512 0 1" "8 256 1 1" "8 1024 1 1" "4 512 1 0" "4 512 1 1" "4 1024 1 1" "2 1024 0 0" "2 1024 0 1" "2 512 0 1"; do
  set -- $cfg
  timeout -k 10 200 python tools/mg_rehearse.py 65536 32 $1 0 $2 4 $3 100 $4 >> gpurun_out/r05_mg_rehearsal_owner_first.jsonl 2> gpurun_out/r05_mg_rehearsal_owner_first.err || { tail -5 gpurun_out/r05_mg_rehearsal_owner_first.err; exit 1; }
  tail -1 gpurun_out/r05_mg_rehearsal_owner_first.jsonl | python -c "
import json,sys
j=json.loads(sys.stdin.read()); p=j['per_step_ms']
print('P=$1 nb=$2 sag=$3 owner_first=$4 step %.4f s  update %.2f wait %.2f transfer %.2f remote-chain %.2f own-chain %.2f ms/step' % (j['rank_step_s'], p['update'], p['exposed_wait'], p['modelled_transfer'], p['modelled_remote_chain'], p['own_chain_per_owned_panel_mean']))"
done
