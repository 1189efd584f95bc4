#!/bin/bash
# tools/r5_run7.sh -- round 5: rehearsal sweep at N = 65536 fp64: one rank's share for P = 8 / 4 / 2 over width x broadcast form
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/r05_mg_rehearsal_sweep.jsonl
for cfg in "8 256 0" "8 256 1" "8 512 0" "8 512 1" "8 1024 0" "8 1024 1" "4 256 1" "4 512 0" "4 512 1" "4 1024 0" "4 1024 1" "2 512 0" "2 1024 0"; do
  set -- $cfg
  timeout -k 10 200 python tools/mg_rehearse.py 65536 32 $1 0 $2 4 $3 >> gpurun_out/r05_mg_rehearsal_sweep.jsonl 2> gpurun_out/r05_mg_rehearsal_sweep.err || { tail -5 gpurun_out/r05_mg_rehearsal_sweep.err; exit 1; }
  tail -1 gpurun_out/r05_mg_rehearsal_sweep.jsonl | python -c "
import json,sys
j=json.loads(sys.stdin.read()); p=j['per_step_ms']
print('P=$1 nb=$2 sag=$3 step %.4f s  update %.2f wait %.2f transfer %.2f remote-chain %.2f own-chain %.2f ms/step' % (j['rank_step_s'], p['update'], p['exposed_wait'], p['modelled_transfer'], p['modelled_remote_chain'], p['own_chain_per_owned_panel_mean']))"
done
