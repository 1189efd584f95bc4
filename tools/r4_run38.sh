#!/bin/bash
# tools/r4_run38.sh -- roctx ranges: the test, then a marker + kernel trace of a small bench run with GPX_ROCTX=1
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p gpurun_out
timeout -k 10 200 python -m pytest tests/test_gpu_round4.py -m gpu -q -p no:cacheprovider -k roctx 2>&1 | tail -2 || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/roctx_tr
export GPX_ROCTX=1
timeout -k 10 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d /tmp/roctx_tr -o t -- python3 $ROOT/bench.py --problem-n 4096 --problem-d 8 --problem-m 256 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-prof > /tmp/roctx_tr.json 2>/tmp/roctx_tr.err || { tail -5 /tmp/roctx_tr.err; exit 1; }
ls /tmp/roctx_tr/* | head -20
f=$(find /tmp/roctx_tr -name "*marker_api_trace.csv" | head -1)
if [ -n "$f" ]; then head -1 $f; python3 - "$f" <<'PY' > $ROOT/gpurun_out/r04_roctx_marker_trace.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("rocprofv3 --marker-trace --kernel-trace -- python3 bench.py --problem-n 4096 --problem-d 8 --steps 2 --warmup 1 (GPX_ROCTX=1): %d marker records" % len(rows))
cnt = collections.Counter(r.get("Function", r.get("Name", "?")) for r in rows)
for k, v in cnt.most_common(): print("%6d  %s" % (v, k))
print("first records:")
for r in rows[:40]: print("  ", {k: r[k] for k in r if k in ("Function", "Name", "Start_Timestamp", "End_Timestamp", "Thread_Id")})
PY
cat $ROOT/gpurun_out/r04_roctx_marker_trace.txt | head -30; fi
f2=$(find /tmp/roctx_tr -name "*marker_api_stats.csv" -o -name "*marker*stats*.csv" | head -1)
if [ -n "$f2" ]; then cp $f2 $ROOT/gpurun_out/r04_roctx_marker_stats.csv; head -20 $f2; fi
exit 0
