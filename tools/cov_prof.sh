#!/bin/bash
# tools/cov_prof.sh [N d m] -- rocprofv3 kernel stats of fit + mean + 2 x cov through the public API (diagnostic)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/cp_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_stats -o g -- python3 $R/tools/cov_bench.py ${1:-8192} ${2:-8} ${3:-1024} > /tmp/cp.out 2>/tmp/cp.err
cat /tmp/cp.out
f=$(find /tmp/cp_stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-60.60s calls %5s total %9.1f us avg %8.1f us" % (r["Name"].replace("void gpx::", ""), r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
