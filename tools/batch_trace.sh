#!/bin/bash
# tools/batch_trace.sh <tag> <rows> [ENV=VAL ...] -- kernel-trace timeline of the last lock-step evaluation, to gpurun_out/<tag>.txt
tag=$1; rows=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -o t -- python3 $ROOT/tools/batch_trace.py $rows > /tmp/tr_$tag.out 2>/tmp/tr_$tag.err || { tail -5 /tmp/tr_$tag.err; exit 1; }
f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" $rows > $ROOT/gpurun_out/$tag.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
count = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(.*", "", n)
    return n.replace("gpx::", "").replace("void ", "")[:44]
idx = [i for i, r in enumerate(rows) if "kmat_kernel" in r["Kernel_Name"]]
# the last evaluation starts at the first kmat launch of the last group (launches of one evaluation follow each other closely)
start = idx[-1]
for a, b in zip(reversed(idx[:-1]), reversed(idx[1:])):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"]) > 2000000:
        break
    start = a
sub = rows[start:]
t0 = int(sub[0]["Start_Timestamp"])
prev_end = {}
for r in sub:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    q = r.get("Queue_Id")
    gap = s - prev_end.get(q, s)
    prev_end[q] = e
    k = r["Kernel_Name"]
    mark = "P" if "panel_res" in k else ("U" if "128, 1, 0>" in k else " ")
    print("%9.1f %8.1f  end %9.1f  gap %6.1f  q=%-3s grid=%-7s y=%-3s %s %s" % (s, e - s, e, gap, q, r["Grid_Size_X"], r.get("Grid_Size_Y"), mark, short(k)))
PY
tail -3 $ROOT/gpurun_out/$tag.txt
