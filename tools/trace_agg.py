"""tools/trace_agg.py -- per-kernel totals of the last N dispatches of a rocprofv3 kernel-trace CSV"""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
rows = rows[-last:]
def short(n):
    n = re.sub(r"\(.*", "", n)
    return n.replace("gpx::", "").replace("void ", "")[:60]
agg = collections.OrderedDict()
for r in rows:
    k = short(r["Kernel_Name"]) + " grid=" + r["Grid_Size_X"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-80s n=%4d total %9.1f us  avg %8.2f us" % (k, n, t, t / n))
print("span %.1f us" % ((int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3))
