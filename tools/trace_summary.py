"""tools/trace_summary.py -- per-kernel totals and a timeline excerpt from a rocprofv3 kernel-trace CSV (last bench step)."""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 40
def short(n):
    n = re.sub(r"\(.*", "", n)
    return n.replace("gpx::", "").replace("void ", "")[:60]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "kmat_kernel" in r["Kernel_Name"]]
sub = rows[idx[-1]:]
t0 = int(sub[0]["Start_Timestamp"])
agg = collections.OrderedDict()
for r in sub:
    k = short(r["Kernel_Name"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += d
for k, (n, t) in agg.items():
    print("%-62s n=%4d total %9.1f us  avg %8.2f us" % (k, n, t, t / n))
print("step span %.1f us" % ((int(sub[-1]["End_Timestamp"]) - t0) / 1e3))
for r in sub[:nshow]:
    print("%9.1f %8.1f q=%s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id"), short(r["Kernel_Name"])))
