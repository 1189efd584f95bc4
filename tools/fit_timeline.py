"""tools/fit_timeline.py -- every dispatch of the LAST fit in a rocprofv3 kernel-trace CSV: start, duration, queue, grid
(diagnostic; one line per launch, panel launches marked)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(.*", "", n)
    return n.replace("gpx::", "").replace("void ", "")[:44]
idx = [i for i, r in enumerate(rows) if "kmat_kernel" in r["Kernel_Name"]]
sub = rows[idx[-2]:idx[-1]] if len(idx) > 1 else rows[idx[-1]:]
t0 = int(sub[0]["Start_Timestamp"])
prev_end = {}
for r in sub:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    q = r.get("Queue_Id")
    gap = s - prev_end.get(q, s)
    prev_end[q] = e
    mark = "P" if "panel_res" in r["Kernel_Name"] else ("U" if "128, 1, 0>" in r["Kernel_Name"] or "128,1,0>" in r["Kernel_Name"].replace(" ", "") else " ")
    print("%9.1f %8.1f  end %9.1f  gap %6.1f  q=%-3s grid=%-6s %s %s" % (s, e - s, e, gap, q, r["Grid_Size_X"], mark, short(r["Kernel_Name"])))
