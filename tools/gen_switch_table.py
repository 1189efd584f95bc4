"""tools/gen_switch_table.py -- rewrite DESIGN.md section 6a's switch table from csrc/gpx_tune.h (the single source)."""
import os, re
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = open(os.path.join(root, "gaussian_processes_amd", "csrc", "gpx_tune.h")).read()
rows = []
for m in re.finditer(r'^\s*(X|XF|XS|X2)\((\w+), "(GPX_\w+)"(?:, ([^\\]*?))?\)\s*\\?$', t, re.M):
    kind, f, name, d = m.groups()
    if kind == "XF":
        dflt = "off (flag)"
    elif kind == "X2":
        a, b = [x.strip() for x in d.rsplit(",", 1)]
        dflt = "%s (fp64) / %s (fp32)" % (a, b)
    elif kind == "XS":
        dflt = "unset"
    else:
        dflt = d.strip()
    rows.append((name, dflt.replace("(int64_t)", "")))
rows += [("GPX_POTRF_WIDTHS", "1,8192,12288"), ("GPX_MG_BCAST", "one collective per chunk (`sag`: scatter + all-gather)"),
         ("GPX_RCCL_LIB", "librccl.so.1")]
txt = "**The table** (round 5; `csrc/gpx_tune.h` is the single source: name = default).  " + "; ".join("`%s` = %s" % r for r in rows) + "."
p = os.path.join(root, "DESIGN.md")
s = open(p).read()
s2 = re.sub(r"\*\*The table\*\* \(round 5;.*?\n", txt + "\n", s, count=1, flags=re.S)
assert s2 != s or txt in s
open(p, "w").write(s2)
print(len(rows), "switches")
