"""tools/pmc_mfma_summary.py <dir> -- per-kernel sums of the SQ / GRBM counters collected by tools/pmc_mfma.sh and the
derived MFMA-busy fraction.

Units (MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units"): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed
over the SIMDs that had an MFMA in flight; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x4);
GRBM_GUI_ACTIVE counts cycles per XCD (summed over the 8 XCDs by rocprofv3).  With 256 CUs x 4 SIMDs:

    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)

which is the fraction of SIMD-cycles with the matrix pipe busy over the kernel's own duration -- the counter-side
reading of bench.py's roofline.frac (flops / duration / peak)."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
files = glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True)
if not files:
    sys.exit("no counter CSV under %s/pmc" % out)
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
         "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"]
rows = []
for k in sorted(tot, key=lambda k: -tot[k].get("GRBM_GUI_ACTIVE", 0.0)):
    t = tot[k]
    gui = t.get("GRBM_GUI_ACTIVE", 0.0)
    simd_cycles = gui / 8.0 * 1024.0
    wave = t.get("SQ_WAVE_CYCLES", 0.0)
    rows.append({
        "kernel": k, "dispatches": cnt[k],
        "mfma_busy_frac": (t.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles) if simd_cycles else None,
        "wait_any_frac_of_wave_cycles": (t.get("SQ_WAIT_ANY", 0.0) / wave) if wave else None,
        "wait_inst_frac_of_wave_cycles": (t.get("SQ_WAIT_INST_ANY", 0.0) / wave) if wave else None,
        "active_inst_frac_of_wave_cycles": (t.get("SQ_ACTIVE_INST_ANY", 0.0) / wave) if wave else None,
        "sums": {n: t.get(n) for n in names},
    })
with open(os.path.join(out, "mfma_by_kernel.json"), "w") as g:
    json.dump({"command": "rocprofv3 --kernel-trace --pmc " + " ".join(names) +
                          " -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-prof",
               "formula": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)",
               "kernels": rows[:12]}, g, indent=1)
for r in rows[:8]:
    print("%-70.70s n=%5d mfma_busy=%s wait=%s" % (r["kernel"], r["dispatches"],
          "%.3f" % r["mfma_busy_frac"] if r["mfma_busy_frac"] is not None else "-",
          "%.3f" % r["wait_any_frac_of_wave_cycles"] if r["wait_any_frac_of_wave_cycles"] is not None else "-"))
