"""tools/pmc_summary.py <dir> -- per-kernel sums of rocprofv3 --pmc counter CSVs (pmc_FETCH_SIZE/, pmc_WRITE_SIZE/
under <dir>) and the traffic JSON of the trailing-update kernel that bench.py reports as roofline.traffic.
Corrections exactly as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE doubled for wide
coalesced reads, WRITE_SIZE as read; both counters are in KB."""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

out = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(out, "pmc_" + cname, "**", "*counter_collection.csv"), recursive=True)
    tot, cnt = defaultdict(float), defaultdict(int)
    summary = os.path.join(out, "bench_%s_by_kernel.csv" % cname)
    if not files and os.path.exists(summary):            # raw per-dispatch CSVs already dropped: re-read the sums
        for r in csv.DictReader(open(summary)):
            tot[r["Kernel_Name"]] = float(r["Sum_KB"]); cnt[r["Kernel_Name"]] = int(r["Dispatches"])
        res[cname] = (tot, cnt)
        continue
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != cname:
                continue
            tot[r["Kernel_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]] += 1
    with open(summary, "w") as g:
        g.write("Kernel_Name,Dispatches,Counter_Name,Sum_KB,Per_Dispatch_KB\n")
        for k in sorted(tot, key=lambda k: -tot[k]):
            g.write('"%s",%d,%s,%.1f,%.1f\n' % (k, cnt[k], cname, tot[k], tot[k] / cnt[k]))
    res[cname] = (tot, cnt)


def schedule(N):
    """(k0, width) of every panel: gpx_potrf.hip outer_block() re-evaluated with the rows left (256 / 512 / 1024 for
    <= 8192 / <= 12288 / more)."""
    out, k0 = [], 0
    while k0 < N:
        left = N - k0
        w = 256 if left <= 8192 else 512 if left <= 12288 else 1024
        w = min(w, left)
        out.append((k0, w)); k0 += w
    return out


key = [k for k in res["FETCH_SIZE"][0] if "gemm_nt_fast_kernel<double, 128, 1, 0>" in k]
if key:
    k = key[0]
    n = res["FETCH_SIZE"][1][k]
    fetch = res["FETCH_SIZE"][0][k] / n * 1024.0 * 2.0
    write = res["WRITE_SIZE"][0][k] / res["WRITE_SIZE"][1][k] * 1024.0
    N, fits = 65536, 2                                  # the profiled command runs one warm-up and one timed fit
    per_fit = n / float(fits)
    # algorithmic bytes of one fit's trailing updates: every element of C that a launch updates is read and written once
    # (8 B each way counted once: WRITE_SIZE sees the atomic add's write), and the launch's panel operand is read once;
    # divided by the launches of that kernel per fit.  The launches are potrf()'s (csrc/gpx_potrf.hip), restated: the pair
    # phase while >= PAIR_ROWS rows lie beyond a pair (far updates of depth K = 2048: U_a, U_b one block column each, U_c the
    # rest; V is a panel-class product and not this kernel), then one panel per update with the next block column first.
    # A launch of at most 2600 tiles of 128 x 128 goes to the 128 x 64 kernel (GPX_SYRK_BN64_TILES) and is not counted here.
    BN64_TILES = 2600
    PAIR_ROWS = int(os.environ.get("GPX_POTRF_PAIR_ROWS", "0")) or (1 << 60)      # (the library's default: no pair phase)

    def width(left):
        return 256 if left <= 8192 else 512 if left <= 12288 else 1024

    def nominal(k):
        return min(1024, width(N - k))

    def c_elems(r0, c0, c1):
        """elements (row >= col) of rows >= r0, columns [c0, c1) of the N x N matrix"""
        tot = 0.0
        for c in (c0, ):
            pass
        a = max(c0, min(c1, r0))                      # columns [c0, a): full height from r0; [a, c1): from the diagonal
        tot += (a - c0) * float(N - r0)
        if c1 > a:
            tot += (c1 - a) * float(N) - 0.5 * (a + c1 - 1) * (c1 - a)
        return tot

    launches = []                                      # (row_begin, c0, c1, K)

    def pair_ok(k):
        return N - (k + 4096) >= PAIR_ROWS and nominal(k + 1024) == 1024 and nominal(k + 2048) == 1024 and nominal(k + 3072) == 1024

    k0, kb = 0, min(nominal(0), N)
    if kb == 1024 and pair_ok(k0):
        launches.append((k0 + 1024, k0 + 1024, k0 + 2048, 1024))
        while pair_ok(k0):
            rC, rD, rE = k0 + 2048, k0 + 3072, k0 + 4096
            launches += [(rC, rC, rD, 2048), (rD, rD, rE, 2048), (rE, rE, N, 2048)]
            k0 = rC
        launches.append((k0 + 2048, k0 + 2048, N, 1024))
        k0, kb = k0 + 1024, 1024
    while k0 + kb < N:
        r = k0 + kb
        kb1 = min(nominal(r), N - r)
        launches.append((r, r, r + kb1, kb))
        if r + kb1 < N:
            launches.append((r, r + kb1, N, kb))
        k0, kb = r, kb1
    c_tot = p_tot = 0.0
    modelled = 0
    for (r0, c0, c1, K) in launches:
        tr, tc = -(-(N - r0) // 128), -(-(c1 - c0) // 128)
        if min(tr * (tr + 1) // 2, tr * tc) <= BN64_TILES:
            continue
        c_tot += 8.0 * c_elems(r0, c0, c1)
        p_tot += 8.0 * (N - min(r0, c0)) * K           # the panel rows this launch reads once (A and B operand rows coincide)
        modelled += 1
    c_bytes, p_bytes = c_tot / max(1, modelled), p_tot / max(1, modelled)
    src = open(os.path.join(root, "gaussian_processes_amd", "csrc", "gpx_gemm.hip"), "rb").read()
    json.dump({
        "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python3 bench.py --steps 1 "
                   "--warmup 1 --no-cpu-baseline --no-secondary --no-prof  (two separate passes)",
        "workload": "N=65536 d=32 f64, 1 GPU", "kernel": "gpx::gemm_nt_fast_kernel<double, 128, 1, 0>",
        "dispatches_profiled": n, "launches_per_step": per_fit, "launches_per_step_modelled": modelled,
        "fetch_bytes_per_launch_corrected": fetch, "write_bytes_per_launch": write,
        "traffic_bytes_per_launch": fetch + write,
        "correction": "FETCH_SIZE doubled (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section); WRITE_SIZE as read; unit KB",
        "algorithmic_bytes_per_launch": {"C_lower_triangle_update": c_bytes, "panel_operand_once": p_bytes},
        "traffic_over_algorithmic": (fetch + write) / (c_bytes + p_bytes),
        "gemm_source_sha256": hashlib.sha256(src).hexdigest(),
    }, open(os.path.join(out, "traffic_n65536.json"), "w"), indent=1)
    print("traffic per launch: fetch %.3e + write %.3e = %.3e B (%.2fx algorithmic)" % (fetch, write, fetch + write, (fetch + write) / (c_bytes + p_bytes)))
