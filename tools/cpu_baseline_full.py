"""tools/cpu_baseline_full.py <part> [N=65536] [d=32] [m=1000] -- the CPU baseline of the headline workload MEASURED AT FULL
SIZE on the GPU box's host cores, once, off the timed bench (north_star: "that CPU path timed on the node's own host cores").

The oracle's stage sequence (oracle/gp_oracle.py = the reference's: C kernel loop, scipy cholesky / cho_solve, numpy
slogdet LU, dot; gp/gp.py:263-367, gp_c.pyx:17-31) at N = 65536, d = 32: ~17 minutes of host work, more than one GPU call
may take, so it is measured in two parts that each build the kernel matrix themselves:
    chol : kmat, potrf (scipy.linalg.cholesky), solve (cho_solve), logdet from diag(L), posterior mean   -- the FAIR variant
    lu   : kmat, slogdet (the LU the reference runs on top, gp_c.pyx:21)
Each part writes gpurun_out/r05_cpu_baseline_n<N>_<part>.json; `merge` combines them into profiles/r05_cpu_baseline_n<N>.json
(bench.py prints that file's numbers beside its sampled extrapolation).  A heartbeat line every 30 s keeps the call alive."""
import json, os, sys, threading, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)

part = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
d = int(sys.argv[3]) if len(sys.argv) > 3 else 32
m = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
out_dir = os.path.join(root, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)

if part == "merge":
    parts = {}
    for p in ("chol", "lu"):
        f = os.path.join(out_dir, "r05_cpu_baseline_n%d_%s.json" % (N, p))
        if os.path.exists(f):
            parts[p] = json.load(open(f))
    assert "chol" in parts, "the chol part is needed"
    c = parts["chol"]
    sec = dict(c["seconds"])
    if "lu" in parts:
        sec["slogdet_lu"] = parts["lu"]["seconds"]["slogdet_lu"]
        sec["kmat_second_build"] = parts["lu"]["seconds"]["kmat"]
    fair = sum(sec[k] for k in ("kmat", "potrf", "solve", "logdet_from_L", "mean"))
    out = {"what": "oracle stage sequence MEASURED at full size on the GPU box's host (tools/cpu_baseline_full.py), off the timed bench",
           "N": N, "d": d, "m": m, "seconds": {k: round(v, 3) for k, v in sec.items()},
           "fair_value_s": round(fair, 3),
           "reference_faithful_value_s": (round(fair - sec["logdet_from_L"] + sec["slogdet_lu"], 3) if "slogdet_lu" in sec else None),
           "gflops": {"potrf": round(N ** 3 / 3.0 / sec["potrf"] / 1e9, 1),
                      "slogdet_lu": (round(2.0 * N ** 3 / 3.0 / sec["slogdet_lu"] / 1e9, 1) if "slogdet_lu" in sec else None)},
           "log_lh": c["log_lh_chol"], "log_lh_lu": parts.get("lu", {}).get("log_lh"),
           "host": c["host"], "parts": parts}
    dst = os.path.join(root, "profiles", "r05_cpu_baseline_n%d.json" % N)
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("seconds", "fair_value_s", "reference_faithful_value_s", "gflops", "log_lh")}))
    sys.exit(0)

t_start = time.perf_counter()
stage = ["start"]
def beat():
    while True:
        time.sleep(30)
        print("[cpu_baseline %s] %.0f s, stage: %s" % (part, time.perf_counter() - t_start, stage[0]), flush=True)
threading.Thread(target=beat, daemon=True).start()

import bench
from oracle import gp_oracle as orc
import scipy.linalg
blas = bench._blas_info()
cpu_model = None
for line in open("/proc/cpuinfo"):
    if line.lower().startswith("model name"):
        cpu_model = line.split(":", 1)[1].strip(); break
host = {"cpu_model": cpu_model, "os_cpu_count": os.cpu_count(), "blas": blas,
        "blas_threads": max([p.get("num_threads") or 1 for p in blas] or [1]),
        "mem_gib": round(os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2.0 ** 30, 1)}
bench._cpu_sample(orc, 768, d, 8)                       # untimed: library loading, thread-pool start-up
X, y, Xo = orc.synth_inputs(N, d, m)
h, w, s = 1.0, 0.5 * np.sqrt(d), 1.0
o = orc.OracleGP("gaussian", (h, w), X, y, s)
sec = {}
def timed(name, f):
    stage[0] = name
    t0 = time.perf_counter(); v = f(); sec[name] = time.perf_counter() - t0
    print("[cpu_baseline %s] %s: %.2f s" % (part, name, sec[name]), flush=True)
    return v
res = {"part": part, "N": N, "d": d, "m": m, "host": host}
timed("kmat", lambda: o.Kxx)
if part == "chol":
    L = timed("potrf", lambda: o.Lxx)
    a = timed("solve", lambda: o.inv_Kxx_y)
    llh = timed("logdet_from_L", lambda: float(o.log_lh_chol))
    mean = timed("mean", lambda: o.mean(Xo))
    res.update({"log_lh_chol": llh, "mean_first": [float(v) for v in mean[:4]], "alpha_first": [float(v) for v in a[:4]]})
elif part == "lu":
    def lu():
        sign, logdet = np.linalg.slogdet(o.Kxx)           # gp_c.pyx:21
        return float(sign), float(logdet)
    sign, logdet = timed("slogdet_lu", lu)
    res.update({"sign": sign, "logdet": logdet, "log_lh": None})
else:
    raise SystemExit("part must be chol | lu | merge")
res["seconds"] = sec
res["total_s"] = time.perf_counter() - t_start
json.dump(res, open(os.path.join(out_dir, "r05_cpu_baseline_n%d_%s.json" % (N, part)), "w"), indent=1)
print(json.dumps({"part": part, "seconds": sec}))
