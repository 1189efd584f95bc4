"""tools/mlii_bench.py -- config 5: 64 ML-II restarts at N=8192, d=8 on one GPU (diagnostic).
Row-at-a-time (by host-thread concurrency) against the lock-step batched route (gpx_gp_fit_batch,
by GPX_BATCH_MAX).  usage: mlii_bench.py [N] ; prints one line per variant and a JSON summary."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import mlii
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d, R = 8, 64
rng = np.random.RandomState(0)
X = rng.uniform(-10, 10, (N, d)); y = np.sin(X.sum(1) / np.sqrt(d)) + 0.1 * rng.randn(N)
r2 = np.random.RandomState(2)
thetas = np.stack([r2.uniform(0.5, 2, R), r2.uniform(0.25, 2, R) * np.sqrt(d), r2.uniform(0.5, 2, R)], 1)   # h, w, s
summary = {"N": N, "d": d, "restarts": R}
mlii.log_lh_batch(X, y, thetas[:2], batched=False)             # warm-up (library load, scratch)
t0 = time.perf_counter(); ref = mlii.log_lh_batch(X, y, thetas, batched=False); dt = time.perf_counter() - t0
print("row at a time, 1 handle : %.3f s = %.2f ms per restart" % (dt, dt / R * 1e3))
summary["row_at_a_time_s"] = dt
t0 = time.perf_counter(); o4 = mlii.log_lh_batch(X, y, thetas, batched=False, concurrency=4); dt = time.perf_counter() - t0
print("row at a time, 4 handles: %.3f s = %.2f ms per restart" % (dt, dt / R * 1e3))
summary["row_at_a_time_4_handles_s"] = dt
for cap in (4, 8, 16, 32, 64):
    os.environ["GPX_BATCH_MAX"] = str(cap)
    t0 = time.perf_counter(); ev = mlii.BatchEvaluator(X, y); out = ev(thetas); cold = time.perf_counter() - t0
    t0 = time.perf_counter(); out = ev(thetas); dt = time.perf_counter() - t0     # steady state: workspace resident
    ev.close()
    summary["batched_%d_cold_s" % cap] = cold
    fin = np.isfinite(out) & np.isfinite(ref)
    err = np.max(np.abs(out[fin] - ref[fin]) / np.abs(ref[fin])) if fin.any() else 0.0
    same = bool((np.isfinite(out) == np.isfinite(ref)).all())
    tf = R * N ** 3 / 3.0 / dt / 1e12
    print("lock-step batches of %2d  : %.3f s = %.2f ms per restart (%.1f TF/s of N^3/3); max rel diff vs row-at-a-time %.1e, same -inf pattern %s"
          % (cap, dt, dt / R * 1e3, tf, err, same))
    summary["batched_%d_s" % cap] = dt
    summary["batched_%d_max_rel_diff" % cap] = float(err)
print(json.dumps(summary))
