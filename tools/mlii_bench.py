"""tools/mlii_bench.py -- config 5: 64 ML-II restarts at N=8192, d=8 on one GPU, by concurrency (diagnostic)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import mlii
N, d, R = 8192, 8, 64
rng = np.random.RandomState(0)
X = rng.uniform(-10, 10, (N, d)); y = np.sin(X.sum(1) / np.sqrt(d)) + 0.1 * rng.randn(N)
r2 = np.random.RandomState(2)
thetas = np.stack([r2.uniform(0.5, 2, R), r2.uniform(0.25, 2, R) * np.sqrt(d), r2.uniform(0.5, 2, R)], 1)   # h, w, s
ref = None
for c in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    mlii.log_lh_batch(X, y, thetas[:c], concurrency=c)           # warm-up (library load, scratch)
    t0 = time.perf_counter()
    out = mlii.log_lh_batch(X, y, thetas, concurrency=c)
    dt = time.perf_counter() - t0
    if ref is None:
        ref = out
    print("concurrency %d: %d restarts in %.3f s = %.1f ms per restart; max |diff| vs first run %.2e" % (c, R, dt, dt / R * 1e3, np.nanmax(np.abs(np.where(np.isfinite(out) & np.isfinite(ref), out - ref, 0.0))) + (0 if (np.isfinite(out) == np.isfinite(ref)).all() else np.inf)))
