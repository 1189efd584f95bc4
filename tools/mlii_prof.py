"""tools/mlii_prof.py -- one warm lock-step sweep of config 5 (64 restarts, N=8192, d=8) for rocprofv3."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import mlii
N, d, R = 8192, 8, 64
rng = np.random.RandomState(0)
X = rng.uniform(-10, 10, (N, d)); y = np.sin(X.sum(1) / np.sqrt(d)) + 0.1 * rng.randn(N)
r2 = np.random.RandomState(2)
thetas = np.stack([r2.uniform(0.5, 2, R), r2.uniform(0.25, 2, R) * np.sqrt(d), r2.uniform(0.5, 2, R)], 1)
ev = mlii.BatchEvaluator(X, y)
ev(thetas)
t0 = time.perf_counter(); out = ev(thetas); dt = time.perf_counter() - t0
print("warm sweep %.3f s" % dt)
ev.close()
