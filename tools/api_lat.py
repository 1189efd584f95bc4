"""tools/api_lat.py [N d m] -- latency of the public calls on a fitted GP: mean, cov, log_lh after set_param (diagnostic)."""
import os, sys, time, statistics
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_processes_amd as gp
N, d, m = (int(a) for a in (sys.argv[1:4] + ["8192", "8", "1024"][len(sys.argv) - 1:]))
rng = np.random.RandomState(0)
X = rng.uniform(-10, 10, (N, d)); y = np.sin(X.sum(1) / np.sqrt(d)) + 0.1 * rng.randn(N)
Xo = np.random.RandomState(1).uniform(-10, 10, (m, d))
g = gp.GP(gp.GaussianKernel(1.0, 0.5 * np.sqrt(d)), X, y, s=1.0)
float(g.log_lh); g.mean(Xo); g.cov(Xo)
def med(f, n=9):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e3
k = [0]
def refit():
    k[0] += 1
    g.set_param("h", 1.0 + k[0] * 1e-9)
    float(g.log_lh)
print("N=%d d=%d m=%d: mean %.3f ms, cov %.3f ms, set_param + log_lh %.3f ms, inv_Kxx_y (memo) %.4f ms"
      % (N, d, m, med(lambda: g.mean(Xo)), med(lambda: g.cov(Xo)), med(refit), med(lambda: g.inv_Kxx_y)))
