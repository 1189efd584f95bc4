"""tools/cov_bench.py N d m -- time GP.cov / GP.mean through the public API (diagnostic)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_processes_amd as gp
N, d, m = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.RandomState(0)
X = rng.uniform(-10, 10, (N, d)); y = np.sin(X.sum(1) / np.sqrt(d)) + 0.1 * rng.randn(N)
Xo = np.random.RandomState(1).uniform(-10, 10, (m, d))
g = gp.GP(gp.GaussianKernel(1.0, 0.5 * np.sqrt(d)), X, y, s=1.0)
t0 = time.perf_counter(); llh = g.log_lh; t1 = time.perf_counter()
mean = g.mean(Xo); t2 = time.perf_counter()
cov = g.cov(Xo); t3 = time.perf_counter()
cov2 = g.cov(Xo); t4 = time.perf_counter()
print("N=%d d=%d m=%d  fit %.3fs  mean %.4fs  cov %.3fs (2nd call %.3fs)  llh=%.6f  cov[0,0]=%.6f min diag %.3e sym %.2e"
      % (N, d, m, t1 - t0, t2 - t1, t3 - t2, t4 - t3, llh, cov[0, 0], np.diag(cov).min(), np.abs(cov - cov.T).max()))
