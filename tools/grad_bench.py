"""tools/grad_bench.py N d -- time of the device-resident dloglh_dtheta after a fit (diagnostic)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_processes_amd as gp
N, d = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.RandomState(0)
X = rng.uniform(-10, 10, (N, d)); y = np.sin(X.sum(1) / np.sqrt(d)) + 0.1 * rng.randn(N)
g = gp.GP(gp.GaussianKernel(1.0, 0.5 * np.sqrt(d)), X, y, s=1.0)
t0 = time.perf_counter(); llh = g.log_lh; t1 = time.perf_counter()
grad = g.dloglh_dtheta; t2 = time.perf_counter()
g.set_param("h", 1.1)
t3 = time.perf_counter(); llh2 = g.log_lh; grad2 = g.dloglh_dtheta; t4 = time.perf_counter()
print("N=%d d=%d  first fit %.3f s, gradient %.3f s; second fit+gradient %.3f s  (inverse: 2 n^3 / 3 = %.2e flop)  grad=%s"
      % (N, d, t1 - t0, t2 - t1, t4 - t3, 2 * N ** 3 / 3.0, np.array2string(grad, precision=6)))
