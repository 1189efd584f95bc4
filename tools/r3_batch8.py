"""tools/r3_batch8.py -- 8 (and 16, 64) restarts x N = 8192 lock-step: seconds per call, median of 5 (diagnostic)."""
import os, sys, time, statistics
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import mlii
import bench
N, d = 8192, 8
X, y, _ = bench.synth(N, d, 4, np.float64)
rs = np.random.RandomState(2)
th = np.column_stack([rs.uniform(0.5, 2, 64), rs.uniform(0.25, 2, 64) * np.sqrt(d), rs.uniform(0.5, 2, 64)])[:, [0, 1, 2]]
with mlii.BatchEvaluator(X, y) as ev:
    for rows in (8, 16, 64):
        ev(th[:rows])
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); ev(th[:rows]); ts.append(time.perf_counter() - t0)
        print("rows %2d: %.4f s = %.3f ms per restart" % (rows, statistics.median(ts), statistics.median(ts) / rows * 1e3))
