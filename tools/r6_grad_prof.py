"""tools/r6_grad_prof.py -- one batched value + gradient sweep of 8 restarts x N = 8192 (run under rocprofv3 --kernel-trace --stats:
where do the gradient's 2 n^3 / 3 flops per row spend their time?)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import mlii
from bench import synth
N, d = 8192, 8
X, y, _ = synth(N, d, 4, np.float64)
rs = np.random.RandomState(2)
w = rs.uniform(0.25, 2, 64) * np.sqrt(d); h = rs.uniform(0.5, 2, 64); sn = rs.uniform(0.5, 2, 64)
th = np.column_stack([h, w, sn])[:8]
with mlii.BatchEvaluator(X, y) as ev:
    ev.value_and_grad(th)
    t0 = time.perf_counter()
    for _ in range(3):
        ev.value_and_grad(th)
    print("value + gradient, 8 rows: %.2f ms per row" % ((time.perf_counter() - t0) / 3 / 8 * 1e3))
    t0 = time.perf_counter()
    for _ in range(3):
        ev(th)
    print("value only, 8 rows: %.2f ms per row" % ((time.perf_counter() - t0) / 3 / 8 * 1e3))
