"""tools/interleave_rep.py -- fp64 / fp32 / batched fits interleaved on one host thread, timed step by step (diagnostic)"""
import os, sys, time, faulthandler
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(45, exit=True)
import gaussian_processes_amd as gp
from gaussian_processes_amd import mlii
from oracle import gp_oracle as orc
N, d = 1350, 2
X, y, _ = orc.synth_inputs(N, d, 4)
thetas = np.array([[1.0, 0.9, 1.0], [0.7, 1.4, 0.8], [1.3, 0.6, 1.2]])
ref = [orc.OracleGP("gaussian", th[:2], X, y, th[2]).log_lh for th in thetas]
def single(dtype, i):
    g = gp.GP(gp.GaussianKernel(*thetas[i, :2]), X, y, s=thetas[i, 2], dtype=dtype)
    return float(g.log_lh)
steps = [("f64 single", lambda: single("float64", 0)), ("f32 single", lambda: single("float32", 1)),
         ("f64 batch 3", lambda: mlii.log_lh_batch(X, y, thetas, dtype="float64")), ("f64 single", lambda: single("float64", 2)),
         ("f32 batch 2", lambda: mlii.log_lh_batch(X, y, thetas[:2], dtype="float32")), ("f32 single", lambda: single("float32", 0)),
         ("f64 batch 3", lambda: mlii.log_lh_batch(X, y, thetas, dtype="float64"))]
for name, fn in steps:
    t0 = time.perf_counter()
    v = fn()
    print("%-12s %.3f s  ->" % (name, time.perf_counter() - t0), v, flush=True)
print("ref", ref)
