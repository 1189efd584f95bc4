"""tools/r3_batch_after_fit.py -- does a single-matrix fit earlier in the process change the time of a later lock-step batch of 8?
(diagnostic for the operator stream of gpx_gp_fit)"""
import os, sys, time, statistics, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_processes_amd as gp
from gaussian_processes_amd import mlii
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
N, d = 8192, 8
X, y, _ = bench.synth(N, d, 4, np.float64)
if mode == "dummy_mem":
    from gaussian_processes_amd.device import DeviceBuffer
    keep = DeviceBuffer((8193, 8192), np.float64)
elif mode == "stream_only":
    import ctypes
    from gaussian_processes_amd import _lib
    lib = _lib.load()
    hh = ctypes.c_void_p()
    _lib.check(lib.gpx_gp_create(ctypes.byref(hh), _lib.F64, _lib.KERNEL_GAUSSIAN, 64, 1))   # a tiny handle: stream + events, no memory to speak of
elif mode == "other_evaluator":
    other = mlii.BatchEvaluator(X, y)
    print("other evaluator", other(np.array([[1.0, 1.4, 1.0]])))
elif mode == "tiny_fit":
    Xs, ys, _ = bench.synth(512, d, 4, np.float64)
    g = gp.GP(gp.GaussianKernel(1.0, 0.5 * np.sqrt(d)), Xs, ys, s=1.0)
    print("tiny fit log_lh", float(g.log_lh))
elif mode != "none":
    g = gp.GP(gp.GaussianKernel(1.0, 0.5 * np.sqrt(d)), X, y, s=1.0)
    print("fit log_lh", float(g.log_lh))
    if mode == "fit_then_free":
        del g; gc.collect()
rs = np.random.RandomState(2)
w = rs.uniform(0.25, 2, 64) * np.sqrt(d); h = rs.uniform(0.5, 2, 64); sn = rs.uniform(0.5, 2, 64)
th = np.column_stack([h, w, sn])
with mlii.BatchEvaluator(X, y) as ev:
    ev(th[:8]); ev(th[:8])
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); ev(th[:8]); ts.append(time.perf_counter() - t0)
print("%-14s OPS_AHEAD=%s: 8 restarts %.4f s" % (mode, os.environ.get("GPX_FIT_OPS_AHEAD", "1"), statistics.median(ts)))
