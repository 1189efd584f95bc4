"""tools/r5_syrk_k.py -- the trailing update alone at K = 1024 / 2048 / 3072 (is a deeper operand worth a two-level schedule?)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib
from gaussian_processes_amd.device import DeviceBuffer, Event
lib = _lib.load()
def run(M, K, dtype, reps=3):
    npdt = np.float64 if dtype == _lib.F64 else np.float32
    n = M + K
    ld = n
    A = DeviceBuffer((n, ld), npdt).zero()
    best = 1e9
    for r in range(reps + 1):
        e0, e1 = Event(), Event()
        e0.record()
        _lib.check(lib.gpx_d_syrk_bc(dtype, n, K, A.ptr, ld, K, n, A.ptr, ld, 0, K, 1024, 1, 0, None))
        e1.record(); e1.sync()
        if r > 0:
            best = min(best, e0.elapsed_ms(e1))
    fl = M * (M + 1) * K
    peak = 78.6 if dtype == _lib.F64 else 157.3
    print("%s M=%6d K=%5d  %8.3f ms  %7.2f TF/s  %.4f of peak" % ("f64" if dtype == _lib.F64 else "f32", M, K, best, fl / best / 1e9, fl / best / 1e9 / peak), flush=True)
    A.free()
for dt, Ms in ((_lib.F64, (16384, 32768, 49152)), (_lib.F32, (16384, 32768))):
    for M in Ms:
        for K in (1024, 2048, 3072):
            run(M, K, dt)
