"""tools/syrk_bench.py -- standalone rate of the trailing update (gpx_d_syrk_bc, P = 1: the exact tile map)
over trailing sizes M and panel depths K, fp64 and fp32 (diagnostic)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib
from gaussian_processes_amd.device import DeviceBuffer, Event

lib = _lib.load()


def run(M, K, dtype, reps=3):
    npdt = np.float64 if dtype == _lib.F64 else np.float32
    n = M + K                                           # panel k0 = 0 of width K, trailing matrix M x M
    ld = n
    A = DeviceBuffer((n, ld), npdt).zero()
    best = 1e9
    for r in range(reps + 1):
        e0, e1 = Event(), Event()
        e0.record()
        _lib.check(lib.gpx_d_syrk_bc(dtype, n, K, A.ptr, ld, K, n, A.ptr, ld, 0, K, K if K <= 1024 else 1024, 1, 0, None))
        e1.record(); e1.sync()
        if r > 0:
            best = min(best, e0.elapsed_ms(e1))
    fl = M * (M + 1) * K
    print("%s M=%6d K=%5d  %8.3f ms  %7.2f TF/s" % ("f64" if dtype == _lib.F64 else "f32", M, K, best, fl / best / 1e9), flush=True)
    A.free()


for dt in (_lib.F64, _lib.F32):
    for K in (256, 512, 1024):
        for M in (4096, 8192, 16384, 32768):
            run(M, K, dt)
