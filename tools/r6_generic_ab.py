"""tools/r6_generic_ab.py [lib.so] -- the generic (K tail / unaligned) fp64 / fp32 GEMM route on the clock: K = 1000 is no
multiple of the 128-byte k-step.  With a library path: that build instead of the in-tree one (A/B of a kernel change)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from gaussian_processes_amd.device import DeviceBuffer, Event
lib = _lib.load()
for dt, npdt in ((_lib.F64, np.float64), (_lib.F32, np.float32)):
    for (M, N, K, tri) in ((4096, 4096, 1000, 0), (8192, 8192, 1000, 1), (8191, 257, 776, 0), (16384, 1024, 1000, 0)):
        rng = np.random.RandomState(0)
        ld = K + (8 if dt == _lib.F64 else 4) * 2
        A = DeviceBuffer.from_host(rng.randn(M, ld).astype(npdt)); B = DeviceBuffer.from_host(rng.randn(N, ld).astype(npdt))
        C = DeviceBuffer((M, N), npdt).zero()
        best = 1e9
        _lib.route_reset()
        for r in range(4):
            e0, e1 = Event(), Event()
            e0.record()
            _lib.check(lib.gpx_d_gemm_nt(dt, M, N, K, -1.0, A.ptr, ld, B.ptr, ld, C.ptr, N, tri, 0, 0, None))
            e1.record(); e1.sync()
            if r:
                best = min(best, e0.elapsed_ms(e1))
        assert _lib.route_count(_lib.ROUTE_GEMM_GENERIC) == 4
        fl = 2.0 * M * N * K * (0.5 if tri else 1.0)
        print("%s %5d x %5d x %4d tri=%d generic route: %8.3f ms %7.2f TF/s" % ("f64" if dt == _lib.F64 else "f32", M, N, K, tri, best, fl / best / 1e9), flush=True)
        for b in (A, B, C):
            b.free()
