"""tools/trsv_bench.py -- forward / backward single-rhs solve timing on a synthetic factor (diagnostic)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib
from gaussian_processes_amd.device import DeviceBuffer, Event, sync
lib = _lib.load()
for n in [int(a) for a in sys.argv[1:]] or [8192, 65536]:
    L = DeviceBuffer((n, n)).zero()
    # well-conditioned lower-triangular factor: small random strictly-lower part built on the host in
    # row blocks (keeps host memory bounded), unit-ish diagonal
    rng = np.random.RandomState(0)
    rows = 2048
    for r0 in range(0, n, rows):
        blk = rng.uniform(-1, 1, (min(rows, n - r0), n)) / n
        for i in range(blk.shape[0]):
            blk[i, r0 + i] = 1.0 + 0.1 * rng.rand()
            blk[i, r0 + i + 1:] = np.nan          # the upper triangle must never be read
        _lib.check(lib.gpx_memcpy_h2d(ctypes.c_void_p(L.ptr.value + r0 * n * 8), blk.ctypes.data_as(ctypes.c_void_p), blk.nbytes, None))
    y = rng.randn(n)
    b = DeviceBuffer.from_host(y)
    z = DeviceBuffer((n,)).zero()
    a = DeviceBuffer((n,)).zero()
    for tr, src, dst, nm in ((0, b, z, "forward"), (1, z, a, "backward")):
        ts = []
        for rep in range(4):
            if tr == 0:
                b2 = DeviceBuffer.from_host(y)
                src = b2
            else:
                src = DeviceBuffer.from_host(z.to_host())
            e0, e1 = Event(), Event()
            e0.record(None)
            _lib.check(lib.gpx_d_trsv_lower(_lib.F64, L.ptr, n, n, src.ptr, dst.ptr, tr, None))
            e1.record(None)
            sync()
            ts.append(e0.elapsed_ms(e1))
        t = min(ts[1:])
        print("n=%6d %-8s %8.3f ms  %7.1f GB/s (n^2/2 * 8 B)  %6.1f us per 512-block" % (n, nm, t, n * n * 4 / t / 1e6, t * 1e3 / (-(-n // 512))))
    if n <= 8192:
        Lh = np.tril(L.to_host())
        zz = np.linalg.solve(Lh, y)
        aa = np.linalg.solve(Lh.T, zz)
        print("   check: max rel err forward %.2e backward %.2e" % (np.abs(z.to_host() - zz).max() / np.abs(zz).max(), np.abs(a.to_host() - aa).max() / np.abs(aa).max()))
