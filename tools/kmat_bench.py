"""tools/kmat_bench.py -- kernel-matrix build timing (diagnostic): N d [dtype]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib
from gaussian_processes_amd.device import DeviceBuffer, Event, sync
if len(sys.argv) > 4:
    _lib.LIB_PATH = os.path.abspath(sys.argv[4])       # another build of the library (A/B of a kernel change)
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
d = int(sys.argv[2]) if len(sys.argv) > 2 else 32
f32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
dt, npdt, es = (_lib.F32, np.float32, 4) if f32 else (_lib.F64, np.float64, 8)
rng = np.random.RandomState(0)
X = DeviceBuffer.from_host(rng.uniform(-10, 10, (N, d)).astype(npdt))
K = DeviceBuffer((N, N), npdt)
params = np.array([1.0, 0.5 * np.sqrt(d)])
for tri, nm in ((_lib.LOWER, "lower"), (_lib.FULL, "full")):
    ts = []
    for rep in range(4):
        e0, e1 = Event(), Event()
        e0.record(None)
        _lib.check(lib.gpx_d_kmat(dt, _lib.KERNEL_GAUSSIAN, _lib.K, X.ptr, N, X.ptr, N, d, _lib.dptr(params), 1.0,
                                  tri, K.ptr, N, None))
        e1.record(None)
        sync()
        ts.append(e0.elapsed_ms(e1))
    t = min(ts[1:])
    wr = N * N * es * (0.5 if tri == _lib.LOWER else 1.0)
    print("N=%d d=%d %s %-5s %8.3f ms  written %.0f GB/s  (N^2*T accounting %.0f GB/s)" % (N, d, "f32" if f32 else "f64", nm, t, wr / t / 1e6, N * N * es / t / 1e6))
