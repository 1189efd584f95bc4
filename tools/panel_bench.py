"""tools/panel_bench.py -- time of one uncontended panel factorisation (gpx_d_potrf_panel): N kb (diagnostic)"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib
from gaussian_processes_amd.device import DeviceBuffer, Event, sync
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
d = 8
rng = np.random.RandomState(0)
X = DeviceBuffer.from_host(rng.uniform(-10, 10, (N, d)))
params = np.array([1.0, 0.5 * np.sqrt(d)])
info = DeviceBuffer((4,), np.int32).zero()
LD = int(os.environ.get("PANEL_LD", "0"))       # 0: packed panel (ld = kb); else rows LD elements apart
for kb in [int(a) for a in sys.argv[2:]] or [256, 512, 1024]:
    ld = LD or kb
    A = DeviceBuffer((N, ld))          # the first block column of K + I
    ts = []
    for rep in range(3):
        _lib.check(lib.gpx_d_kmat(_lib.F64, _lib.KERNEL_GAUSSIAN, _lib.K, X.ptr, N, X.ptr, kb, d, _lib.dptr(params), 1.0,
                                  _lib.FULL, A.ptr, ld, None))
        e0, e1 = Event(), Event()
        e0.record(None)
        _lib.check(lib.gpx_d_potrf_panel(_lib.F64, A.ptr, ld, N, 0, 0, kb, info.ptr, None))
        e1.record(None)
        sync()
        ts.append(e0.elapsed_ms(e1))
    t = min(ts[1:])
    print("ld=%d " % ld + "N=%d kb=%4d  panel %.3f ms  (%.1f TF/s of N*kb^2 flop)  info=%d" % (N, kb, t, N * kb * kb / t / 1e9, int(info.to_host()[0])))
    A.free()
