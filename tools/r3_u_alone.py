"""tools/r3_u_alone.py -- the trailing update of an n = 8192 factorisation ALONE on the chip (random data): time per
launch for the trailing sizes of steps 0, 4, 8, ... at K = 256 and K = 512, to set against its in-situ times
(gpurun_out/r3_timeline8192.txt: 389 / 300 / 220 us for 5 / 4 / 3 rounds of tiles beside the panel kernel)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib
from gaussian_processes_amd.device import DeviceBuffer, Event
lib = _lib.load()
n = 8192
rng = np.random.RandomState(0)
A = DeviceBuffer.from_host(rng.randn(n, n))
for K in (256, 512):
    for k0 in range(0, n - K, 1024 if K == 256 else 1024):
        M = n - k0 - K
        best = 1e9
        for r in range(4):
            e0, e1 = Event(), Event()
            e0.record()
            _lib.check(lib.gpx_d_syrk_bc(_lib.F64, n, k0 + K, A.ptr, n, k0 + K, n, ctypes.c_void_p(A.ptr.value + (k0 * n + k0) * 8), n, k0, K, K, 1, 0, None))
            e1.record(); e1.sync()
            if r > 0:
                best = min(best, e0.elapsed_ms(e1))
        tiles = (M // 128) * (M // 128 + 1) // 2
        print("K=%d rows left %5d tiles %5d (%.2f rounds of 512)  %7.1f us  %6.2f TF/s" % (K, M, tiles, tiles / 512.0, best * 1e3, M * (M + 1) * K / best / 1e9), flush=True)
