"""tools/gemm_one.py M K [reps] -- launch the trailing-update kernel (gpx_d_syrk_bc, P = 1) on an
M x M lower triangle with a K-wide panel; used under rocprofv3 --pmc for HBM traffic."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib
from gaussian_processes_amd.device import DeviceBuffer, sync
lib = _lib.load()
M, K = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rng = np.random.RandomState(0)
n = M + K
P = DeviceBuffer.from_host(rng.randn(n, K))         # panel rows k0.. = all rows (k0 = 0)
C = DeviceBuffer((n, n)).zero()
for _ in range(reps):
    _lib.check(lib.gpx_d_syrk_bc(_lib.F64, n, K, C.ptr, n, K, n, P.ptr, K, 0, K, 512, 1, 0, None))
sync()
print("ok", M, K)
