"""tools/r6_fetch_probe.py [M] [K] -- ONE size of the trailing update alone, a few launches (run it under
rocprofv3 --kernel-trace --pmc FETCH_SIZE from the tree whose library is to be measured: A/B of a kernel change's L2-miss traffic)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib
from gaussian_processes_amd.device import DeviceBuffer, Event
lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
if len(sys.argv) > 3:
    os.environ["GPX_GEMM_ABLATE"] = sys.argv[3]
n = M + K
A = DeviceBuffer((n, n), np.float64).zero()
for r in range(4):
    e0, e1 = Event(), Event()
    e0.record()
    _lib.check(lib.gpx_d_syrk_bc(_lib.F64, n, K, A.ptr, n, K, n, A.ptr, n, 0, K, 1024, 1, 0, None))
    e1.record(); e1.sync()
    print("M=%d K=%d launch %d: %.3f ms" % (M, K, r, e0.elapsed_ms(e1)), flush=True)
