"""tools/syrk_bc_bench.py -- one rank's trailing update in the block-cyclic layout (gpx_d_syrk_bc, P ranks, rank 1),
standalone on one GPU: the work a rank of the multi-GPU factorisation does per panel (diagnostic).
usage: syrk_bc_bench.py [N] [nb] ; run with GPX_GEMM_EXACT=0 for the patch-grid map."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib
from gaussian_processes_amd.device import DeviceBuffer, Event

lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
for P in (2, 4, 8):
    rank = 1
    nblk = N // nb
    mine = [j for j in range(nblk) if j % P == rank]
    ncl = len(mine) * nb
    C = DeviceBuffer((N, ncl), np.float64).zero()
    Pb = DeviceBuffer((N, nb), np.float64).zero()
    for k in (0, nblk // 2):
        k0 = k * nb
        first = next(jl for jl, j in enumerate(mine) if j > k)
        cl0 = first * nb
        best = 1e9
        for r in range(3):
            e0, e1 = Event(), Event()
            e0.record()
            _lib.check(lib.gpx_d_syrk_bc(_lib.F64, N, k0 + nb, C.ptr, ncl, cl0, ncl, Pb.ptr, nb, k0, nb, nb, P, rank, None))
            e1.record(); e1.sync()
            if r > 0:
                best = min(best, e0.elapsed_ms(e1))
        fl = 0.0
        for jl in range(first, len(mine)):
            g = mine[jl] * nb
            fl += sum(2.0 * nb * (N - max(k0 + nb, g + c)) for c in range(0, nb, 64)) * 64
        print("P=%d rank %d panel %3d: %8.3f ms  %6.2f TF/s" % (P, rank, k, best, fl / best / 1e9), flush=True)
    C.free(); Pb.free()
