"""tools/gemm_bench.py -- time gpx_d_gemm_nt over shapes (diagnostic).  usage: python tools/gemm_bench.py"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib
from gaussian_processes_amd.device import DeviceBuffer, Event, sync

lib = _lib.load()
rng = np.random.RandomState(0)


def run(M, N, K, tri, reps=3, dtype=_lib.F64, lda=None):
    es = 8 if dtype == _lib.F64 else 4
    npdt = np.float64 if dtype == _lib.F64 else np.float32
    if lda is None:
        lda = K
        A = DeviceBuffer.from_host(rng.randn(max(M, N), K).astype(npdt))
    else:                       # operand rows `lda` elements apart, as inside the big matrix (contents irrelevant)
        A = DeviceBuffer((max(M, N), lda), npdt).zero()
    C = DeviceBuffer((M, N), npdt).zero()
    best = 1e9
    for r in range(reps + 1):
        e0, e1 = Event(), Event()
        e0.record()
        _lib.check(lib.gpx_d_gemm_nt(dtype, M, N, K, -1.0, A.ptr, lda, A.ptr, lda, C.ptr, N, tri, 0, 0, None))
        e1.record(); e1.sync()
        if r > 0:
            best = min(best, e0.elapsed_ms(e1))
    fl = (M * (M + 1) if tri else 2 * M * N) * K
    print("M=%6d N=%6d K=%5d lda=%6d %s  %8.3f ms  %6.2f TF/s" % (M, N, K, lda, "LOWER" if tri else "FULL ", best, fl / best / 1e9), flush=True)
    A.free(); C.free()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "k1024":       # the trailing updates of the n = 65536 factorisation
        for M in (4096, 8192, 16384, 24576, 32768, 49152, 61440):
            run(M, M, 1024, 1, reps=2)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "strided":     # the same with the panel where potrf leaves it
        for M in (16384, 32768, 49152, 61440):
            run(M, M, 1024, 1, reps=2)
            run(M, M, 1024, 1, reps=2, lda=65536)
        sys.exit(0)
    for (M, K) in [(4096, 512), (8192, 512), (16384, 512), (32768, 512), (49152, 512)]:
        run(M, M, K, 1)
    for K in (64, 128, 256, 1024, 2048, 4096):
        run(16384, 16384, K, 1)
    run(16384, 16384, 512, 0)
    run(32768, 64, 448, 0)
    run(32768, 128, 448, 0)
    run(65536, 512, 512, 0)
