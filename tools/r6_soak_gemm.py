"""Repeat-run soak of gpx_d_gemm_nt itself, beside a neighbour thread that runs products of its own: for a list of shapes
that reach every instantiation (128 x 128 and 128 x 64 tiles, the generic kernel for K tails / unaligned operands, lower
triangles, fp64 and fp32) the product C -= A B^T is repeated from the same C, every result equal to the first bit for bit.

    python tools/r6_soak_gemm.py [reps=300]
"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib                      # noqa: E402
from gaussian_processes_amd.device import DeviceBuffer, sync  # noqa: E402

reps = int(dict(a.split("=", 1) for a in sys.argv[1:]).get("reps", 300))
lib = _lib.load()
stop = threading.Event()


def neighbour():
    rng = np.random.RandomState(9)
    A = DeviceBuffer.from_host(rng.randn(3072, 512))
    C = DeviceBuffer.from_host(np.zeros((3072, 3072)))
    Af = DeviceBuffer.from_host(rng.randn(2048, 256).astype(np.float32))
    Cf = DeviceBuffer.from_host(np.zeros((2048, 2048), np.float32))
    while not stop.is_set():
        _lib.check(lib.gpx_d_gemm_nt(_lib.F64, 3072, 3072, 512, -1.0, A.ptr, 512, A.ptr, 512, C.ptr, 3072, _lib.LOWER, 0, 0, None))
        _lib.check(lib.gpx_d_gemm_nt(_lib.F32, 2048, 2048, 256, -1.0, Af.ptr, 256, Af.ptr, 256, Cf.ptr, 2048, _lib.FULL, 0, 0, None))
        sync()


t = threading.Thread(target=neighbour)
t.start()
total_bad = 0
try:
    # (M, N, K, tri): aligned big (fast 128 x 128), narrow (128 x 64), K tail / odd sizes (generic), lower triangles
    shapes = [(4096, 4096, 256, "full"), (4096, 4096, 1024, "lower"), (8192, 256, 256, "full"), (6144, 6144, 512, "lower"),
              (2048, 64, 1024, "full"), (1025, 515, 160, "full"), (2500, 2500, 200, "lower"), (3000, 1000, 37, "full"),
              (8192, 8192, 256, "lower"), (512, 512, 4096, "full")]
    for dt, npdt in ((_lib.F64, np.float64), (_lib.F32, np.float32)):
        for M, N, K, tri in shapes:
            rng = np.random.RandomState(M + N + K)
            lda = ((K + 15) // 16) * 16
            ldc = ((N + 15) // 16) * 16
            A = np.zeros((M, lda), npdt); A[:, :K] = rng.randn(M, K)
            B = np.zeros((N, lda), npdt); B[:, :K] = rng.randn(N, K)
            C0 = np.zeros((M, ldc), npdt); C0[:, :N] = rng.randn(M, N)
            dA, dB = DeviceBuffer.from_host(A), DeviceBuffer.from_host(B if tri == "full" else A[:N].copy())
            dC0 = DeviceBuffer.from_host(C0)
            dC = DeviceBuffer((M, ldc), npdt)
            first, bad, t0 = None, 0, time.time()
            n_rep = int(min(reps, max(30, reps * (4096.0 * 4096 * 256) / (M * N * K))))    # (about the same time per shape)
            for rep in range(n_rep):
                _lib.check(lib.gpx_memcpy_d2d(dC.ptr, dC0.ptr, M * ldc * np.dtype(npdt).itemsize, None))
                _lib.check(lib.gpx_d_gemm_nt(dt, M, N, K, -1.0, dA.ptr, lda, dB.ptr, lda, dC.ptr, ldc,
                                             _lib.FULL if tri == "full" else _lib.LOWER, 0, 0, None))
                sync()
                got = dC.to_host()
                if first is None:
                    first = got
                elif not np.array_equal(got, first):
                    bad += 1
                    d = np.nonzero(got != first)
                    if bad <= 2:
                        print("   rep %d: %d entries differ, first at (%d, %d): %r vs %r" % (rep, d[0].size, d[0][0], d[1][0],
                                                                                             got[d[0][0], d[1][0]], first[d[0][0], d[1][0]]), flush=True)
            total_bad += bad
            print("%s M=%d N=%d K=%d %s: %d of %d repetitions differed (%.1f s)" % ("f64" if dt == _lib.F64 else "f32", M, N, K, tri, bad,
                                                                                  n_rep - 1, time.time() - t0), flush=True)
finally:
    stop.set()
    t.join(60)
print("total differing repetitions: %d" % total_bad)
