"""Drop-in for the reference's Cython module gp/ext/gp_c.pyx.

In the reference this module is glue: every function is a loop of np.dot /
np.trace / np.linalg.slogdet calls on dense (n, n) matrices (gp_c.pyx:17-131).
Here the same formulas run with their matrix products on the fp64 matrix cores
(gpx_gemm_nt_host -> csrc/gpx_gemm.hip); traces and the final scalar
combinations (O(n) or O(n_params^2) work) stay on the host.

`log_lh(y, K, Kiy)` keeps the reference's signature; the second argument may be
either the kernel matrix K (reference semantics: it is factored on the device,
replacing the LU-based slogdet of gp_c.pyx:21) or, via `log_lh_chol`, its
Cholesky factor.
"""
import ctypes

import numpy as np

from .. import _lib
from ._buffers import as_buffer

__all__ = ["log_lh", "log_lh_chol", "dloglh_dtheta", "dlh_dtheta", "d2lh_dtheta2", "dm_dtheta",
           "MIN"]

DTYPE = np.float64
MIN = _lib.MIN_LOG   # gp_c.pyx:14


def _mm(A, B):
    """np.dot(A, B) for 2-D float64 operands, on the device (C = A (B^T)^T)."""
    A = np.ascontiguousarray(A, dtype=DTYPE)
    Bt = np.ascontiguousarray(np.asarray(B, dtype=DTYPE).T)
    C = np.empty((A.shape[0], Bt.shape[0]), dtype=DTYPE)
    _lib.check(_lib.load().gpx_gemm_nt_host(_lib.dptr(C), _lib.dptr(A), _lib.dptr(Bt),
                                            A.shape[0], Bt.shape[0], A.shape[1]))
    return C


def _mv(A, v):
    """np.dot(A, v) for a matrix and a vector, on the device."""
    return _mm(A, np.asarray(v, dtype=DTYPE).reshape(-1, 1)).ravel()


def log_lh_chol(y, L, Kiy):
    """log_lh given the lower Cholesky factor L of K (logdet = 2 sum log diag L)."""
    y = as_buffer(y, 1, "y")
    L = as_buffer(L, 2, "L")
    Kiy = as_buffer(Kiy, 1, "Kiy")
    out = ctypes.c_double(0.0)
    _lib.check(_lib.load().gpx_gp_c_log_lh(_lib.dptr(y), _lib.dptr(L), _lib.dptr(Kiy), y.size,
                                           ctypes.byref(out)))
    return out.value


def log_lh(y, K, Kiy):                 # gp_c.pyx:17-31
    y = as_buffer(y, 1, "y")
    K = as_buffer(K, 2, "K")
    Kiy = as_buffer(Kiy, 1, "Kiy")
    n = y.size
    L = np.empty((n, n), dtype=DTYPE)
    info = ctypes.c_int(0)
    _lib.check(_lib.load().gpx_cholesky(_lib.dptr(L), _lib.dptr(K), n, ctypes.byref(info)))
    if info.value != 0:                # sign != 1 branch of gp_c.pyx:22
        return -np.inf
    return log_lh_chol(y, L, Kiy)


def _dK(Kj, i, m, s):
    # gp_c.pyx:42-46: the last "parameter" is the noise s, dK/ds = 2 s I
    return Kj[i] if i < Kj.shape[0] else np.eye(m) * 2 * s


def dloglh_dtheta(y, Ki, Kj, Kiy, s, dloglh):      # gp_c.pyx:34-49
    n, m = Kj.shape[0], Kj.shape[1]
    for i in range(n + 1):
        k = _mm(Ki, _dK(Kj, i, m, s))
        t0 = 0.5 * np.dot(y, _mv(k, Kiy))
        t1 = -0.5 * np.trace(k)
        dloglh[i] = t0 + t1


def dlh_dtheta(y, Ki, Kj, Kiy, s, lh, dlh):        # gp_c.pyx:52-67
    n, m = Kj.shape[0], Kj.shape[1]
    for i in range(n + 1):
        k = _mm(Ki, _dK(Kj, i, m, s))
        t0 = np.dot(y, _mv(k, Kiy))
        t1 = np.trace(k)
        dlh[i] = 0.5 * lh * (t0 - t1)


def d2lh_dtheta2(y, Ki, Kj, Kh, Kiy, s, lh, dlh, d2lh):   # gp_c.pyx:70-111
    n, m = Kj.shape[0], Kj.shape[1]
    dK = [_dK(Kj, i, m, s) for i in range(n + 1)]
    dKi = [_mm(-Ki, _mm(dK[i], Ki)) for i in range(n + 1)]
    for i in range(n + 1):
        KidK_i = _mm(Ki, dK[i])
        ydKi_iy = np.dot(y, _mv(KidK_i, Kiy))
        ydKi_iy_tr = ydKi_iy - np.trace(KidK_i)
        for j in range(n + 1):
            if j < n and i < n:
                d2k = Kh[i, j]
            elif j == n and i == n:
                d2k = np.eye(m) * 2
            else:
                d2k = np.zeros((m, m))
            dKi_jdK_i = _mm(dKi[j], dK[i])
            t0 = dlh[j] * ydKi_iy_tr
            t1a = np.dot(y, _mv(dKi_jdK_i, Kiy))
            t1b = np.dot(Kiy, _mv(d2k, Kiy))
            t1c = np.dot(Kiy, _mv(dK[i], _mv(dKi[j], y)))
            t1 = lh * (t1a + t1b + t1c - np.trace(dKi_jdK_i + _mm(Ki, d2k)))
            d2lh[i, j] = 0.5 * (t0 + t1)


def dm_dtheta(y, Ki, Kj, Kjxo, Kxox, s, dm):       # gp_c.pyx:114-131
    n, m = Kj.shape[0], Kj.shape[1]
    Kiy = _mv(Ki, y)
    for i in range(n + 1):
        if i < n:
            dKxox, dKxx = Kjxo[i], Kj[i]
        else:
            dKxox, dKxx = np.zeros_like(Kxox), np.eye(m) * 2 * s
        dm[i] = _mv(dKxox, Kiy)
        dm[i] -= _mv(Kxox, _mv(_mm(Ki, _mm(dKxx, Ki)), y))
