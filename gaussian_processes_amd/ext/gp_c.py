"""Drop-in for the reference's Cython module gp/ext/gp_c.pyx.

Same function names and argument lists as the reference (gp_c.pyx:17, :34, :52,
:70, :114): float64 C-contiguous numpy arrays in, results written into the
caller's output array.  In the reference every function is a loop of dense
(n, n) products; here each call is ONE entry of libgpx.so (include/gpx.h,
``gpx_gp_c_*``): the host matrices are uploaded once, the arithmetic is
re-expressed on the device as matrix-vector work, transposed-tile trace
reductions and one GEMM per kernel parameter (csrc/gpx_deriv.hip, "Glue"), and
only the output scalars come back.

`gp.GP` takes this route for plugin kernels only (kernels without a native id);
the built-in kernels never materialise a Jacobian or an inverse at all
(`gpx_gp_dloglh_dtheta`, `gpx_gp_dlh_d2lh`, `gpx_gp_dm_dtheta`).

`log_lh(y, K, Kiy)` keeps the reference's signature; the second argument is the
kernel matrix K (it is factored on the device, replacing the LU-based slogdet of
gp_c.pyx:21); `log_lh_chol` takes its Cholesky factor instead.
"""
import ctypes

import numpy as np

from .. import _lib
from ._buffers import as_buffer, check_out

__all__ = ["log_lh", "log_lh_chol", "dloglh_dtheta", "dlh_dtheta", "d2lh_dtheta2", "dm_dtheta",
           "MIN"]

DTYPE = np.float64
MIN = _lib.MIN_LOG   # gp_c.pyx:14


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


def _common(y, Ki, Kj, Kiy=None):
    """Validated views of the arguments every derivative function shares, and (n, n_kernel_params)."""
    y = as_buffer(y, 1, "y")
    Ki = as_buffer(Ki, 2, "Ki")
    Kj = as_buffer(Kj, 3, "Kj")
    n = y.size
    npar = Kj.shape[0]
    if Ki.shape != (n, n) or Kj.shape[1:] != (n, n):
        raise ValueError("Ki / Kj do not match y: %s, %s, n = %d" % (Ki.shape, Kj.shape, n))
    if Kiy is not None:
        Kiy = as_buffer(Kiy, 1, "Kiy")
        if Kiy.size != n:
            raise ValueError("Kiy has %d entries, expected %d" % (Kiy.size, n))
    return y, Ki, Kj, Kiy, n, npar


def dloglh_dtheta(y, Ki, Kj, Kiy, s, dloglh):      # gp_c.pyx:34-49
    y, Ki, Kj, Kiy, n, npar = _common(y, Ki, Kj, Kiy)
    dloglh = as_buffer(dloglh, 1, "dloglh")
    check_out(dloglh, (npar + 1,))
    _lib.check(_lib.load().gpx_gp_c_dloglh_dtheta(_lib.dptr(y), _lib.dptr(Ki), _lib.dptr(Kj), _lib.dptr(Kiy),
                                                  float(s), n, npar, _lib.dptr(dloglh)))


def dlh_dtheta(y, Ki, Kj, Kiy, s, lh, dlh):        # gp_c.pyx:52-67
    y, Ki, Kj, Kiy, n, npar = _common(y, Ki, Kj, Kiy)
    dlh = as_buffer(dlh, 1, "dlh")
    check_out(dlh, (npar + 1,))
    _lib.check(_lib.load().gpx_gp_c_dlh_dtheta(_lib.dptr(y), _lib.dptr(Ki), _lib.dptr(Kj), _lib.dptr(Kiy),
                                               float(s), float(lh), n, npar, _lib.dptr(dlh)))


def d2lh_dtheta2(y, Ki, Kj, Kh, Kiy, s, lh, dlh, d2lh):   # gp_c.pyx:70-111
    y, Ki, Kj, Kiy, n, npar = _common(y, Ki, Kj, Kiy)
    Kh = as_buffer(Kh, 4, "Kh")
    if Kh.shape != (npar, npar, n, n):
        raise ValueError("Kh has shape %s, expected %s" % (Kh.shape, (npar, npar, n, n)))
    dlh = as_buffer(dlh, 1, "dlh")
    if dlh.size != npar + 1:
        raise ValueError("dlh has %d entries, expected %d" % (dlh.size, npar + 1))
    d2lh = as_buffer(d2lh, 2, "d2lh")
    check_out(d2lh, (npar + 1, npar + 1))
    _lib.check(_lib.load().gpx_gp_c_d2lh_dtheta2(_lib.dptr(y), _lib.dptr(Ki), _lib.dptr(Kj), _lib.dptr(Kh),
                                                 _lib.dptr(Kiy), float(s), float(lh), _lib.dptr(dlh), n, npar,
                                                 _lib.dptr(d2lh)))


def dm_dtheta(y, Ki, Kj, Kjxo, Kxox, s, dm):       # gp_c.pyx:114-131
    y, Ki, Kj, _, n, npar = _common(y, Ki, Kj)
    Kjxo = as_buffer(Kjxo, 3, "Kjxo")
    Kxox = as_buffer(Kxox, 2, "Kxox")
    m = Kxox.shape[0]
    if Kxox.shape != (m, n) or Kjxo.shape != (npar, m, n):
        raise ValueError("Kjxo / Kxox do not match: %s, %s" % (Kjxo.shape, Kxox.shape))
    dm = as_buffer(dm, 2, "dm")
    check_out(dm, (npar + 1, m))
    _lib.check(_lib.load().gpx_gp_c_dm_dtheta(_lib.dptr(y), _lib.dptr(Ki), _lib.dptr(Kj), _lib.dptr(Kjxo),
                                              _lib.dptr(Kxox), float(s), n, npar, m, _lib.dptr(dm)))
