"""Shared machinery of the two built-in (device-native) kernels."""
import numpy as np

from .. import _lib
from ..ext._buffers import as_buffer, check_out

DTYPE = np.float64
EPS = np.finfo(DTYPE).eps


def points(x):
    """Validate an input-location array: (n,) as in the reference, or (n, d)."""
    if not isinstance(x, np.ndarray):
        x = np.asarray(x, dtype=DTYPE)
    if x.ndim not in (1, 2):
        raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % x.ndim)
    if x.dtype != DTYPE:
        raise ValueError("Buffer dtype mismatch, expected 'DTYPE_t' but got '%s'" % x.dtype.name)
    if not x.flags.c_contiguous:
        raise ValueError("ndarray is not C-contiguous")
    n = x.shape[0]
    d = 1 if x.ndim == 1 else x.shape[1]
    return x, n, d


def member_matrix(kernel_id, member, params, x1, x2, out):
    """out (n, m) <- member(x1, x2) for a native kernel; allocates when out is None."""
    x1, n, d = points(x1)
    x2, m, d2 = points(x2)
    if d != d2:
        raise ValueError("x1 and x2 have different dimensionality: %d vs %d" % (d, d2))
    if out is None:
        out = np.empty((n, m), dtype=DTYPE)
    else:
        as_buffer(out, 2, "out")
        check_out(out, (n, m))
    p = np.ascontiguousarray(params, dtype=DTYPE)
    _lib.check(_lib.load().gpx_kmat_host(kernel_id, member, _lib.dptr(out), _lib.dptr(x1), n,
                                         _lib.dptr(x2), m, d, _lib.dptr(p), 0.0))
    return out


def positive_param(name, val):
    """Reference rule: a parameter below machine epsilon is invalid (gaussian.py:62-69)."""
    if val < EPS:
        raise ValueError("invalid value for %s: %s" % (name, val))
    return DTYPE(val)
