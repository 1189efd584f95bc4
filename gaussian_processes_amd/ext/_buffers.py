"""Argument validation shared by the ext shims.

The reference's Cython signatures are ``np.ndarray[float64, mode='c', ndim=k]``
(gp/ext/gaussian_c.pyx:18); the buffer protocol rejects anything else with a
ValueError, and these are the three messages it uses.  The
reference does NOT check the shape of `out` (boundscheck off); here a wrong
shape is a ValueError instead of a silent overrun.
"""
import numpy as np

DTYPE = np.float64


def as_buffer(a, ndim, name="buffer"):
    if not isinstance(a, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)"
                        % (name, type(a).__name__))
    if a.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions (expected %d, got %d)"
                         % (ndim, a.ndim))
    if a.dtype != DTYPE:
        raise ValueError("Buffer dtype mismatch, expected 'DTYPE_t' but got '%s'" % a.dtype.name)
    if not a.flags.c_contiguous:
        raise ValueError("ndarray is not C-contiguous")
    return a


def check_out(out, shape):
    if tuple(out.shape) != tuple(shape):
        raise ValueError("out has shape %s, expected %s" % (out.shape, tuple(shape)))
    if not out.flags.writeable:
        raise ValueError("buffer source array is read-only")
