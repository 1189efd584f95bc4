"""Drop-in for the reference's Cython module gp/ext/gaussian_c.pyx.

Same function names and argument meaning -- ``K(out, x1, x2, h, w)`` writes the
(n, m) kernel matrix into `out` in place and returns None -- but every entry is
computed by the HIP kernel-matrix kernel of libgpx.so (csrc/gpx_kmat.hip).
"""
from .. import _lib
from ._buffers import as_buffer, check_out

__all__ = ["K", "jacobian", "hessian", "dK_dh", "dK_dw", "d2K_dhdh", "d2K_dhdw", "d2K_dwdh",
           "d2K_dwdw", "MIN"]

MIN = _lib.MIN_LOG   # gaussian_c.pyx:15


def _member(member, out, x1, x2, h, w):
    out = as_buffer(out, 2, "out")
    x1 = as_buffer(x1, 1, "x1")
    x2 = as_buffer(x2, 1, "x2")
    check_out(out, (x1.size, x2.size))
    _lib.check(_lib.load().gpx_gaussian_c(member, _lib.dptr(out), _lib.dptr(x1), x1.size,
                                          _lib.dptr(x2), x2.size, float(h), float(w)))


def K(out, x1, x2, h, w):             # gaussian_c.pyx:18-36
    _member(_lib.K, out, x1, x2, h, w)


def jacobian(out, x1, x2, h, w):      # gaussian_c.pyx:39-41
    out = as_buffer(out, 3, "out")
    x1 = as_buffer(x1, 1, "x1")
    x2 = as_buffer(x2, 1, "x2")
    check_out(out, (2, x1.size, x2.size))
    _lib.check(_lib.load().gpx_gaussian_c_jacobian(_lib.dptr(out), _lib.dptr(x1), x1.size,
                                                   _lib.dptr(x2), x2.size, float(h), float(w)))


def hessian(out, x1, x2, h, w):       # gaussian_c.pyx:44-48
    out = as_buffer(out, 4, "out")
    x1 = as_buffer(x1, 1, "x1")
    x2 = as_buffer(x2, 1, "x2")
    check_out(out, (2, 2, x1.size, x2.size))
    _lib.check(_lib.load().gpx_gaussian_c_hessian(_lib.dptr(out), _lib.dptr(x1), x1.size,
                                                  _lib.dptr(x2), x2.size, float(h), float(w)))


def dK_dh(out, x1, x2, h, w):         # gaussian_c.pyx:51-69
    _member(_lib.DK_DH, out, x1, x2, h, w)


def dK_dw(out, x1, x2, h, w):         # gaussian_c.pyx:72-92
    _member(_lib.DK_DW, out, x1, x2, h, w)


def d2K_dhdh(out, x1, x2, h, w):      # gaussian_c.pyx:95-113
    _member(_lib.D2K_DHDH, out, x1, x2, h, w)


def d2K_dhdw(out, x1, x2, h, w):      # gaussian_c.pyx:116-136
    _member(_lib.D2K_DHDW, out, x1, x2, h, w)


def d2K_dwdh(out, x1, x2, h, w):      # gaussian_c.pyx:139-140
    _member(_lib.D2K_DHDW, out, x1, x2, h, w)


def d2K_dwdw(out, x1, x2, h, w):      # gaussian_c.pyx:143-164
    _member(_lib.D2K_DWDW, out, x1, x2, h, w)
