"""Drop-in for the reference's Cython module gp/ext/periodic_c.pyx (HIP-backed)."""
from .. import _lib
from ._buffers import as_buffer, check_out

__all__ = ["K", "jacobian", "hessian", "dK_dh", "dK_dw", "dK_dp", "d2K_dhdh", "d2K_dhdw",
           "d2K_dhdp", "d2K_dwdh", "d2K_dwdw", "d2K_dwdp", "d2K_dpdh", "d2K_dpdw", "d2K_dpdp"]


def _member(member, out, x1, x2, h, w, p):
    out = as_buffer(out, 2, "out")
    x1 = as_buffer(x1, 1, "x1")
    x2 = as_buffer(x2, 1, "x2")
    check_out(out, (x1.size, x2.size))
    _lib.check(_lib.load().gpx_periodic_c(member, _lib.dptr(out), _lib.dptr(x1), x1.size,
                                          _lib.dptr(x2), x2.size, float(h), float(w), float(p)))


def K(out, x1, x2, h, w, p):          # periodic_c.pyx:18-30
    _member(_lib.K, out, x1, x2, h, w, p)


def jacobian(out, x1, x2, h, w, p):   # periodic_c.pyx:33-36
    out = as_buffer(out, 3, "out")
    x1 = as_buffer(x1, 1, "x1")
    x2 = as_buffer(x2, 1, "x2")
    check_out(out, (3, x1.size, x2.size))
    _lib.check(_lib.load().gpx_periodic_c_jacobian(_lib.dptr(out), _lib.dptr(x1), x1.size,
                                                   _lib.dptr(x2), x2.size, float(h), float(w),
                                                   float(p)))


def hessian(out, x1, x2, h, w, p):    # periodic_c.pyx:39-50
    out = as_buffer(out, 4, "out")
    x1 = as_buffer(x1, 1, "x1")
    x2 = as_buffer(x2, 1, "x2")
    check_out(out, (3, 3, x1.size, x2.size))
    _lib.check(_lib.load().gpx_periodic_c_hessian(_lib.dptr(out), _lib.dptr(x1), x1.size,
                                                  _lib.dptr(x2), x2.size, float(h), float(w),
                                                  float(p)))


def _make(name, member):
    def f(out, x1, x2, h, w, p):
        _member(member, out, x1, x2, h, w, p)
    f.__name__ = name
    return f


dK_dh = _make("dK_dh", _lib.DK_DH)            # periodic_c.pyx:53-65
dK_dw = _make("dK_dw", _lib.DK_DW)            # :68-80
dK_dp = _make("dK_dp", _lib.DK_DP)            # :83-96
d2K_dhdh = _make("d2K_dhdh", _lib.D2K_DHDH)   # :99-111
d2K_dhdw = _make("d2K_dhdw", _lib.D2K_DHDW)   # :114-126
d2K_dhdp = _make("d2K_dhdp", _lib.D2K_DHDP)   # :129-142
d2K_dwdh = _make("d2K_dwdh", _lib.D2K_DHDW)   # :145-157
d2K_dwdw = _make("d2K_dwdw", _lib.D2K_DWDW)   # :160-172
d2K_dwdp = _make("d2K_dwdp", _lib.D2K_DWDP)   # :175-188
d2K_dpdh = _make("d2K_dpdh", _lib.D2K_DHDP)   # :191-204
d2K_dpdw = _make("d2K_dpdw", _lib.D2K_DWDP)   # :207-220
d2K_dpdp = _make("d2K_dpdp", _lib.D2K_DPDP)   # :223-235
