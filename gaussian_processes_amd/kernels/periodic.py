"""PeriodicKernel -- API mirror of gp/kernels/periodic.py:14-190, HIP-backed.

K(x1, x2) = h^2 exp(-2 sin^2((x1 - x2) / (2 p)) / w^2)  (RW06 eq. 4.31); no
underflow clamp, as in periodic_c.pyx:27-30.  (n, d) inputs use the sum of the
per-dimension sin^2 terms in the exponent; the derivative members are 1-D only.
"""
import numpy as np

from .. import _lib
from ..ext import periodic_c
from .base import Kernel
from ._native import DTYPE, EPS, member_matrix, positive_param

__all__ = ["PeriodicKernel"]


class PeriodicKernel(Kernel):
    _native_kernel = _lib.KERNEL_PERIODIC
    _param_names = ("h", "w", "p")

    def __init__(self, h, w, p):
        self.h = None   #: output scale
        self.w = None   #: input scale
        self.p = None   #: period
        self.set_param("h", h)
        self.set_param("w", w)
        self.set_param("p", p)

    @property
    def params(self):
        """``(h, w, p)`` as a float64 array."""
        return np.array([self.h, self.w, self.p], dtype=DTYPE)

    @params.setter
    def params(self, val):
        self.set_param("h", val[0])
        self.set_param("w", val[1])
        self.set_param("p", val[2])

    def set_param(self, name, val):
        if name not in self._param_names:
            raise ValueError("unknown parameter: %s" % name)
        setattr(self, name, positive_param(name, val))

    @property
    def sym_K(self):
        import sympy as sym
        h, w, p, d = sym.Symbol("h"), sym.Symbol("w"), sym.Symbol("p"), sym.Symbol("d")
        return h ** 2 * sym.exp(-2.0 * (sym.sin(d / (2.0 * p)) ** 2) / w ** 2)

    def _member(self, member, x1, x2, out):
        return member_matrix(self._native_kernel, member, (self.h, self.w, self.p), x1, x2, out)

    def K(self, x1, x2, out=None):
        return self._member(_lib.K, x1, x2, out)

    def jacobian(self, x1, x2, out=None):
        if out is None:
            out = np.empty((3, x1.shape[0], x2.shape[0]), dtype=DTYPE)
        periodic_c.jacobian(out, x1, x2, self.h, self.w, self.p)
        return out

    def hessian(self, x1, x2, out=None):
        if out is None:
            out = np.empty((3, 3, x1.shape[0], x2.shape[0]), dtype=DTYPE)
        periodic_c.hessian(out, x1, x2, self.h, self.w, self.p)
        return out


def _add_member(name):
    member = _lib.MEMBER_BY_NAME[name]

    def f(self, x1, x2, out=None):
        return self._member(member, x1, x2, out)
    f.__name__ = name
    setattr(PeriodicKernel, name, f)


for _name in ("dK_dh", "dK_dw", "dK_dp", "d2K_dhdh", "d2K_dhdw", "d2K_dhdp", "d2K_dwdh",
              "d2K_dwdw", "d2K_dwdp", "d2K_dpdh", "d2K_dpdw", "d2K_dpdp"):
    _add_member(_name)
