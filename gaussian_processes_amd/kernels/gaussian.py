"""GaussianKernel -- API mirror of gp/kernels/gaussian.py:14-144, HIP-backed.

K(x1, x2) = h^2 / sqrt(2 pi w^2) * exp(-(x1 - x2)^2 / (2 w^2)): a *normalised*
Gaussian, with the reference's underflow clamp (entries whose
exponent is below MIN are exactly 0, gaussian_c.pyx:31-34).  Inputs may be (n,)
as in the reference or (n, d); for d > 1 the squared Euclidean distance replaces
(x1 - x2)^2.
"""
import numpy as np

from .. import _lib
from ..ext import gaussian_c
from .base import Kernel
from ._native import DTYPE, EPS, member_matrix, positive_param

__all__ = ["GaussianKernel"]


class GaussianKernel(Kernel):
    _native_kernel = _lib.KERNEL_GAUSSIAN
    _param_names = ("h", "w")

    def __init__(self, h, w):
        self.h = None   #: output scale
        self.w = None   #: input scale (standard deviation of the Gaussian)
        self.set_param("h", h)
        self.set_param("w", w)

    @property
    def params(self):
        """``(h, w)`` as a float64 array."""
        return np.array([self.h, self.w], dtype=DTYPE)

    @params.setter
    def params(self, val):
        self.set_param("h", val[0])
        self.set_param("w", val[1])

    def set_param(self, name, val):
        if name not in self._param_names:
            raise ValueError("unknown parameter: %s" % name)
        setattr(self, name, positive_param(name, val))

    @property
    def sym_K(self):
        import sympy as sym
        h, w, d = sym.Symbol("h"), sym.Symbol("w"), sym.Symbol("d")
        return h ** 2 * (1.0 / sym.sqrt(2 * sym.pi * w ** 2)) * sym.exp(-d ** 2 / (2.0 * w ** 2))

    def _member(self, member, x1, x2, out):
        return member_matrix(self._native_kernel, member, (self.h, self.w), x1, x2, out)

    def K(self, x1, x2, out=None):
        return self._member(_lib.K, x1, x2, out)

    def jacobian(self, x1, x2, out=None):
        if out is None:
            out = np.empty((2, x1.shape[0], x2.shape[0]), dtype=DTYPE)
        if x1.ndim == 1 and x2.ndim == 1:
            gaussian_c.jacobian(out, x1, x2, self.h, self.w)
        else:
            self._member(_lib.DK_DH, x1, x2, out[0])
            self._member(_lib.DK_DW, x1, x2, out[1])
        return out

    def hessian(self, x1, x2, out=None):
        if out is None:
            out = np.empty((2, 2, x1.shape[0], x2.shape[0]), dtype=DTYPE)
        if x1.ndim == 1 and x2.ndim == 1:
            gaussian_c.hessian(out, x1, x2, self.h, self.w)
        else:
            self._member(_lib.D2K_DHDH, x1, x2, out[0, 0])
            self._member(_lib.D2K_DHDW, x1, x2, out[0, 1])
            self._member(_lib.D2K_DHDW, x1, x2, out[1, 0])
            self._member(_lib.D2K_DWDW, x1, x2, out[1, 1])
        return out

    def dK_dh(self, x1, x2, out=None):
        return self._member(_lib.DK_DH, x1, x2, out)

    def dK_dw(self, x1, x2, out=None):
        return self._member(_lib.DK_DW, x1, x2, out)

    def d2K_dhdh(self, x1, x2, out=None):
        return self._member(_lib.D2K_DHDH, x1, x2, out)

    def d2K_dhdw(self, x1, x2, out=None):
        return self._member(_lib.D2K_DHDW, x1, x2, out)

    def d2K_dwdh(self, x1, x2, out=None):
        return self._member(_lib.D2K_DHDW, x1, x2, out)

    def d2K_dwdw(self, x1, x2, out=None):
        return self._member(_lib.D2K_DWDW, x1, x2, out)
