"""Kernel plugin contract (mirrors gp/kernels/base.py:7-121).

`GP` touches a kernel only through: ``K(x1, x2)`` / ``__call__``, the ``params``
property (get, and set in subclasses), ``jacobian``, ``hessian``,
``getattr(kernel, name)`` and ``set_param(name, val)`` (gp/gp.py:211-238,264-276).
Subclasses that set ``_native_kernel`` (a GPX_KERNEL_* id) get the fully
device-resident fit path; any other subclass is honoured through its own
Python ``K()`` evaluated on the host, with the matrix uploaded for the
factorisation (slow path, same results).
"""
from copy import copy

__all__ = ["Kernel"]


class Kernel(object):
    #: GPX_KERNEL_* id understood by libgpx, or None for a pure-Python plugin
    _native_kernel = None

    # pickling / copying carry the parameter vector only (base.py:9-20)
    def __getstate__(self):
        return {"params": self.params}

    def __setstate__(self, state):
        self.params = state["params"]

    def __copy__(self):
        return type(self)(*self.params)

    def __deepcopy__(self, memo):
        return type(self)(*self.params)

    def copy(self):
        """New kernel object of ``type(self)`` with the same parameters."""
        return copy(self)

    def sym_K(self):
        """Symbolic (sympy) form of the kernel function."""
        raise NotImplementedError

    @property
    def params(self):
        """Kernel parameters (numpy.ndarray)."""
        raise NotImplementedError

    def K(self, x1, x2, out=None):
        r"""Kernel function evaluated at `x1` (n,) and `x2` (m,): an (n, m) array."""
        raise NotImplementedError

    def __call__(self, x1, x2, out=None):
        return self.K(x1, x2, out=out)

    def jacobian(self, x1, x2, out=None):
        r"""(n_p, n, m) array of first parameter derivatives of the kernel matrix."""
        raise NotImplementedError

    def hessian(self, x1, x2, out=None):
        r"""(n_p, n_p, n, m) array of second parameter derivatives."""
        raise NotImplementedError
