"""GP -- drop-in for the reference's ``gp.GP`` (gp/gp.py:44-700) on MI355X.

Same constructor, properties, methods, error behaviour and memoisation rules as
the reference; the arithmetic of the hot path -- kernel-matrix build, Cholesky,
triangular solves, log-determinant, posterior mean / covariance -- runs in
hand-written HIP kernels behind the C ABI of libgpx.so.  The fitted state
(x, y, the factor, alpha) lives in HBM inside a ``gpx_gp`` handle; host copies
of the big matrices (`Kxx`, `Lxx`, `inv_Kxx`) are only made when those
properties are actually read, so `log_lh`, `inv_Kxx_y`, `mean`, `cov` work at
sizes whose n x n matrices would not fit host memory comfortably.

Extensions beyond the reference (which is strictly 1-D, float64):
  * ``x`` may be (n, d); ``y`` must then be (n,).
  * ``dtype='float32'`` runs the device path in fp32 (results are still
    returned as float64 arrays / numpy.float64 scalars).
  * ``device=k`` selects the GPU.
"""
import ctypes
import logging
from copy import copy, deepcopy

import numpy as np

from . import _lib
from .ext import gp_c

__all__ = ["GP"]

logger = logging.getLogger("gp.gp")

DTYPE = np.float64
EPS = np.finfo(DTYPE).eps
MIN = np.log(np.exp2(DTYPE(np.finfo(DTYPE).minexp + 4)))   # gp/gp.py:17

_DTYPES = {"float64": _lib.F64, "f64": _lib.F64, np.float64: _lib.F64,
           "float32": _lib.F32, "f32": _lib.F32, np.float32: _lib.F32}


def memoprop(f):
    """Memoised property: computed on first access, cached in ``self._memoized``
    under the property's name, evicted by ``del obj.prop`` (gp/gp.py:20-41)."""
    name = f.__name__

    def fget(self):
        cache = self._memoized
        if name not in cache:
            cache[name] = f(self)
        return cache[name]

    def fdel(self):
        del self._memoized[name]

    return property(fget=fget, fdel=fdel, doc=f.__doc__)


class _DeviceState(object):
    """Owns the gpx_gp handle of one GP; rebuilt whenever shape/dtype change."""

    def __init__(self, dtype_id, kernel_id, n, d, device):
        self.key = (dtype_id, kernel_id, n, d, device)
        self.handle = ctypes.c_void_p()
        self.data_version = -1
        self.params_version = -1
        self.fit_version = -1
        self.info = None
        lib = _lib.load()
        if device is not None:
            _lib.check(lib.gpx_set_device(int(device)))
        _lib.check(lib.gpx_gp_create(ctypes.byref(self.handle), dtype_id, kernel_id, n, d))

    @classmethod
    def adopt(cls, handle, key):
        """Wrap an existing fitted gpx_gp handle (gpx_gp_load)."""
        st = cls.__new__(cls)
        st.key = key
        st.handle = handle
        st.data_version = st.params_version = st.fit_version = -1
        st.info = None
        return st

    def close(self):
        if self.handle:
            _lib.load().gpx_gp_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown
            pass


class GP(object):
    r"""Gaussian process regression object.

    Parameters
    ----------
    K : :class:`~gaussian_processes_amd.kernels.Kernel`
        Kernel object
    x : numpy.ndarray
        :math:`n` array of input locations (or ``(n, d)``)
    y : numpy.ndarray
        :math:`n` array of observations
    s : number (default=0)
        Standard deviation of the observation noise
    """

    def __init__(self, K, x, y, s=0, dtype="float64", device=None):
        self.K = K
        self._x = None
        self._y = None
        self._s = None
        self._memoized = {}
        self._dtype = _DTYPES[dtype]
        self._device = device
        self._dev = None          # _DeviceState, created lazily
        self._version = 0         # bumped by every invalidation
        self._data_version = 0    # bumped when x or y change

        self.x = x
        self.y = y
        self.s = s

    # ---- state: exactly the reference's five entries (gp/gp.py:78-92) ----
    def __getstate__(self):
        state = {"K": self.K, "_x": self._x, "_y": self._y, "_s": self._s,
                 "_memoized": self._memoized}
        # the two extensions ride along only when they are in use: a GP built the reference's way
        # (float64, default device) pickles to exactly the reference's five keys
        if self._dtype != _lib.F64 or self._device is not None:
            state["_gpx"] = {"dtype": self._dtype, "device": self._device}
        return state

    def __setstate__(self, state):
        self.K = state["K"]
        self._x = state["_x"]
        self._y = state["_y"]
        self._s = state["_s"]
        self._memoized = state["_memoized"]
        extra = state.get("_gpx", {})
        self._dtype = extra.get("dtype", _lib.F64)
        self._device = extra.get("device", None)
        self._dev = None
        self._version = 0
        self._data_version = 0

    def __copy__(self):
        new = type(self).__new__(type(self))
        new.__setstate__(self.__getstate__())
        return new

    def __deepcopy__(self, memo):
        new = type(self).__new__(type(self))
        new.__setstate__(deepcopy(self.__getstate__(), memo))
        return new

    def copy(self, deep=True):
        """Deep (default) or shallow copy of the GP object."""
        return deepcopy(self) if deep else copy(self)

    # ---- invalidation ----
    def _invalidate(self, data=False):
        self._memoized = {}
        self._version += 1
        if data:
            self._data_version += 1

    # ---- inputs (gp/gp.py:129-197) ----
    @property
    def x(self):
        r"""Input locations, read-only float64 copy."""
        return self._x

    @x.setter
    def x(self, val):
        if np.any(val != self._x):
            arr = np.array(val, copy=True, dtype=DTYPE)
            if arr.ndim not in (1, 2):
                raise ValueError("invalid shape for x: %s" % str(arr.shape))
            self._invalidate(data=True)
            self._x = arr
            self._x.flags.writeable = False

    @property
    def y(self):
        r"""Observations, read-only float64 copy."""
        return self._y

    @y.setter
    def y(self, val):
        if np.any(val != self._y):
            self._invalidate(data=True)
            self._y = np.array(val, copy=True, dtype=DTYPE)
            self._y.flags.writeable = False
            expected = self._x.shape if self._x.ndim == 1 else (self._x.shape[0],)
            if self._y.shape != expected:
                raise ValueError("invalid shape for y: %s" % str(self._y.shape))

    @property
    def s(self):
        r"""Standard deviation of the observation noise (numpy.float64)."""
        return self._s

    @s.setter
    def s(self, val):
        if val < 0:
            raise ValueError("invalid value for s: %s" % val)
        if val != self._s:
            self._invalidate()
            self._s = DTYPE(val)

    @property
    def params(self):
        r"""``(kernel parameters..., s)`` (gp/gp.py:199-214)."""
        kp = self.K.params
        out = np.empty(kp.size + 1)
        out[:-1] = kp
        out[-1] = self._s
        return out

    @params.setter
    def params(self, val):
        if np.any(self.params != val):
            self._invalidate()
            self.K.params = val[:-1]
            self.s = val[-1]

    def get_param(self, name):
        return self.s if name == "s" else getattr(self.K, name)

    def set_param(self, name, val):
        if name == "s":
            self.s = val
            return
        if getattr(self.K, name) != val:      # AttributeError for an unknown name
            self._invalidate()
            self.K.set_param(name, val)

    # ---- device plumbing ----
    @property
    def _n(self):
        return self._x.shape[0]

    @property
    def _d(self):
        return 1 if self._x.ndim == 1 else self._x.shape[1]

    def _state(self):
        kid = getattr(self.K, "_native_kernel", None)
        key = (self._dtype, _lib.KERNEL_GAUSSIAN if kid is None else kid, self._n, self._d,
               self._device)
        if self._dev is None or self._dev.key != key:
            if self._dev is not None:
                self._dev.close()
            self._dev = _DeviceState(*key)
        st = self._dev
        if st.data_version != self._data_version:
            xs = np.ascontiguousarray(self._x, dtype=DTYPE)
            ys = np.ascontiguousarray(self._y, dtype=DTYPE)
            _lib.check(_lib.load().gpx_gp_set_data(st.handle, _lib.dptr(xs), _lib.dptr(ys)))
            st.data_version = self._data_version
            st.params_version = -1
            st.fit_version = -1
        return st

    def _sync_params(self, st):
        """Push (kernel params, s) to the handle once per invalidation."""
        if st.params_version != self._version:
            p = np.ascontiguousarray(self.K.params, dtype=DTYPE)
            _lib.check(_lib.load().gpx_gp_set_params(st.handle, _lib.dptr(p), float(self._s)))
            st.params_version = self._version
            st.fit_version = -1

    def _fit(self):
        """Make the device state current: Kxx -> Lxx -> alpha -> logdet (one pass)."""
        st = self._state()
        if st.fit_version == self._version:
            return st
        lib = _lib.load()
        native = getattr(self.K, "_native_kernel", None) is not None
        if native:
            self._sync_params(st)
        else:
            # plugin kernel: its own Python K() on the host, matrix uploaded
            Kxx = np.ascontiguousarray(self.Kxx, dtype=DTYPE)
            _check_finite(Kxx)
            _lib.check(lib.gpx_gp_set_K(st.handle, _lib.dptr(Kxx), Kxx.shape[1]))
        info = ctypes.c_int(0)
        _lib.check(lib.gpx_gp_fit(st.handle, ctypes.byref(info)))
        st.info = info.value
        st.fit_version = self._version
        return st

    def _fit_pd(self):
        st = self._fit()
        if st.info != 0:
            raise _lib.lapack_info_error(st.info)
        return st

    # ---- memoised hot-path properties ----
    @memoprop
    def Kxx(self):
        r"""Kernel covariance matrix :math:`K(x, x) + s^2 I` (gp/gp.py:242-266)."""
        if getattr(self.K, "_native_kernel", None) is None:
            K = self.K(self._x, self._x)
            K[np.diag_indices_from(K)] += self._s ** 2
            return K
        st = self._state()
        self._sync_params(st)
        out = np.empty((self._n, self._n), dtype=DTYPE)
        _lib.check(_lib.load().gpx_gp_get_Kxx(st.handle, _lib.dptr(out), self._n))
        return out

    @memoprop
    def Kxx_J(self):
        return self.K.jacobian(self._x, self._x)

    @memoprop
    def Kxx_H(self):
        return self.K.hessian(self._x, self._x)

    @memoprop
    def Lxx(self):
        r"""Lower Cholesky factor of `Kxx` (gp/gp.py:278-294); raises
        numpy.linalg.LinAlgError when `Kxx` is not positive definite."""
        st = self._fit_pd()
        out = np.empty((self._n, self._n), dtype=DTYPE)
        _lib.check(_lib.load().gpx_gp_get_Lxx(st.handle, _lib.dptr(out), self._n))
        return out

    @memoprop
    def inv_Kxx(self):
        r"""Explicit inverse :math:`K_{xx}^{-1} = L^{-\top} L^{-1}` (gp/gp.py:296-312)."""
        st = self._fit_pd()
        out = np.empty((self._n, self._n), dtype=DTYPE)
        _lib.check(_lib.load().gpx_gp_get_inv_Kxx(st.handle, _lib.dptr(out), self._n))
        return out

    @memoprop
    def inv_Kxx_y(self):
        r""":math:`K_{xx}^{-1} y` by forward + back substitution (gp/gp.py:314-335)."""
        st = self._fit_pd()
        out = np.empty(self._n, dtype=DTYPE)
        _lib.check(_lib.load().gpx_gp_get_alpha(st.handle, _lib.dptr(out)))
        return out

    @memoprop
    def log_lh(self):
        r"""Log marginal likelihood, RW06 eq. 5.8 (gp/gp.py:337-367, gp_c.pyx:17-31):
        ``-inf`` when `Kxx` is not positive definite or ``logdet < MIN``."""
        st = self._fit()
        if st.info != 0:
            return -np.inf
        out = ctypes.c_double(0.0)
        _lib.check(_lib.load().gpx_gp_log_lh(st.handle, ctypes.byref(out)))
        return DTYPE(out.value)

    @memoprop
    def lh(self):
        r"""Marginal likelihood ``exp(log_lh)``, 0 below ``MIN`` (gp/gp.py:369-396)."""
        llh = self.log_lh
        if llh < MIN:
            return 0
        return np.exp(llh)

    # ---- derivative stack (gp/gp.py:398-502): formulas of gp_c.pyx, products on device ----
    def _nan_or(self, shape):
        out = np.empty(shape)
        try:
            Ki = self.inv_Kxx
        except np.linalg.LinAlgError:
            out.fill(np.nan)
            return out, None
        return out, Ki

    @memoprop
    def dloglh_dtheta(self):
        r"""Gradient of the log marginal likelihood, RW06 eq. 5.9 (gp/gp.py:398-433).  Native
        kernels: computed entirely on the device (K^-1 never leaves HBM, the kernel Jacobian
        is never materialised); plugin kernels: the reference's formula with device products."""
        npar = len(self.params)
        native = getattr(self.K, "_native_kernel", None) is not None
        if native and (self._d == 1 or self.K._native_kernel == _lib.KERNEL_GAUSSIAN):
            st = self._fit()
            out = np.empty(npar, dtype=DTYPE)
            _lib.check(_lib.load().gpx_gp_dloglh_dtheta(st.handle, _lib.dptr(out)))
            return out
        out, Ki = self._nan_or(npar)
        if Ki is not None:
            gp_c.dloglh_dtheta(_c(self._y), Ki, _c(self.Kxx_J), self.inv_Kxx_y, self._s, out)
        return out

    def _native_derivs(self):
        """True when the derivative stack can stay on the device (built-in kernel; periodic: 1-D)."""
        kid = getattr(self.K, "_native_kernel", None)
        return kid is not None and (self._d == 1 or kid == _lib.KERNEL_GAUSSIAN)

    @memoprop
    def dlh_dtheta(self):
        r"""Gradient of the marginal likelihood (gp/gp.py:435-465).  Native kernels: device resident."""
        if self._native_derivs():
            # gp_c.pyx:52-67 is lh times gp_c.pyx:34-49 term by term: 0.5 lh (y^T K^-1 dK K^-1 y - tr(K^-1 dK))
            return np.asarray(self.lh * self.dloglh_dtheta, dtype=DTYPE)
        out, Ki = self._nan_or(len(self.params))
        if Ki is not None:
            gp_c.dlh_dtheta(_c(self._y), Ki, _c(self.Kxx_J), self.inv_Kxx_y, self._s, self.lh, out)
        return out

    @memoprop
    def d2lh_dtheta2(self):
        r"""Hessian of the marginal likelihood (gp/gp.py:467-502).  Native kernels: device resident
        (K^-1, the K^-1 dK_i products and all traces / quadratic forms stay in HBM)."""
        if self._native_derivs():
            st = self._fit()
            npar = len(self.params)
            out = np.empty((npar, npar), dtype=DTYPE)
            hess = np.empty((npar, npar), dtype=DTYPE)
            _lib.check(_lib.load().gpx_gp_dlh_d2lh(st.handle, None, _lib.dptr(out), _lib.dptr(hess)))
            self._memoized.setdefault("d2loglh_dtheta2", hess)       # same device pass
            return out
        npar = len(self.params)
        out, Ki = self._nan_or((npar, npar))
        if Ki is not None:
            gp_c.d2lh_dtheta2(_c(self._y), Ki, _c(self.Kxx_J), _c(self.Kxx_H), self.inv_Kxx_y, self._s,
                              self.lh, _c(self.dlh_dtheta), out)
        return out

    @memoprop
    def d2loglh_dtheta2(self):
        r"""Hessian of the LOG marginal likelihood w.r.t. ``(kernel params..., s)`` -- an extension:
        the reference only offers the lh-scaled `d2lh_dtheta2`, which is identically zero once
        ``log_lh < MIN`` (any n beyond a few hundred).  ``d2lh / lh - (dlh / lh)(dlh / lh)^T`` from the
        same device pass (csrc/gpx_deriv.hip); native kernels only.  NaN when `Kxx` is not PD."""
        if not self._native_derivs():
            raise NotImplementedError("d2loglh_dtheta2 needs a built-in kernel (periodic: 1-D inputs)")
        st = self._fit()
        npar = len(self.params)
        out = np.empty((npar, npar), dtype=DTYPE)
        _lib.check(_lib.load().gpx_gp_dlh_d2lh(st.handle, None, None, _lib.dptr(out)))
        return out

    # ---- prediction (gp/gp.py:504-662) ----
    def Kxoxo(self, xo):
        r""":math:`K(x^*, x^*)`, ``(m, m)``."""
        return self.K(xo, xo)

    def Kxxo(self, xo):
        r""":math:`K(x, x^*)`, ``(n, m)``."""
        return self.K(self._x, xo)

    def Kxox(self, xo):
        r""":math:`K(x^*, x)`, ``(m, n)``."""
        return self.K(xo, self._x)

    def _xo(self, xo):
        xo = np.ascontiguousarray(xo, dtype=DTYPE)
        d = 1 if xo.ndim == 1 else xo.shape[1]
        if xo.ndim not in (1, 2) or d != self._d:
            raise ValueError("invalid shape for xo: %s" % str(xo.shape))
        return xo, xo.shape[0]

    def mean(self, xo):
        r"""Predictive mean :math:`K(x^*, x) K_{xx}^{-1} y`, RW06 eq. 2.23 (gp/gp.py:574-597).
        Fused on the device: the ``(m, n)`` cross-kernel matrix is never materialised."""
        st = self._fit_pd()
        xo, m = self._xo(xo)
        out = np.empty(m, dtype=DTYPE)
        lib = _lib.load()
        if getattr(self.K, "_native_kernel", None) is not None:
            _lib.check(lib.gpx_gp_mean(st.handle, _lib.dptr(xo), m, _lib.dptr(out)))
        else:
            Kxox = np.ascontiguousarray(self.Kxox(xo), dtype=DTYPE)
            _lib.check(lib.gpx_gp_mean_from_K(st.handle, _lib.dptr(Kxox), m, _lib.dptr(out)))
        return out

    def cov(self, xo):
        r"""Predictive covariance :math:`K(x^*,x^*) - K(x^*,x) K_{xx}^{-1} K(x,x^*)`, RW06
        eq. 2.24 (gp/gp.py:599-625), computed as :math:`K(x^*,x^*) - V^\top V` with
        :math:`V = L^{-1} K(x, x^*)` (no explicit inverse)."""
        st = self._fit_pd()
        xo, m = self._xo(xo)
        out = np.empty((m, m), dtype=DTYPE)
        lib = _lib.load()
        if getattr(self.K, "_native_kernel", None) is not None:
            _lib.check(lib.gpx_gp_cov(st.handle, _lib.dptr(xo), m, _lib.dptr(out)))
        else:
            Kxox = np.ascontiguousarray(self.Kxox(xo), dtype=DTYPE)
            Kxoxo = np.ascontiguousarray(self.Kxoxo(xo), dtype=DTYPE)
            _lib.check(lib.gpx_gp_cov_from_K(st.handle, _lib.dptr(Kxox), _lib.dptr(Kxoxo), m,
                                             _lib.dptr(out)))
        return out

    def dm_dtheta(self, xo):
        r"""Derivative of the predictive mean w.r.t. the parameters, ``(n_p, m)``
        (gp/gp.py:627-662, gp_c.pyx:114-131)."""
        if self._native_derivs():
            st = self._fit_pd()                      # LinAlgError when not PD, as inv_Kxx raises in the reference
            xo, m = self._xo(xo)
            dm = np.empty((len(self.params), m), dtype=DTYPE)
            _lib.check(_lib.load().gpx_gp_dm_dtheta(st.handle, _lib.dptr(xo), m, _lib.dptr(dm)))
            return dm
        Ki = self.inv_Kxx
        Kj = _c(self.Kxx_J)
        Kjxo = _c(self.K.jacobian(xo, self._x))
        Kxox = _c(self.Kxox(xo))
        dm = np.empty((len(self.params), Kxox.shape[0]))
        gp_c.dm_dtheta(_c(self._y), Ki, Kj, Kjxo, Kxox, self._s, dm)
        return dm

    # ---- persistence of the DEVICE state (extension; the reference pickles host arrays, gp/gp.py:78-92) ----
    def save_fitted(self, path):
        """Checkpoint the fitted device state -- x, y, alpha and the Cholesky factor -- to `path`,
        streamed from HBM in row blocks (host memory use does not grow with n; a 32 GiB factor never
        needs a host copy).  Restore with `GP.load_fitted`."""
        st = self._fit()
        _lib.check(_lib.load().gpx_gp_save(st.handle, str(path).encode()))

    @classmethod
    def load_fitted(cls, path, K=None, device=None):
        """A GP restored from `save_fitted`: the factor goes straight back into HBM and nothing is
        recomputed (`log_lh`, `mean`, `cov`, ... are served from the loaded state).  `K`: the kernel
        object for a plugin kernel; built-in kernels are rebuilt from the file."""
        lib = _lib.load()
        if device is not None:
            _lib.check(lib.gpx_set_device(int(device)))
        h = ctypes.c_void_p()
        _lib.check(lib.gpx_gp_load(ctypes.byref(h), str(path).encode()))
        try:
            dt, kid, n, d = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64(), ctypes.c_int()
            prm = np.zeros(3)
            s = ctypes.c_double()
            _lib.check(lib.gpx_gp_describe(h, ctypes.byref(dt), ctypes.byref(kid), ctypes.byref(n), ctypes.byref(d),
                                           _lib.dptr(prm), ctypes.byref(s)))
            x = np.empty((n.value, d.value), dtype=DTYPE)
            y = np.empty(n.value, dtype=DTYPE)
            _lib.check(lib.gpx_gp_get_xy(h, _lib.dptr(x), _lib.dptr(y)))
            if K is None:
                from .kernels import GaussianKernel, PeriodicKernel
                K = GaussianKernel(*prm[:2]) if kid.value == _lib.KERNEL_GAUSSIAN else PeriodicKernel(*prm[:3])
            obj = cls(K, x.ravel() if d.value == 1 else x, y, s=s.value,
                      dtype="float64" if dt.value == _lib.F64 else "float32", device=device)
            key = (obj._dtype, kid.value, obj._n, obj._d, obj._device)
            st = _DeviceState.adopt(h, key)
        except Exception:
            lib.gpx_gp_destroy(h)
            raise
        info = ctypes.c_int(0)
        _lib.check(lib.gpx_gp_info(st.handle, ctypes.byref(info)))
        st.info = info.value
        st.data_version = obj._data_version
        st.params_version = st.fit_version = obj._version
        obj._dev = st
        return obj

    def fit_timing(self):
        """Milliseconds of the last device fit: kernel build, potrf, solve, reductions, total
        (HIP events on the handle's stream)."""
        st = self._fit()
        ms = (ctypes.c_float * 5)()
        _lib.check(_lib.load().gpx_gp_last_timing(st.handle, ms))
        return dict(zip(("kernel_build", "potrf", "solve", "reduce", "total"), list(ms)))

    def plot(self, ax=None, xlim=None, color="k", markercolor="r"):
        """Plot the predictive mean +/- one standard deviation (gp/gp.py:664-700)."""
        import matplotlib.pyplot as plt
        x, y = self._x, self._y
        if ax is None:
            ax = plt.gca()
        if xlim is None:
            xlim = (x.min(), x.max())
        X = np.linspace(xlim[0], xlim[1], 1000)
        mean = self.mean(X)
        std = np.sqrt(np.diag(self.cov(X)))
        ax.fill_between(X, mean - std, mean + std, color=color, alpha=0.3)
        ax.plot(X, mean, lw=2, color=color)
        ax.plot(x, y, "o", ms=5, color=markercolor)
        ax.set_xlim(*xlim)


def _c(a):
    """float64 C-contiguous view / copy (what the ext modules' buffer arguments require)."""
    return np.ascontiguousarray(a, dtype=DTYPE)


def _check_finite(a):
    # scipy.linalg.cholesky(..., check_finite=True) (gp/gp.py:294)
    if not np.isfinite(a).all():
        raise ValueError("array must not contain infs or NaNs")
