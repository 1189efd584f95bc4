"""CPU restatement of the reference's GP fit/predict path (numpy/scipy + gp_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported only by tests/, by
__graft_entry__.smoke() and by bench.py's ``cpu_baseline`` leg -- as the checker
or as the timed CPU baseline, never from gaussian_processes_amd/ (the product
path fails loudly when its HIP library is missing; it has no CPU fallback).

Parity status: PINNED against golden vectors produced by the real reference
(oracle/make_golden.py -> tests/golden/*.npz; checked by tests/test_oracle.py).

Stage sequence restated (paths relative to /root/reference):
  kernel matrix      gp/ext/gaussian_c.pyx:18-36, gp/ext/periodic_c.pyx:18-30
  Kxx = K + I s^2    gp/gp.py:242-266
  Lxx                gp/gp.py:278-294   scipy.linalg.cholesky(lower=True)
  inv_Kxx_y          gp/gp.py:314-335   scipy.linalg.cho_solve
  inv_Kxx            gp/gp.py:296-312   inv(L).T @ inv(L)
  log_lh             gp/gp.py:337-367 + gp/ext/gp_c.pyx:17-31 (slogdet LU, MIN clamp)
  lh                 gp/gp.py:369-396
  mean / cov         gp/gp.py:574-625
  derivative stack   gp/ext/gp_c.pyx:34-131, gp/gp.py:398-502,627-662
The linear algebra itself lives outside the reference tree (SciPy/NumPy ->
LAPACK/BLAS, unpinned: requirements.txt lists numpy>=1.7.1 only); it is called
here exactly where the reference calls it.
"""
import ctypes
import os
import subprocess

import numpy as np
from scipy.linalg import cho_solve, cholesky

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgp_oracle.so")
_SRC = os.path.join(_HERE, "gp_oracle.c")

DTYPE = np.float64
EPS = np.finfo(DTYPE).eps
MIN = np.log(np.exp2(DTYPE(np.finfo(DTYPE).minexp + 4)))  # gp/gp.py:17

GAUSSIAN_WHICH = {"K": 0, "dK_dh": 1, "dK_dw": 2, "d2K_dhdh": 3, "d2K_dhdw": 4,
                  "d2K_dwdh": 4, "d2K_dwdw": 5}
PERIODIC_WHICH = {"K": 0, "dK_dh": 1, "dK_dw": 2, "dK_dp": 3, "d2K_dhdh": 4,
                  "d2K_dhdw": 5, "d2K_dwdh": 5, "d2K_dhdp": 6, "d2K_dpdh": 6,
                  "d2K_dwdw": 7, "d2K_dwdp": 8, "d2K_dpdw": 8, "d2K_dpdp": 9}


def build(force=False):
    """Compile gp_oracle.c -> libgp_oracle.so (gcc, single thread, no FMA contraction)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC",
                               _SRC, "-lm", "-o", _SO])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        dp = ctypes.POINTER(ctypes.c_double)
        L.oracle_gaussian.argtypes = [ctypes.c_int, dp, ctypes.c_long, dp, ctypes.c_long,
                                      dp, ctypes.c_long, ctypes.c_int, ctypes.c_double,
                                      ctypes.c_double]
        L.oracle_gaussian.restype = ctypes.c_int
        L.oracle_periodic.argtypes = [ctypes.c_int, dp, ctypes.c_long, dp, ctypes.c_long,
                                      dp, ctypes.c_long, ctypes.c_int, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_double]
        L.oracle_periodic.restype = ctypes.c_int
        L.oracle_add_diag.argtypes = [dp, ctypes.c_long, ctypes.c_long, ctypes.c_double]
        L.oracle_add_diag.restype = None
        L.oracle_potrf_lower.argtypes = [dp, ctypes.c_long, ctypes.c_long]
        L.oracle_potrf_lower.restype = ctypes.c_int
        L.oracle_potrs_lower.argtypes = [dp, ctypes.c_long, ctypes.c_long, dp]
        L.oracle_potrs_lower.restype = None
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _as2d(x):
    x = np.ascontiguousarray(x, dtype=DTYPE)
    if x.ndim == 1:
        x = x.reshape(-1, 1)
    return x


def kernel_matrix(kind, which, x1, x2, params, out=None):
    """out[i, j] = kernel-or-derivative(x1[i], x2[j]); x1: (n,) or (n, d)."""
    a, b = _as2d(x1), _as2d(x2)
    n, d = a.shape
    m = b.shape[0]
    assert b.shape[1] == d
    if out is None:
        out = np.empty((n, m), dtype=DTYPE)
    assert out.flags.c_contiguous and out.shape == (n, m)
    if kind == "gaussian":
        h, w = params
        rc = lib().oracle_gaussian(GAUSSIAN_WHICH[which], _p(out), m, _p(a), n, _p(b), m, d,
                                   float(h), float(w))
    elif kind == "periodic":
        h, w, p = params
        rc = lib().oracle_periodic(PERIODIC_WHICH[which], _p(out), m, _p(a), n, _p(b), m, d,
                                   float(h), float(w), float(p))
    else:
        raise ValueError(kind)
    if rc != 0:
        raise ValueError("oracle rc=%d" % rc)
    return out


def jacobian(kind, x1, x2, params):
    names = ["dK_dh", "dK_dw"] + (["dK_dp"] if kind == "periodic" else [])
    return np.stack([kernel_matrix(kind, nm, x1, x2, params) for nm in names])


def hessian(kind, x1, x2, params):
    ps = "hw" + ("p" if kind == "periodic" else "")
    return np.stack([np.stack([kernel_matrix(kind, "d2K_d%sd%s" % (a, b), x1, x2, params)
                               for b in ps]) for a in ps])


class OracleGP(object):
    """Stage-by-stage restatement of gp.GP for kind in {'gaussian','periodic'}."""

    def __init__(self, kind, kparams, x, y, s):
        self.kind = kind
        self.kparams = tuple(float(v) for v in kparams)
        self.x = np.array(x, dtype=DTYPE)
        self.y = np.array(y, dtype=DTYPE)
        self.s = DTYPE(s)
        self._m = {}

    def _memo(self, name, f):
        if name not in self._m:
            self._m[name] = f()
        return self._m[name]

    def K(self, a, b):
        return kernel_matrix(self.kind, "K", a, b, self.kparams)

    @property
    def n(self):
        return self.x.shape[0]

    @property
    def Kxx(self):  # gp/gp.py:263-266
        def f():
            K = self.K(self.x, self.x)
            lib().oracle_add_diag(_p(K), self.n, self.n, float(self.s))
            return K
        return self._memo("Kxx", f)

    @property
    def Lxx(self):  # gp/gp.py:294
        return self._memo("Lxx", lambda: cholesky(self.Kxx, lower=True, overwrite_a=False,
                                                  check_finite=True))

    @property
    def inv_Kxx(self):  # gp/gp.py:311-312
        def f():
            iL = np.linalg.inv(self.Lxx)
            return np.dot(iL.T, iL)
        return self._memo("inv_Kxx", f)

    @property
    def inv_Kxx_y(self):  # gp/gp.py:332-335
        return self._memo("inv_Kxx_y", lambda: cho_solve((self.Lxx, True), self.y,
                                                         overwrite_b=False, check_finite=True))

    @property
    def log_lh(self):  # gp/gp.py:360-367 + gp_c.pyx:17-31
        def f():
            K = self.Kxx
            try:
                Kiy = self.inv_Kxx_y
            except np.linalg.LinAlgError:
                return -np.inf
            sign, logdet = np.linalg.slogdet(K)
            if sign != 1 or logdet < MIN:
                return DTYPE(-np.inf)
            data_fit = -0.5 * DTYPE(np.dot(self.y, Kiy))
            complexity_penalty = -0.5 * logdet
            constant = -0.5 * self.y.size * np.log(2 * np.pi)
            return DTYPE(data_fit + complexity_penalty + constant)
        return self._memo("log_lh", f)

    @property
    def log_lh_chol(self):
        """'Fair' variant: logdet from diag(L) instead of the reference's second LU."""
        def f():
            try:
                L = self.Lxx
                Kiy = self.inv_Kxx_y
            except np.linalg.LinAlgError:
                return -np.inf
            logdet = 2.0 * np.sum(np.log(np.diag(L)))
            if logdet < MIN:
                return DTYPE(-np.inf)
            return DTYPE(-0.5 * np.dot(self.y, Kiy) - 0.5 * logdet
                         - 0.5 * self.y.size * np.log(2 * np.pi))
        return self._memo("log_lh_chol", f)

    @property
    def lh(self):  # gp/gp.py:392-396
        llh = self.log_lh
        return 0 if llh < MIN else np.exp(llh)

    def Kxoxo(self, xo):
        return self.K(xo, xo)

    def Kxxo(self, xo):
        return self.K(self.x, xo)

    def Kxox(self, xo):
        return self.K(xo, self.x)

    def mean(self, xo):  # gp/gp.py:597
        return np.dot(self.Kxox(xo), self.inv_Kxx_y)

    def cov(self, xo):  # gp/gp.py:622-625
        return self.Kxoxo(xo) - np.dot(self.Kxox(xo), np.dot(self.inv_Kxx, self.Kxxo(xo)))

    # ---- derivative stack (gp/ext/gp_c.pyx:34-131) ----
    @property
    def Kxx_J(self):
        return self._memo("Kxx_J", lambda: jacobian(self.kind, self.x, self.x, self.kparams))

    @property
    def Kxx_H(self):
        return self._memo("Kxx_H", lambda: hessian(self.kind, self.x, self.x, self.kparams))

    def _dK(self, i):
        Kj = self.Kxx_J
        return Kj[i] if i < Kj.shape[0] else np.eye(self.n) * 2 * self.s

    @property
    def dloglh_dtheta(self):  # gp_c.pyx:34-49
        Ki, Kiy, y = self.inv_Kxx, self.inv_Kxx_y, self.y
        npar = self.Kxx_J.shape[0]
        out = np.empty(npar + 1)
        for i in range(npar + 1):
            k = np.dot(Ki, self._dK(i))
            out[i] = 0.5 * np.dot(y, np.dot(k, Kiy)) + -0.5 * np.trace(k)
        return out

    @property
    def dlh_dtheta(self):  # gp_c.pyx:52-67
        Ki, Kiy, y, lh = self.inv_Kxx, self.inv_Kxx_y, self.y, self.lh
        npar = self.Kxx_J.shape[0]
        out = np.empty(npar + 1)
        for i in range(npar + 1):
            k = np.dot(Ki, self._dK(i))
            out[i] = 0.5 * lh * (np.dot(y, np.dot(k, Kiy)) - np.trace(k))
        return out

    @property
    def d2lh_dtheta2(self):  # gp_c.pyx:70-111
        Ki, Kiy, y, lh = self.inv_Kxx, self.inv_Kxx_y, self.y, self.lh
        Kh, dlh = self.Kxx_H, self.dlh_dtheta
        n = self.Kxx_J.shape[0]
        m = self.n
        dK = [self._dK(i) for i in range(n + 1)]
        dKi = [np.dot(-Ki, np.dot(dK[i], Ki)) for i in range(n + 1)]
        out = np.empty((n + 1, n + 1))
        for i in range(n + 1):
            KidK_i = np.dot(Ki, dK[i])
            ydKi_iy_tr = np.dot(y, np.dot(KidK_i, Kiy)) - np.trace(KidK_i)
            for j in range(n + 1):
                if j < n and i < n:
                    d2k = Kh[i, j]
                elif j == n and i == n:
                    d2k = np.eye(m) * 2
                else:
                    d2k = np.zeros((m, m))
                dKi_jdK_i = np.dot(dKi[j], dK[i])
                t0 = dlh[j] * ydKi_iy_tr
                t1a = np.dot(y, np.dot(dKi_jdK_i, Kiy))
                t1b = np.dot(Kiy, np.dot(d2k, Kiy))
                t1c = np.dot(Kiy, np.dot(dK[i], np.dot(dKi[j], y)))
                t1 = lh * (t1a + t1b + t1c - np.trace(dKi_jdK_i + np.dot(Ki, d2k)))
                out[i, j] = 0.5 * (t0 + t1)
        return out

    def dm_dtheta(self, xo):  # gp_c.pyx:114-131 + gp/gp.py:652-662
        Ki, y = self.inv_Kxx, self.y
        Kj = self.Kxx_J
        Kjxo = jacobian(self.kind, xo, self.x, self.kparams)
        Kxox = self.Kxox(xo)
        n = Kj.shape[0]
        dm = np.empty((n + 1, np.asarray(xo).shape[0]))
        for i in range(n + 1):
            if i < n:
                dKxox, dKxx = Kjxo[i], Kj[i]
            else:
                dKxox, dKxx = np.zeros_like(Kxox), np.eye(self.n) * 2 * self.s
            dm[i] = np.dot(dKxox, np.dot(Ki, y))
            dm[i] -= np.dot(Kxox, np.dot(np.dot(Ki, np.dot(dKxx, Ki)), y))
        return dm


def synth_inputs(N, d, m, seed=0):
    """SURVEY section 8(d) synthetic workload (shared by tests and bench.py)."""
    rng = np.random.RandomState(seed)
    X = rng.uniform(-10, 10, (N, d))
    if d == 1:
        X = np.sort(X.ravel()).reshape(N, 1)
    y = np.sin(X.sum(1) / np.sqrt(d)) + 0.1 * rng.randn(N)
    Xo = np.random.RandomState(seed + 1).uniform(-10, 10, (m, d))
    return X, y, Xo
