"""Where does a repeat fit differ?  The neighbour phase of tests/test_gpu_round4.py::test_soak_repeat_fits_are_bitwise_identical
as a long loop: fit n x n beside another handle factoring n = 3000 on a second host thread; when log_lh or alpha differ from the
first fit's, download the factor and report which entries differ (first row / column, count, size), then carry on.

    python tools/r6_soak_probe.py [n=8192] [dtype=float32] [reps=2000] [lib=path]
"""
import ctypes
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib          # noqa: E402

args = dict(a.split("=", 1) for a in sys.argv[1:])
n, dtype, reps = int(args.get("n", 8192)), args.get("dtype", "float32"), int(args.get("reps", 2000))
nn = int(args.get("neighbour", 3000))
if "lib" in args:                                   # another build (an older one lacks symbols _lib.load() binds): raw prototypes
    lib = ctypes.CDLL(os.path.abspath(args["lib"]))
    P, D, I64 = ctypes.c_void_p, ctypes.c_double, ctypes.c_int64
    for name, argt in (("gpx_gp_create", [ctypes.POINTER(P), ctypes.c_int, ctypes.c_int, I64, ctypes.c_int]),
                       ("gpx_gp_set_data", [P, P, P]), ("gpx_gp_set_params", [P, P, D]),
                       ("gpx_gp_fit", [P, ctypes.POINTER(ctypes.c_int)]), ("gpx_gp_log_lh", [P, ctypes.POINTER(D)]),
                       ("gpx_gp_get_alpha", [P, P]), ("gpx_gp_get_Lxx", [P, P, I64])):
        getattr(lib, name).argtypes = argt
        getattr(lib, name).restype = ctypes.c_int
else:
    lib = _lib.load()
d = 3
dt = _lib.F64 if dtype == "float64" else _lib.F32


def synth_inputs(N, d, seed=0):
    """The synthetic workload of the tests (same seeds, same draws)."""
    rng = np.random.RandomState(seed)
    X = rng.uniform(-10, 10, (N, d))
    y = np.sin(X.sum(1) / np.sqrt(d)) + 0.1 * rng.randn(N)
    return X, y


class Fit(object):
    def __init__(self, X, y):
        self.n = X.shape[0]
        self.h = ctypes.c_void_p()
        _lib.check(lib.gpx_gp_create(ctypes.byref(self.h), dt, _lib.KERNEL_GAUSSIAN, X.shape[0], X.shape[1]))
        xs, ys = np.ascontiguousarray(X, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
        _lib.check(lib.gpx_gp_set_data(self.h, _lib.dptr(xs), _lib.dptr(ys)))

    def __call__(self, params, s):
        p = np.ascontiguousarray(params, dtype=np.float64)
        _lib.check(lib.gpx_gp_set_params(self.h, _lib.dptr(p), float(s)))
        info = ctypes.c_int(0)
        _lib.check(lib.gpx_gp_fit(self.h, ctypes.byref(info)))
        out = ctypes.c_double(0.0)
        _lib.check(lib.gpx_gp_log_lh(self.h, ctypes.byref(out)))
        a = np.empty(self.n)
        _lib.check(lib.gpx_gp_get_alpha(self.h, _lib.dptr(a)))
        return out.value, a

    def factor(self):
        L = np.empty((self.n, self.n))
        _lib.check(lib.gpx_gp_get_Lxx(self.h, _lib.dptr(L), self.n))
        return L


X, y = synth_inputs(n, d)
params, s = np.array([1.0, 0.5 * np.sqrt(d)]), 0.9
fit = Fit(X, y)
l0, a0 = fit(params, s)
L0 = fit.factor()
for _ in range(3):
    l1, a1 = fit(params, s)
    assert l1 == l0 and np.array_equal(a1, a0), "differs before the neighbour started"
stop = threading.Event()
nb_err, nb_fits = [], [0]


def neighbour():
    Xn, yn = synth_inputs(nn, d, seed=5)
    other = Fit(Xn, yn)
    b0, c0 = other(params, 1.1)
    while not stop.is_set():
        b1, c1 = other(params, 1.1)
        nb_fits[0] += 1
        if b1 != b0 or not np.array_equal(c1, c0):
            nb_err.append((nb_fits[0], b1, b0, int((c1 != c0).sum())))


t = threading.Thread(target=neighbour)
if nn > 0:
    t.start()
bad = 0
t0 = time.time()
for rep in range(reps):
    l1, a1 = fit(params, s)
    if l1 != l0 or not np.array_equal(a1, a0):
        bad += 1
        L1 = fit.factor()
        diff = np.tril(L1 != L0)
        rows, cols = np.nonzero(diff)
        print("rep %d: log_lh %r vs %r; alpha differs in %d entries (first %s); factor differs in %d entries" % (
            rep, l1, l0, int((a1 != a0).sum()), (np.nonzero(a1 != a0)[0][:1].tolist()), int(diff.sum())), flush=True)
        if rows.size:
            k = np.lexsort((rows, cols))[0]                  # first in column order: the step it came from
            print("   first differing column %d (row %d there), first differing row %d (col %d there); columns %d..%d rows %d..%d; "
                  "max |dL| %.3e at L %.3e" % (cols[k], rows[k], rows[0], cols[0], cols.min(), cols.max(), rows.min(), rows.max(),
                                               float(np.abs(L1 - L0)[diff].max()), float(np.abs(L0[rows[k], cols[k]]))), flush=True)
            c = cols[k]
            rr = rows[cols == c]
            first256 = cols // 256 == c // 256                # the block column it started in: which rows, from which column on
            for r_ in np.unique(rows[first256])[:8]:
                cc = cols[first256 & (rows == r_)]
                print("   row %d (%% 64 = %d): columns %d..%d (%d); L0 %.9e L1 %.9e at the first" % (
                    r_, r_ % 64, cc.min(), cc.max(), cc.size, L0[r_, cc.min()], L1[r_, cc.min()]), flush=True)
            print("   in that column: %d rows differ, %d..%d; by 256-col block: %s" % (
                rr.size, rr.min(), rr.max(), np.bincount(cols // 256, minlength=n // 256 + 1).tolist()), flush=True)
        if bad >= 5:
            break
    if rep % 500 == 499:
        print("rep %d: %d bad so far, neighbour fits %d, %.1f s" % (rep, bad, nb_fits[0], time.time() - t0), flush=True)
stop.set()
if nn > 0:
    t.join(120)
print("n=%d %s: %d of %d fits differed; neighbour: %d fits, %d of its own differed %s" % (
    n, dtype, bad, rep + 1, nb_fits[0], len(nb_err), nb_err[:3]), flush=True)
