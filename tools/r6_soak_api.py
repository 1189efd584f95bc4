"""Repeat-run soak of the whole gp.GP API beside a neighbour: log_lh, inv_Kxx_y, inv_Kxx, mean, cov, dloglh_dtheta, dlh_dtheta,
d2lh_dtheta2, dm_dtheta of one GP, recomputed `reps` times (a changed and restored s drops the memoised values), every result
equal to the first bit for bit, while a second host thread factors in a loop.

    python tools/r6_soak_api.py [n=2048] [reps=100] [dtype=float64] [kernel=gaussian]
"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_processes_amd as gp              # noqa: E402

args = dict(a.split("=", 1) for a in sys.argv[1:])
n, reps, dtype = int(args.get("n", 2048)), int(args.get("reps", 100)), args.get("dtype", "float64")
kern = args.get("kernel", "gaussian")
d = 1 if kern == "periodic" else 3
rng = np.random.RandomState(0)
X = rng.uniform(-10, 10, (n, d))
if d == 1:
    X = X.ravel()
y = np.sin(np.atleast_2d(X.T).sum(0) / np.sqrt(d)) + 0.1 * rng.randn(n)
xo = rng.uniform(-10, 10, (37, d))
if d == 1:
    xo = xo.ravel()
K = gp.PeriodicKernel(1.1, 0.8, 2.3) if kern == "periodic" else gp.GaussianKernel(1.0, 0.5 * np.sqrt(d))
g = gp.GP(K, X, y, s=0.9, dtype=dtype)
members = [("log_lh", lambda: np.float64(g.log_lh)), ("inv_Kxx_y", lambda: g.inv_Kxx_y), ("inv_Kxx", lambda: g.inv_Kxx),
           ("mean", lambda: g.mean(xo)), ("cov", lambda: g.cov(xo)), ("dloglh", lambda: g.dloglh_dtheta),
           ("dlh", lambda: g.dlh_dtheta), ("d2lh", lambda: g.d2lh_dtheta2), ("dm", lambda: g.dm_dtheta(xo))]


def snapshot():
    g.set_param("s", 0.91)
    _ = g.log_lh
    g.set_param("s", 0.9)
    return [np.array(f(), copy=True) for _, f in members]


stop = threading.Event()
nb = [0, 0]


def neighbour():
    r2 = np.random.RandomState(5)
    Xn = r2.uniform(-10, 10, (3000, 3))
    yn = np.sin(Xn.sum(1)) + 0.1 * r2.randn(3000)
    h = gp.GP(gp.GaussianKernel(1.0, 0.9), Xn, yn, s=1.1, dtype=dtype)
    first = {}
    while not stop.is_set():
        for sv in (1.1, 1.2):
            h.set_param("s", sv)
            v = float(h.log_lh)
            nb[0] += 1
            if first.setdefault(sv, v) != v:
                nb[1] += 1


first = snapshot()
t = threading.Thread(target=neighbour)
t.start()
bad = {k: 0 for k, _ in members}
t0 = time.time()
try:
    for rep in range(reps):
        cur = snapshot()
        for (k, _), a, b in zip(members, cur, first):
            if not np.array_equal(a, b, equal_nan=True):
                bad[k] += 1
                if bad[k] <= 2:
                    print("rep %d: %s differs in %d entries (max |d| %.3e)" % (rep, k, int((a != b).sum()), float(np.nanmax(np.abs(a - b)))), flush=True)
finally:
    stop.set()
    t.join(120)
print("n=%d %s %s: %d repetitions in %.1f s, members that ever differed: %s; neighbour fits %d (%d of its own differed)" % (
    n, kern, dtype, reps, time.time() - t0, {k: v for k, v in bad.items() if v} or "none", nb[0], nb[1]), flush=True)
