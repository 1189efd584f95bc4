"""Repeat-run soak of the lock-step batch paths beside a neighbour: `BatchEvaluator.__call__` (gpx_gp_fit_batch) and
`.value_and_grad` (gpx_gp_fit_batch_grad) on one table of restarts, every repetition equal to the first bit for bit, while a
second host thread factors single matrices in a loop; then a few fits of the headline size, bitwise.

    python tools/r6_soak_batch.py [n=4096] [rows=8] [reps=300] [dtype=float64] [big=0]
"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_processes_amd as gp              # noqa: E402
from gaussian_processes_amd import mlii          # noqa: E402

args = dict(a.split("=", 1) for a in sys.argv[1:])
n, rows, reps = int(args.get("n", 4096)), int(args.get("rows", 8)), int(args.get("reps", 300))
dtype, big = args.get("dtype", "float64"), int(args.get("big", 0))
d = 8


def synth_inputs(N, d, seed=0):
    rng = np.random.RandomState(seed)
    X = rng.uniform(-10, 10, (N, d))
    y = np.sin(X.sum(1) / np.sqrt(d)) + 0.1 * rng.randn(N)
    return X, y


X, y = synth_inputs(n, d)
rs = np.random.RandomState(2)
thetas = np.column_stack([rs.uniform(0.5, 2, rows), rs.uniform(0.25, 2, rows) * np.sqrt(d), rs.uniform(0.5, 2, rows)])
stop = threading.Event()
nb = [0, 0]


def neighbour():
    Xn, yn = synth_inputs(3000, 3, seed=5)
    g = gp.GP(gp.GaussianKernel(1.0, 0.9), Xn, yn, s=1.1, dtype=dtype)
    first = {}
    while not stop.is_set():
        for sv in (1.1, 1.2):                    # (a changed s drops the memoised fit: every log_lh factors again)
            g.set_param("s", sv)
            v = float(g.log_lh)
            nb[0] += 1
            if first.setdefault(sv, v) != v:
                nb[1] += 1


t = threading.Thread(target=neighbour)
t.start()
bad = 0
try:
    with mlii.BatchEvaluator(X, y, dtype=dtype) as ev:
        v0 = ev(thetas).copy()
        t0 = time.time()
        for rep in range(reps):
            v = ev(thetas)
            if not np.array_equal(v, v0, equal_nan=True):
                bad += 1
                print("value rep %d: rows %s differ" % (rep, np.nonzero(~((v == v0) | (np.isnan(v) & np.isnan(v0))))[0].tolist()), flush=True)
        print("n=%d rows=%d %s: values %d of %d repetitions differed (%.1f s, neighbour fits %d)" % (
            n, rows, dtype, bad, reps, time.time() - t0, nb[0]), flush=True)
        out0 = [np.array(a, copy=True) for a in ev.value_and_grad(thetas)]
        badg, t0 = 0, time.time()
        for rep in range(max(1, reps // 3)):
            out = ev.value_and_grad(thetas)
            if not all(np.array_equal(np.asarray(a), b, equal_nan=True) for a, b in zip(out, out0)):
                badg += 1
                print("gradient rep %d differs" % rep, flush=True)
        print("n=%d rows=%d %s: value + gradient %d of %d repetitions differed (%.1f s, neighbour fits %d, %d of its own differed)" % (
            n, rows, dtype, badg, max(1, reps // 3), time.time() - t0, nb[0], nb[1]), flush=True)
    if big:
        Xb, yb = synth_inputs(big, 32)
        g = gp.GP(gp.GaussianKernel(1.0, 0.5 * np.sqrt(32)), Xb, yb, s=1.0)
        l0, a0 = float(g.log_lh), np.array(g.inv_Kxx_y, copy=True)
        badb = 0
        for rep in range(4):
            g.set_param("s", 1.5)
            _ = float(g.log_lh)
            g.set_param("s", 1.0)
            l1, a1 = float(g.log_lh), g.inv_Kxx_y
            if l1 != l0 or not np.array_equal(a1, a0):
                badb += 1
                print("N=%d fit %d: log_lh %r vs %r, alpha differs in %d entries" % (big, rep, l1, l0, int((a1 != a0).sum())), flush=True)
        print("N=%d: %d of 4 repeat fits differed (neighbour fits %d)" % (big, badb, nb[0]), flush=True)
finally:
    stop.set()
    t.join(120)
