"""Batched ML-II evaluation: log marginal likelihood of many hyper-parameter settings.

The reference has no optimiser (``fit_MLII`` was removed in 1.0.3, CHANGELOG.md:19);
its ML-II inner step is "set params -> read log_lh" (gp/gp.py:216-223, 337-367).
This harness runs that step for a whole table of restarts on one resident data set:
one ``gpx_gp`` handle per process, x / y uploaded once, then ONE call of
``gpx_gp_fit_batch``: the kernel matrices of all rows sit in HBM side by side and are
factored in lock-step (every launch covers all of them), so the chain of small
dependent launches that bounds a single n = 8192 factorisation is paid once per batch.
With a process group the rows are dealt round-robin to the ranks
(replicas only: no intra-GP sharding) and one all-reduce assembles the
table on every rank.
"""
import ctypes

import numpy as np

from . import _lib

__all__ = ["log_lh_batch", "best_restart", "BatchEvaluator", "optimize"]

_KERNEL_IDS = {"gaussian": (_lib.KERNEL_GAUSSIAN, 2), "periodic": (_lib.KERNEL_PERIODIC, 3)}


class BatchEvaluator(object):
    """One resident data set, many sweeps: keeps the ``gpx_gp`` handle (x, y and the lock-step
    workspace in HBM) alive between calls, so that an outer optimiser pays the upload and the
    tens-of-GB workspace allocation once.  ``ev(thetas)`` -> log_lh of every row."""

    def __init__(self, x, y, kernel="gaussian", dtype="float64", device=None):
        self.kid, self.nkp = _KERNEL_IDS[kernel]
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        n = x.shape[0]
        d = 1 if x.ndim == 1 else x.shape[1]
        if y.shape != (n,):
            raise ValueError("invalid shape for y: %s" % str(y.shape))
        self.lib = _lib.load()
        if device is not None:
            _lib.check(self.lib.gpx_set_device(int(device)))
        dt = _lib.F64 if dtype in ("float64", "f64") else _lib.F32
        self.h = ctypes.c_void_p()
        self.n = n
        _lib.check(self.lib.gpx_gp_create(ctypes.byref(self.h), dt, self.kid, n, d))
        try:
            _lib.check(self.lib.gpx_gp_set_data(self.h, _lib.dptr(x), _lib.dptr(y)))
        except Exception:
            self.close()
            raise

    def __call__(self, thetas):
        th = np.ascontiguousarray(np.atleast_2d(np.asarray(thetas, dtype=np.float64)))
        if th.shape[1] != self.nkp + 1:
            raise ValueError("thetas must have %d columns (kernel params + s)" % (self.nkp + 1))
        res = np.empty(th.shape[0], dtype=np.float64)
        if th.shape[0]:
            _lib.check(self.lib.gpx_gp_fit_batch(self.h, _lib.dptr(th), th.shape[0], _lib.dptr(res), None))
        return res

    def value_and_grad(self, thetas, clamp=True):
        """(log_lh, dloglh_dtheta) of every row: one ``gpx_gp_fit_batch_grad`` call -- the lock-step factorisation,
        then per row the reference's ``dloglh_dtheta`` (gp/gp.py:398-433, gp_c.pyx:34-49) from the row's own factor.
        ``clamp=False`` returns the log marginal likelihood WITHOUT the reference's ``logdet < MIN -> -inf`` clamp
        (gp_c.pyx:22-29) wherever the matrix is positive definite: -1/2 y^T a - 1/2 logdet - n/2 log 2 pi from the
        row's (logdet, y^T a) -- what an optimiser needs at n beyond a few thousand, where the clamped value is
        -inf on most of the parameter space (an extension; the gradient is the reference's either way)."""
        th = np.ascontiguousarray(np.atleast_2d(np.asarray(thetas, dtype=np.float64)))
        if th.shape[1] != self.nkp + 1:
            raise ValueError("thetas must have %d columns (kernel params + s)" % (self.nkp + 1))
        B = th.shape[0]
        val = np.empty(B, dtype=np.float64)
        grad = np.empty((B, self.nkp + 1), dtype=np.float64)
        ldy = np.empty((B, 2), dtype=np.float64)
        if B:
            _lib.check(self.lib.gpx_gp_fit_batch_grad(self.h, _lib.dptr(th), B, _lib.dptr(val), _lib.dptr(grad), _lib.dptr(ldy), None))
            if not clamp:
                ok = np.isfinite(ldy).all(axis=1)
                raw = -0.5 * ldy[:, 1] - 0.5 * ldy[:, 0] - 0.5 * self.n * np.log(2 * np.pi)
                val = np.where(ok, raw, val)
        return val, grad

    def close(self):
        if self.h:
            self.lib.gpx_gp_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def log_lh_batch(x, y, thetas, kernel="gaussian", dtype="float64", dist=None, device=None, concurrency=None,
                 batched=True):
    """log_lh for every row ``(kernel params..., s)`` of `thetas`.

    x: (n,) or (n, d); y: (n,); thetas: (r, n_params + 1).  Rows with invalid parameters
    (kernel parameter < EPS or s < 0, the reference's ValueError conditions) and rows whose
    kernel matrix is not positive definite give ``-inf`` / ``nan`` as the reference would
    (-inf for non-PD, gp/gp.py:362-365; nan marks a row that would have raised ValueError).
    `dist`: an initialised ``torch.distributed`` module (any backend) or None: rows are dealt round-robin
    to the ranks (rank r takes rows r, r + world, ...; each rank runs them on ITS GPU -- `device`), one
    all-reduce of a (value, code) table over a CPU group assembles the result on every rank.
    `batched` (default): this process's rows go through one ``gpx_gp_fit_batch`` call (lock-step
    factorisation of all their matrices).  ``batched=False`` is the row-at-a-time route:
    `concurrency`: handles (= host threads, each with its own HIP streams) working on this
    process's rows at the same time (default 1).  ctypes drops the GIL for the duration of a
    call, the library keeps its per-thread scratch and look-ahead streams thread-local, and every
    handle has its own stream.  Measured on one MI355X, 64 restarts at n = 8192, d = 8
    (tools/mlii_bench.py): 13.0 ms per restart with 1 handle, 10.7 ms with 4, worse with 2 or 8 --
    the factorisation of a matrix this small is a chain of dependent launches, not throughput.
    """
    kid, nkp = _KERNEL_IDS[kernel]
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    thetas = np.atleast_2d(np.asarray(thetas, dtype=np.float64))
    if thetas.shape[1] != nkp + 1:
        raise ValueError("thetas must have %d columns (kernel params + s)" % (nkp + 1))
    n = x.shape[0]
    d = 1 if x.ndim == 1 else x.shape[1]
    if y.shape != (n,):
        raise ValueError("invalid shape for y: %s" % str(y.shape))
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    lib = _lib.load()
    if device is not None:
        _lib.check(lib.gpx_set_device(int(device)))
    dt = _lib.F64 if dtype in ("float64", "f64") else _lib.F32
    out = np.zeros(thetas.shape[0], dtype=np.float64)
    eps = np.finfo(np.float64).eps
    mine = list(range(rank, thetas.shape[0], world))
    if concurrency is None:
        concurrency = 1
    concurrency = max(1, min(int(concurrency), len(mine) or 1))

    def work(rows):
        if device is not None:
            _lib.check(lib.gpx_set_device(int(device)))         # the current device is per host thread
        h = ctypes.c_void_p()
        _lib.check(lib.gpx_gp_create(ctypes.byref(h), dt, kid, n, d))
        try:
            _lib.check(lib.gpx_gp_set_data(h, _lib.dptr(x), _lib.dptr(y)))
            for i in rows:
                p = np.ascontiguousarray(thetas[i, :nkp])
                s = float(thetas[i, nkp])
                if (p < eps).any() or s < 0 or not np.isfinite(thetas[i]).all():
                    out[i] = np.nan
                    continue
                _lib.check(lib.gpx_gp_set_params(h, _lib.dptr(p), s))
                _lib.check(lib.gpx_gp_fit(h, None))
                v = ctypes.c_double(0.0)
                _lib.check(lib.gpx_gp_log_lh(h, ctypes.byref(v)))
                out[i] = v.value
        finally:
            lib.gpx_gp_destroy(h)

    if batched and concurrency == 1:
        if device is not None:
            _lib.check(lib.gpx_set_device(int(device)))
        if mine:
            with BatchEvaluator(x, y, kernel=kernel, dtype=dtype, device=device) as ev:
                out[mine] = ev(thetas[mine])
    elif concurrency == 1:
        work(mine)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=concurrency) as pool:
            for f in [pool.submit(work, mine[t::concurrency]) for t in range(concurrency)]:
                f.result()                                      # re-raises a worker's exception
    if world > 1:
        import torch
        # -inf / nan do not survive a SUM all-reduce of zero-padded tables: ship a finite code
        code = np.where(np.isnan(out), 2.0, np.where(np.isneginf(out), 1.0, 0.0))
        val = np.where(code > 0, 0.0, out)
        # control-plane traffic (2 x rows doubles): a CPU group.  A process group whose default backend is RCCL gets
        # a gloo side group here -- the product package never touches torch.cuda
        group = _cpu_group(dist)
        t = torch.from_numpy(np.stack([val, code]))
        dist.all_reduce(t, group=group)
        t = t.numpy()
        out = np.where(t[1] == 2.0, np.nan, np.where(t[1] == 1.0, -np.inf, t[0]))
    return out


_CPU_GROUPS = {}


def _cpu_group(dist):
    """None (the default group) when it is a CPU backend already, else one gloo group per process group."""
    if dist.get_backend() != "nccl":
        return None
    key = id(dist)
    if key not in _CPU_GROUPS:
        _CPU_GROUPS[key] = dist.new_group(backend="gloo")          # collective: every rank gets here together
    return _CPU_GROUPS[key]


def best_restart(x, y, thetas, **kw):
    """(index, theta, log_lh) of the restart with the largest log marginal likelihood."""
    llh = log_lh_batch(x, y, thetas, **kw)
    ok = np.where(np.isnan(llh), -np.inf, llh)
    i = int(np.argmax(ok))
    return i, np.asarray(thetas)[i], llh[i]


class _LockStep(object):
    """Rendezvous of R optimiser threads with ONE serving thread: every optimiser thread hands in its current point and
    sleeps; when all optimisers that are still running have done so, the serving thread (the caller of `optimize`: the
    only thread that ever touches the GPU handle, so the library's per-thread streams and scratch exist once) evaluates
    the whole table in a single batched call and wakes them.  An optimiser that has finished leaves; the batch shrinks."""

    def __init__(self, evaluate, count):
        import threading
        self.evaluate, self.active = evaluate, count
        self.cv = threading.Condition()
        self.pending, self.results, self.generation, self.calls, self.error = {}, {}, 0, 0, None

    def request(self, i, theta):
        with self.cv:
            gen = self.generation
            self.pending[i] = np.array(theta, dtype=np.float64)
            self.cv.notify_all()
            while self.generation == gen and self.error is None:
                self.cv.wait()
            if self.error is not None:
                raise RuntimeError("batched evaluation failed: %r" % (self.error,))
            return self.results[i]

    def leave(self):
        with self.cv:
            self.active -= 1
            self.cv.notify_all()

    def serve(self):
        """Run on the serving thread until every optimiser has left."""
        while True:
            with self.cv:
                while self.active > 0 and len(self.pending) < self.active:
                    self.cv.wait()
                if self.active <= 0:
                    return
                idx = sorted(self.pending)
                table = np.array([self.pending[i] for i in idx])
            try:
                vals, grads = self.evaluate(table)               # outside the lock: the optimisers are all asleep
                res = {i: (vals[k], grads[k]) for k, i in enumerate(idx)}
                err = None
            except BaseException as exc:                         # noqa: BLE001 -- every sleeper must wake and see it
                res, err = {}, exc
            with self.cv:
                self.results, self.pending, self.error = res, {}, err
                self.generation += 1
                self.calls += 1
                self.cv.notify_all()
            if err is not None:
                raise err


def optimize(x, y, thetas0, kernel="gaussian", dtype="float64", device=None, bounds=None, maxiter=50, clamp=False,
             options=None, evaluator=None):
    """ML-II by L-BFGS-B from every row of `thetas0` at once, all restarts in LOCK-STEP (what the reference's
    ``fit_MLII`` did one restart at a time before it was removed, CHANGELOG.md:19; its objective and gradient are
    ``GP.log_lh`` / ``GP.dloglh_dtheta``, gp/gp.py:337-367, 398-433).

    One optimiser (scipy's L-BFGS-B, bounded) per restart, each on its own host thread; a function evaluation does
    not run by itself: it joins a rendezvous, and when every restart that is still running has asked, ONE
    ``gpx_gp_fit_batch_grad`` call (made by the calling thread) evaluates value and gradient of all of them -- the
    lock-step factorisation the value-only sweep uses, plus K^-1 per row.  The optimisers work in log(theta) (every
    parameter is positive: gp/kernels/gaussian.py:62-69, gp/gp.py:192-193), minimising -log_lh with gradient
    -theta * dloglh/dtheta.

    thetas0: (R, n_params + 1) rows (kernel params..., s).  bounds: (n_params + 1, 2) on theta (default: each start value
    / 100 ... x 100 over all restarts, never below 1e-6).  clamp=False (default) follows the unclamped log marginal
    likelihood (`BatchEvaluator.value_and_grad`); clamp=True keeps the reference's -inf plateau, on which an
    optimiser cannot move.  `evaluator`: an object with ``value_and_grad(thetas, clamp=...)`` to use instead of a new
    `BatchEvaluator` (tests; a caller that keeps its data set resident).  Returns a dict: theta (R, n_params + 1),
    log_lh (R,), log_lh0 (R,), nit, nfev (R,), batched_calls, best (index of the largest final log_lh), messages."""
    import threading
    from scipy.optimize import minimize
    th0 = np.atleast_2d(np.asarray(thetas0, dtype=np.float64))
    R = th0.shape[0]
    if bounds is None:
        lo = np.maximum(th0.min(0) / 100.0, 1e-6)
        hi = th0.max(0) * 100.0
    else:
        bounds = np.asarray(bounds, dtype=np.float64)
        lo, hi = bounds[:, 0], bounds[:, 1]
    lb = list(zip(np.log(lo), np.log(hi)))
    big = 1e300
    ev = evaluator if evaluator is not None else BatchEvaluator(x, y, kernel=kernel, dtype=dtype, device=device)
    try:
        v0, _ = ev.value_and_grad(th0, clamp=clamp)
        ls = _LockStep(lambda table: ev.value_and_grad(table, clamp=clamp), R)
        out = [None] * R

        def run(i):
            try:
                last = [None, None]                               # the last finite point this optimiser saw, and its value

                def f(u):
                    th = np.exp(u)
                    val, grad = ls.request(i, th)
                    if not np.isfinite(val) or not np.isfinite(grad).all():
                        # not positive definite (or clamped): a wall.  The line search needs something it can
                        # interpolate: a value well above the last finite one, rising with the distance from it
                        if last[0] is None:
                            return big, np.zeros_like(u)
                        du = u - last[0]
                        pen = 1e3 * (1.0 + abs(last[1]))
                        return last[1] + pen * (1.0 + float(du @ du)), 2.0 * pen * du
                    last[0], last[1] = np.array(u), -float(val)
                    return -val, -(grad * th)
                opts = {"maxiter": int(maxiter)}
                opts.update(options or {})
                out[i] = minimize(f, np.log(np.clip(th0[i], lo, hi)), jac=True, method="L-BFGS-B", bounds=lb, options=opts)
            except BaseException as exc:      # noqa: BLE001 -- reported below; the other restarts go on
                out[i] = exc
            finally:
                ls.leave()

        threads = [threading.Thread(target=run, args=(i,), daemon=True) for i in range(R)]
        for t in threads:
            t.start()
        try:
            ls.serve()
        finally:
            for t in threads:
                t.join()
        for o in out:
            if isinstance(o, BaseException):
                raise o
        theta = np.exp(np.array([o.x for o in out]))
        final, _ = ev.value_and_grad(theta, clamp=clamp)
    finally:
        if evaluator is None:
            ev.close()
    # a restart never ends below where it started (a start on the wall stays there)
    keep = ~(final >= v0) | ~np.isfinite(v0)
    theta[keep] = th0[keep]
    final = np.where(keep, v0, final)
    ok = np.where(np.isfinite(final), final, -np.inf)
    return {"theta": theta, "log_lh": final, "log_lh0": v0, "nit": np.array([o.nit for o in out]),
            "nfev": np.array([o.nfev for o in out]), "batched_calls": ls.calls + 2, "best": int(np.argmax(ok)),
            "messages": [str(o.message) for o in out]}
