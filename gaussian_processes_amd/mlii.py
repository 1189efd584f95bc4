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

__all__ = ["log_lh_batch", "best_restart", "BatchEvaluator"]

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
