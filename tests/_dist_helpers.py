"""Test infrastructure for the multi-GPU schedule (gaussian_processes_amd/multi_gpu.py).

CpuOps is an EMULATOR of the device-ops interface (HipOps) on torch CPU tensors
with numpy/scipy arithmetic.  It exists so that the distributed schedule -- the
ownership maps, the look-ahead order, the panel broadcasts, the all-reduce /
broadcast pattern of the solves -- can be run with world_size > 1 over gloo in a
container without GPUs.  It is not part of the product and is never importable
from it.
"""
import contextlib
import os
import sys

import numpy as np
import scipy.linalg
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import gp_oracle as orc  # noqa: E402


class _Stream(object):
    pass


class CpuOps(object):
    def __init__(self):
        self.torch = torch
        self.main, self.side = _Stream(), _Stream()

    def empty(self, shape, dtype=None):
        return torch.zeros(shape, dtype=dtype or torch.float64)

    zeros = empty

    def from_host(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).clone()

    def to_host(self, t):
        return t.detach().numpy().astype(np.float64)

    def copy_(self, dst, src, stream):
        dst.copy_(src)

    def record(self, stream):
        return None

    def wait(self, stream, event):
        pass

    def sync(self):
        pass

    def stream_ctx(self, stream):
        return contextlib.nullcontext()

    # ---- the kernels, restated with numpy ----
    def kmat_block(self, A, ld, x, n, d, r0, cl, kb, kernel_id, params, s, stream):
        xs = x.numpy()
        kind = "gaussian" if kernel_id == 0 else "periodic"
        K = orc.kernel_matrix(kind, "K", xs[r0:n], xs[r0:r0 + kb], params)
        K[np.arange(kb), np.arange(kb)] += float(s) ** 2
        A.numpy()[r0:n, cl:cl + kb] = K

    def potrf_panel(self, A, ld, n, r0, c0, kb, info, stream):
        a = A.numpy()
        blk = np.tril(a[r0:r0 + kb, c0:c0 + kb])
        blk = blk + np.tril(blk, -1).T
        try:
            L = scipy.linalg.cholesky(blk, lower=True)
        except np.linalg.LinAlgError as e:
            if info[0] == 0:
                info[0] = r0 + int(str(e).split("-")[0])
            L = np.full_like(blk, np.nan)
        a[r0:r0 + kb, c0:c0 + kb] = np.tril(L) + np.triu(a[r0:r0 + kb, c0:c0 + kb], 1)
        if r0 + kb < n:
            a[r0 + kb:n, c0:c0 + kb] = scipy.linalg.solve_triangular(
                L, a[r0 + kb:n, c0:c0 + kb].T, lower=True).T

    def pack_panel(self, A, ld, r0, c0, rows, kb, buf, nb, stream):
        buf.view(-1)[: rows * nb].view(rows, nb)[:, :kb].copy_(A[r0:r0 + rows, c0:c0 + kb])

    def syrk_bc(self, A, ld, n, row_begin, cl0, cl1, buf, ldp, k0, kb, nb, P, rank, stream):
        a = A.numpy()
        pan = buf.numpy().reshape(-1)[: (n - k0) * ldp].reshape(n - k0, ldp)[:, :kb]
        for c in range(cl0, cl1):
            gc = ((c // nb) * P + rank) * nb + c % nb
            lo = max(row_begin, gc)
            if lo < n:
                a[lo:n, c] -= pan[lo - k0:n - k0] @ pan[gc - k0]

    def trsv_cols(self, A, ld, r0, cl, nrows, ncols, w, z, stream):
        a = A.numpy()
        L = a[r0:r0 + ncols, cl:cl + ncols]
        zz = scipy.linalg.solve_triangular(L, w.numpy()[r0:r0 + ncols], lower=True)
        z.numpy()[r0:r0 + ncols] = zz
        if nrows > ncols:
            w.numpy()[r0 + ncols:r0 + nrows] -= a[r0 + ncols:r0 + nrows, cl:cl + ncols] @ zz

    def panel_gemv_t(self, A, ld, r0, cl, rows, ncols, x, x_off, y, work, stream):
        a = A.numpy()
        y.numpy()[:ncols] -= a[r0:r0 + rows, cl:cl + ncols].T @ x.numpy()[x_off:x_off + rows]

    def trsv_diag_t(self, A, ld, r0, cl, kb, b, x, x_off, stream):
        L = A.numpy()[r0:r0 + kb, cl:cl + kb]
        x.numpy()[x_off:x_off + kb] = scipy.linalg.solve_triangular(L, b.numpy()[:kb], lower=True, trans="T")

    def logdet_block(self, A, ld, r0, cl, kb, out, stream):
        out[0] = 2.0 * float(np.log(np.diag(A.numpy()[r0:r0 + kb, cl:cl + kb])).sum())

    def dot(self, a, b, n, out, stream):
        out[0] = float(a.numpy()[:n] @ b.numpy()[:n])

    def mean(self, kernel_id, xo, m0, m1, x, n, d, params, alpha, out, stream):
        if m1 > m0:
            kind = "gaussian" if kernel_id == 0 else "periodic"
            K = orc.kernel_matrix(kind, "K", xo.numpy()[m0:m1], x.numpy(), params)
            out.numpy()[m0:m1] = K @ alpha.numpy()


def worker(rank, world, port, backend, use_hip, N, d, nb, m, outdir, s=1.0):
    """One rank of a world_size-`world` fit + predict; rank 0 writes results to outdir."""
    import torch.distributed as dist
    from gaussian_processes_amd import multi_gpu, _lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    try:
        X, y, Xo = orc.synth_inputs(N, d, m)
        h, w = 1.0, 0.5 * np.sqrt(d)
        params = np.array([h, w])
        import _py_schedule
        ops = _py_schedule.HipOps(_lib.F64, 0) if use_hip else CpuOps()
        comm = _py_schedule.TorchComm(dist)
        g = _py_schedule.DistributedGP(ops, comm, N, d, nb=nb)
        g.set_data(X, y)
        llh = g.fit(params, s)
        xo_dev = ops.from_host(Xo)
        out = ops.empty((m,))
        g.mean(xo_dev, m, params, out)
        ops.sync()
        if rank == 0:
            np.savez(os.path.join(outdir, "result.npz"), log_lh=llh, alpha=ops.to_host(g.alpha),
                     mean=ops.to_host(out), logdet=g.logdet, info=g.info_host)
    finally:
        dist.destroy_process_group()


def run_world(world, backend, use_hip, N, d, nb, m, outdir, s=1.0):
    import socket
    import torch.multiprocessing as mp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    mp.spawn(worker, args=(world, port, backend, use_hip, N, d, nb, m, outdir, s), nprocs=world, join=True)
    return np.load(os.path.join(outdir, "result.npz"))


def native_worker(rank, world, port, N, d, nb, m, outdir, dtype_id, s=1.0, opts=None):
    """One rank of the C schedule (gpx_mg_*) with host-callback collectives over gloo; all ranks on GPU 0.
    opts: {"sag": True} panel broadcasts as scatter + all-gather; {"inject": {rank: value}} that rank's first fit
    behaves as if its factorisation had left `value` in the device info word (every rank must then fail)."""
    import torch.distributed as dist
    from gaussian_processes_amd import multi_gpu, _lib
    opts = opts or {}
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        X, y, Xo = orc.synth_inputs(N, d, m)
        params = np.array([1.0, 0.5 * np.sqrt(d)])
        g = multi_gpu.NativeDistributedGP(N, d, dtype_id=dtype_id, nb=nb, dist=dist, backend="callbacks", device=0)
        g.set_data(X, y)
        if opts.get("sag"):
            g.set_bcast(True)
        failed = None
        if rank in opts.get("inject", {}):
            _lib.check(_lib.load().gpx_debug_mg_inject_info(g.h, int(opts["inject"][rank])))
        if opts.get("inject"):
            try:
                g.fit(params, s)
                failed = "no error"
            except _lib.GpxError as exc:
                failed = "GpxError: %s" % exc
            flags = [None] * world
            dist.all_gather_object(flags, failed)
        _lib.route_reset()
        llh = g.fit(params, s)
        llh2 = g.fit(params, s)                       # a second fit reuses buffers, events and streams
        mean = g.mean(params, Xo)
        if rank == 0:
            np.savez(os.path.join(outdir, "result.npz"), log_lh=llh, log_lh2=llh2, alpha=g.alpha, mean=mean,
                     logdet=g.logdet, info=g.info, timing=np.array(list(g.timing().values())),
                     sag_routes=_lib.route_count(_lib.ROUTE_MG_BCAST_SAG), one_routes=_lib.route_count(_lib.ROUTE_MG_BCAST_ONE),
                     inject_outcomes=np.array([str(f) for f in (flags if opts.get("inject") else [])]))
        g.close()
    finally:
        dist.destroy_process_group()


def run_native_world(world, N, d, nb, m, outdir, dtype_id=0, s=1.0, opts=None):
    import socket
    import torch.multiprocessing as mp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    mp.spawn(native_worker, args=(world, port, N, d, nb, m, outdir, dtype_id, s, opts), nprocs=world, join=True)
    return np.load(os.path.join(outdir, "result.npz"))
