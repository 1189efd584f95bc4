"""The Python reference schedule of the multi-GPU factorisation -- TEST INFRASTRUCTURE.

The product's multi-GPU path is the C schedule behind the ABI (csrc/gpx_mg.hip, `NativeDistributedGP`).  This
module keeps the same schedule written out in Python over an `ops` object: `HipOps` drives libgpx's device entry
points on torch CUDA tensors / streams, `CpuOps` (tests/_dist_helpers.py) emulates them on the CPU so that the
ownership maps, buffer reuse and the order of the collectives are exercised by the gloo tests without a GPU.
It lived in the product package until round 3; nothing in gaussian_processes_amd/ imports it.

Its solves are the ROUND-2 formulation (forward substitution block by block with one all-reduce per block, backward
substitution with a transposed panel product per block: `gpx_d_trsv_lower_cols`, `gpx_d_panel_gemv_t`, still
exported building blocks of the ABI); the C schedule has since moved on (y rides along as a row of the matrix, the
backward sweep is right-looking -- csrc/gpx_mg.hip).  Both must produce the oracle's alpha: two independent
formulations over the same layout maps, checked against the same numbers.
"""
import ctypes
import math
import os

import numpy as np

from gaussian_processes_amd import _lib
from gaussian_processes_amd.multi_gpu import BlockCyclic, default_nb


# ------------------------------------------------------------------- HIP ops --
class HipOps(object):
    """Device operations through the C ABI, on torch CUDA tensors (torch is used
    for memory, streams and torch.distributed only)."""

    def __init__(self, dtype_id, device):
        import torch
        self.torch = torch
        self.lib = _lib.load()
        self.dtype_id = dtype_id
        self.tdtype = torch.float64 if dtype_id == _lib.F64 else torch.float32
        self.es = 8 if dtype_id == _lib.F64 else 4
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        _lib.check(self.lib.gpx_set_device(device))
        self.main = torch.cuda.Stream(device=self.device)
        # panel factorisation + broadcast: critical path of the next step -> high priority
        self.side = torch.cuda.Stream(device=self.device, priority=-1)

    # memory
    def empty(self, shape, dtype=None):
        return self.torch.empty(shape, dtype=dtype or self.tdtype, device=self.device)

    def zeros(self, shape, dtype=None):
        return self.torch.zeros(shape, dtype=dtype or self.tdtype, device=self.device)

    def from_host(self, a):
        return self.torch.as_tensor(np.ascontiguousarray(a)).to(self.device, dtype=self.tdtype)

    def to_host(self, t):
        return t.detach().to("cpu").numpy().astype(np.float64)

    def copy_(self, dst, src, stream):
        with self.torch.cuda.stream(stream):
            dst.copy_(src)

    # streams / events
    def record(self, stream):
        e = self.torch.cuda.Event()
        e.record(stream)
        return e

    def wait(self, stream, event):
        stream.wait_event(event)

    def sync(self):
        self.torch.cuda.synchronize(self.device)

    def stream_ctx(self, stream):
        return self.torch.cuda.stream(stream)

    def _p(self, t, off_elems=0):
        return ctypes.c_void_p(t.data_ptr() + off_elems * self.es)

    @staticmethod
    def _s(stream):
        return ctypes.c_void_p(stream.cuda_stream)

    # kernels
    def kmat_block(self, A, ld, x, n, d, r0, cl, kb, kernel_id, params, s, stream):
        """A[r0:n, cl:cl+kb] <- K(x[r0:n], x[r0:r0+kb]) + s^2 on the block's diagonal (lower tiles)."""
        p = np.ascontiguousarray(params, dtype=np.float64)
        _lib.check(self.lib.gpx_d_kmat(self.dtype_id, kernel_id, _lib.K, self._p(x, r0 * d), n - r0,
                                       self._p(x, r0 * d), kb, d, _lib.dptr(p), float(s) * float(s),
                                       _lib.LOWER, self._p(A, r0 * ld + cl), ld, self._s(stream)))

    def potrf_panel(self, A, ld, n, r0, c0, kb, info, stream):
        _lib.check(self.lib.gpx_d_potrf_panel(self.dtype_id, self._p(A), ld, n, r0, c0, kb,
                                              ctypes.c_void_p(info.data_ptr()), self._s(stream)))

    def pack_panel(self, A, ld, r0, c0, rows, kb, buf, nb, stream):
        with self.torch.cuda.stream(stream):
            buf.view(-1)[: rows * nb].view(rows, nb)[:, :kb].copy_(A[r0:r0 + rows, c0:c0 + kb])

    def syrk_bc(self, A, ld, n, row_begin, cl0, cl1, buf, ldp, k0, kb, nb, P, rank, stream):
        _lib.check(self.lib.gpx_d_syrk_bc(self.dtype_id, n, row_begin, self._p(A), ld, cl0, cl1,
                                          self._p(buf), ldp, k0, kb, nb, P, rank, self._s(stream)))

    def trsv_cols(self, A, ld, r0, cl, nrows, ncols, w, z, stream):
        """Trapezoid forward solve on block column (r0, cl): z[r0:r0+ncols], w[r0+ncols:] updated."""
        _lib.check(self.lib.gpx_d_trsv_lower_cols(self.dtype_id, self._p(A, r0 * ld + cl), nrows, ld,
                                                  ncols, self._p(w, r0), self._p(z, r0), self._s(stream)))

    def panel_gemv_t(self, A, ld, r0, cl, rows, ncols, x, x_off, y, work, stream):
        _lib.check(self.lib.gpx_d_panel_gemv_t(self.dtype_id, self._p(A, r0 * ld + cl), ld, rows, ncols,
                                               self._p(x, x_off), self._p(y),
                                               ctypes.c_void_p(work.data_ptr()), self._s(stream)))

    def trsv_diag_t(self, A, ld, r0, cl, kb, b, x, x_off, stream):
        """x[x_off:x_off+kb] <- L_jj^-T b for the kb x kb diagonal block at (r0, cl)."""
        _lib.check(self.lib.gpx_d_trsv_lower(self.dtype_id, self._p(A, r0 * ld + cl), kb, ld, self._p(b),
                                             self._p(x, x_off), 1, self._s(stream)))

    def logdet_block(self, A, ld, r0, cl, kb, out, stream):
        _lib.check(self.lib.gpx_d_logdet_chol(self.dtype_id, self._p(A, r0 * ld + cl), kb, ld,
                                              ctypes.c_void_p(out.data_ptr()), self._s(stream)))

    def dot(self, a, b, n, out, stream):
        _lib.check(self.lib.gpx_d_dot(self.dtype_id, self._p(a), self._p(b), n,
                                      ctypes.c_void_p(out.data_ptr()), self._s(stream)))

    def mean(self, kernel_id, xo, m0, m1, x, n, d, params, alpha, out, stream):
        p = np.ascontiguousarray(params, dtype=np.float64)
        if m1 > m0:
            _lib.check(self.lib.gpx_d_mean(self.dtype_id, kernel_id, self._p(xo, m0 * d), m1 - m0,
                                           self._p(x), n, d, _lib.dptr(p), self._p(alpha),
                                           self._p(out, m0), self._s(stream)))


class TorchComm(object):
    """torch.distributed collectives on flat slices (backend nccl = RCCL on GPUs, gloo in tests)."""

    def __init__(self, dist, to_tensor=None):
        self.dist = dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self.to_tensor = to_tensor or (lambda a: a)
        # rehearsal switch: issue the collectives even in a world of one rank
        self.always = bool(os.environ.get("GPX_FORCE_COLLECTIVES"))
        # panel broadcast algorithm: "bcast" (one collective, default) or "sag" (scatter + all-gather)
        self.bcast_mode = os.environ.get("GPX_DIST_BCAST", "bcast")
        self.sag_min = int(os.environ.get("GPX_DIST_SAG_MIN", str(1 << 20)))     # elements

    def broadcast(self, arr, start, count, src):
        if not ((self.world > 1 or self.always) and count > 0):
            return
        flat = self.to_tensor(arr).view(-1)[start:start + count]
        if self.bcast_mode == "sag" and self.world > 2 and count % self.world == 0 and count >= self.sag_min:
            # scatter + direct all-gather, point-to-point only: (1) the root sends a different 1/P of
            # the payload to every peer, (2) every rank sends its piece straight to every other rank.
            # On a fully connected xGMI node each phase moves 1/P of the bytes over every link at once,
            # where a ring collective pushes the whole payload through one link after the other.
            # Opt-in (GPX_DIST_BCAST=sag) until measured on a real node.
            dist = self.dist
            chunk = count // self.world
            piece = lambda p: flat[p * chunk:(p + 1) * chunk]
            if self.rank == src:
                ops = [dist.P2POp(dist.isend, piece(p), p) for p in range(self.world) if p != src]
            else:
                ops = [dist.P2POp(dist.irecv, piece(self.rank), src)]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            ops = [dist.P2POp(dist.isend, piece(self.rank), p)
                   for p in range(self.world) if p != self.rank and p != src]
            if self.rank != src:
                ops += [dist.P2POp(dist.irecv, piece(q), q) for q in range(self.world) if q != self.rank]
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            return
        self.dist.broadcast(flat, src=src)

    def all_reduce_sum(self, arr, start, count):
        if (self.world > 1 or self.always) and count > 0:
            self.dist.all_reduce(self.to_tensor(arr).view(-1)[start:start + count])

    def all_reduce_max(self, arr, start, count):
        if (self.world > 1 or self.always) and count > 0:
            self.dist.all_reduce(self.to_tensor(arr).view(-1)[start:start + count],
                                 op=self.dist.ReduceOp.MAX)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()


class LocalComm(object):
    """Single-rank stand-in (world size 1): every collective is a no-op."""
    rank, world = 0, 1

    def broadcast(self, arr, start, count, src):
        pass

    def all_reduce_sum(self, arr, start, count):
        pass

    def all_reduce_max(self, arr, start, count):
        pass

    def barrier(self):
        pass


# ------------------------------------------------------------ the distributed GP --
class DistributedGP(object):
    """One GP spread over P ranks (this object = one rank's share)."""

    def __init__(self, ops, comm, n, d, kernel_id=_lib.KERNEL_GAUSSIAN, nb=None):
        self.ops, self.comm = ops, comm
        self.n, self.d, self.kernel_id = int(n), int(d), kernel_id
        self.lay = BlockCyclic(n, nb or default_nb(n, comm.world), comm.world, comm.rank)
        lay = self.lay
        self.A = ops.empty((lay.n, lay.ld))
        self.pbuf = [ops.empty((lay.n, lay.nb)), ops.empty((lay.n, lay.nb))]
        self.w = ops.empty((lay.n,))
        self.z = ops.empty((lay.n,))
        self.alpha = ops.empty((lay.n,))
        self.tmp = ops.empty((lay.nb,))
        self.work = ops.empty((max(1, -(-lay.n // 256)) * lay.nb,), dtype=ops.torch.float64)
        self.scal = ops.zeros((4,), dtype=ops.torch.float64)    # [0] logdet block [1] y^T alpha
        self.info = ops.zeros((4,), dtype=ops.torch.int32)
        self.x = self.y = None
        self.logdet = None
        self.yta = None
        self.info_host = None

    def set_data(self, x, y):
        self.x = self.ops.from_host(np.asarray(x).reshape(self.n, self.d))
        self.y = self.ops.from_host(np.asarray(y).reshape(self.n))
        self.ops.sync()

    # -- kernel matrix: every rank builds the block columns it owns (no exchange) --
    def build(self, params, s):
        ops, lay = self.ops, self.lay
        for j in lay.my_blocks:
            ops.kmat_block(self.A, lay.ld, self.x, lay.n, self.d, lay.k0(j), lay.local_col(j), lay.kb(j),
                           self.kernel_id, params, s, ops.main)

    # -- factorisation --
    def _factor_and_bcast(self, j, buf, stream):
        ops, lay, comm = self.ops, self.lay, self.comm
        r0, kb = lay.k0(j), lay.kb(j)
        rows = lay.n - r0
        if lay.owner(j) == comm.rank:
            cl = lay.local_col(j)
            ops.potrf_panel(self.A, lay.ld, lay.n, r0, cl, kb, self.info, stream)
            ops.pack_panel(self.A, lay.ld, r0, cl, rows, kb, buf, lay.nb, stream)
        with ops.stream_ctx(stream):
            comm.broadcast(buf, 0, rows * lay.nb, lay.owner(j))

    def factor(self):
        ops, lay, comm = self.ops, self.lay, self.comm
        S, Q = ops.main, ops.side
        ops.wait(Q, ops.record(S))                       # the kernel build is done
        self._factor_and_bcast(0, self.pbuf[0], Q)
        ep = ops.record(Q)
        readers_done = [None, None]                      # last update that read pbuf[i]
        for k in range(lay.nblk):
            k0, kb = lay.k0(k), lay.kb(k)
            r = k0 + kb
            ops.wait(S, ep)                              # panel k is here, in pbuf[k % 2]
            if r >= lay.n:
                break
            P_k = self.pbuf[k % 2]
            nxt = k + 1
            own_next = lay.owner(nxt) == comm.rank
            jl_first = lay.first_local_block_after(k)
            if own_next:
                cl = lay.local_col(nxt)
                ops.syrk_bc(self.A, lay.ld, lay.n, r, cl, cl + lay.nb, P_k, lay.nb, k0, kb, lay.nb,
                            lay.P, lay.rank, S)
                ops.wait(Q, ops.record(S))
                jl_first = lay.first_local_block_after(nxt)
            if readers_done[nxt % 2] is not None:
                ops.wait(Q, readers_done[nxt % 2])       # update k-1 no longer reads that buffer
            self._factor_and_bcast(nxt, self.pbuf[nxt % 2], Q)
            ep = ops.record(Q)
            if jl_first is not None:
                ops.syrk_bc(self.A, lay.ld, lay.n, r, jl_first * lay.nb, lay.ncols_local, P_k, lay.nb, k0,
                            kb, lay.nb, lay.P, lay.rank, S)
            readers_done[k % 2] = ops.record(S)
        ops.wait(S, ops.record(Q))

    # -- solves: alpha = K^-1 y, replicated on every rank at the end --
    def solve(self):
        ops, lay, comm = self.ops, self.lay, self.comm
        S = ops.main
        with ops.stream_ctx(S):
            if comm.rank == 0:
                ops.copy_(self.w, self.y, S)
            else:
                self.w.zero_()
            for j in range(lay.nblk):                    # forward: L z = y
                r0, kb = lay.k0(j), lay.kb(j)
                comm.all_reduce_sum(self.w, r0, kb)
                if lay.owner(j) == comm.rank:
                    ops.trsv_cols(self.A, lay.ld, r0, lay.local_col(j), lay.n - r0, kb, self.w, self.z, S)
            for j in reversed(range(lay.nblk)):          # backward: L^T alpha = z
                r0, kb = lay.k0(j), lay.kb(j)
                if lay.owner(j) == comm.rank:
                    cl = lay.local_col(j)
                    ops.copy_(self.tmp[:kb], self.z[r0:r0 + kb], S)
                    below = lay.n - r0 - kb
                    if below > 0:
                        ops.panel_gemv_t(self.A, lay.ld, r0 + kb, cl, below, kb, self.alpha, r0 + kb,
                                         self.tmp, self.work, S)
                    ops.trsv_diag_t(self.A, lay.ld, r0, cl, kb, self.tmp, self.alpha, r0, S)
                comm.broadcast(self.alpha, r0, kb, lay.owner(j))

    def reduce_scalars(self):
        ops, lay, comm = self.ops, self.lay, self.comm
        S = ops.main
        with ops.stream_ctx(S):
            acc = ops.zeros((2,), dtype=ops.torch.float64)
            for j in lay.my_blocks:
                ops.logdet_block(self.A, lay.ld, lay.k0(j), lay.local_col(j), lay.kb(j), self.scal, S)
                acc[0:1] += self.scal[0:1]
            ops.dot(self.y, self.alpha, lay.n, self.scal[1:2], S)
            comm.all_reduce_sum(acc, 0, 1)
            # LAPACK info: the FIRST failing minor over all ranks; 0 when none failed
            big = 2 ** 30
            info = self.info[0:1]
            key = ((big - info) * (info > 0)).to(ops.torch.int32)
            comm.all_reduce_max(key, 0, 1)
        ops.sync()
        host = ops.to_host(acc)
        self.logdet = float(host[0])
        self.yta = float(ops.to_host(self.scal)[1])
        k = int(ops.to_host(key)[0])
        self.info_host = 0 if k == 0 else big - k

    def fit(self, params, s):
        with self.ops.stream_ctx(self.ops.main):
            self.info.zero_()
        self.build(params, s)
        self.factor()
        self.solve()
        self.reduce_scalars()
        return self.log_lh

    @property
    def log_lh(self):
        # gp/gp.py:362-365 and gp_c.pyx:22-29
        if self.info_host != 0 or not (self.logdet >= _lib.MIN_LOG):
            return -np.inf
        return -0.5 * self.yta - 0.5 * self.logdet - 0.5 * self.n * math.log(2 * math.pi)

    def mean(self, xo_dev, m, params, out):
        """Posterior mean at m test points: every rank evaluates a slice, one all-reduce."""
        ops, comm = self.ops, self.comm
        S = ops.main
        per = -(-m // comm.world)
        m0, m1 = min(m, comm.rank * per), min(m, (comm.rank + 1) * per)
        with ops.stream_ctx(S):
            out.zero_()
            ops.mean(self.kernel_id, xo_dev, m0, m1, self.x, self.n, self.d, params, self.alpha, out, S)
            comm.all_reduce_sum(out, 0, m)
        return out
