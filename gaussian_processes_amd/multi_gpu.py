"""Multi-GPU GP fit/predict: 1-D block-cyclic block columns, one process per GPU.

The reference has nothing distributed (SURVEY sections 2 and 5); this is the
north-star's scaling path.  The n x n kernel matrix is split into
block columns of width nb; global block column j lives on rank j % P as local
block j // P (row-major local matrix, n rows x ceil(nblk / P) * nb columns).

Per panel k (right-looking, one-panel look-ahead):
  owner(k)   factors its block column below the diagonal        gpx_d_potrf_panel
             packs it into a contiguous (n - k0) x nb buffer    (device 2-D copy)
  all ranks  broadcast of that buffer, root = owner(k)          torch.distributed
             (backend "nccl" = RCCL over xGMI on the node)
  all ranks  update their own block columns j > k               gpx_d_syrk_bc
The owner of panel k + 1 updates that block column first and factors +
broadcasts it on a side stream while everybody finishes update k.

Solves: forward substitution walks the block columns with one all-reduce of an
nb-vector per block (the residual of that block, whose updates are spread over
the ranks); back substitution broadcasts each finished alpha block.  logdet and
the posterior mean are local sums + one all-reduce.

Device work is reached through an `ops` object (HipOps: libgpx.so on torch CUDA
tensors).  The schedule itself is plain Python and is exercised on CPU by the
gloo tests with an emulator `ops` that lives in tests/.
"""
import ctypes
import json
import math
import os
import time

import numpy as np

from . import _lib


# --------------------------------------------------------------------- layout --
class BlockCyclic(object):
    """Index maps of the 1-D block-cyclic column distribution."""

    def __init__(self, n, nb, P, rank):
        assert nb % 64 == 0 and 1024 % nb == 0, "nb must be 64/128/256/512/1024"
        self.n, self.nb, self.P, self.rank = int(n), int(nb), int(P), int(rank)
        self.nblk = -(-self.n // self.nb)
        self.my_blocks = [j for j in range(self.nblk) if j % self.P == self.rank]
        self.ncols_local = max(1, len(self.my_blocks)) * self.nb
        self.ld = self.ncols_local            # multiple of 64 => 16-element aligned

    def owner(self, j):
        return j % self.P

    def local_col(self, j):
        return (j // self.P) * self.nb

    def k0(self, j):
        return j * self.nb

    def kb(self, j):
        return min(self.nb, self.n - j * self.nb)

    def first_local_block_after(self, k):
        """Index into my_blocks of the first local block whose global index is > k."""
        for jl, j in enumerate(self.my_blocks):
            if j > k:
                return jl
        return None


def default_nb(n, world=8):
    """Block-column width.  Few ranks (the update per rank outlasts the owner's chain of column
    update + panel + pack + broadcast): the single-GPU choice, 1024 for large n.  More ranks: 512 --
    the chain is the critical path and twice as many, half as long steps keep every rank's share of
    the update even."""
    env = os.environ.get("GPX_POTRF_NB")
    if env:
        return int(env)
    if n <= 2048:
        return 128
    if n <= 12288:
        return 256
    if n <= 32768 or world > 2:
        return 512
    return 1024


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


# ------------------------------------------------- the C schedule (gpx_mg_*, RCCL) --
class GlooCallbacks(object):
    """Host-side collectives for gpx_mg_create_cb: device pointer -> host staging -> torch.distributed
    (any CPU backend) -> device.  Slow by construction; it exists so that the C schedule can be run with
    several ranks on ONE GPU (RCCL cannot put two ranks on a device) and verified in tests/"""

    _NP = {_lib.F64: np.float64, _lib.F32: np.float32, 2: np.int32}

    def __init__(self, dist):
        import torch
        self.torch, self.dist = torch, dist
        self.rank = dist.get_rank()
        self.lib = _lib.load()
        self.error = None
        self.bcast = _lib.MG_BCAST_FN(self._bcast)
        self.allreduce = _lib.MG_ALLREDUCE_FN(self._allreduce)

    def _bcast(self, user, dev_ptr, nbytes, root, stream):
        try:
            buf = np.empty(nbytes, dtype=np.uint8)
            _lib.check(self.lib.gpx_stream_sync(stream))
            if self.rank == root:
                _lib.check(self.lib.gpx_memcpy_d2h(buf.ctypes.data_as(ctypes.c_void_p), dev_ptr, nbytes, stream))
            self.dist.broadcast(self.torch.from_numpy(buf), src=root)
            if self.rank != root:
                _lib.check(self.lib.gpx_memcpy_h2d(dev_ptr, buf.ctypes.data_as(ctypes.c_void_p), nbytes, stream))
            return 0
        except Exception as exc:       # an exception must not unwind through the C frames
            self.error = exc
            return 1

    def _allreduce(self, user, dev_ptr, count, dtype, op, stream):
        try:
            buf = np.empty(count, dtype=self._NP[dtype])
            _lib.check(self.lib.gpx_stream_sync(stream))
            _lib.check(self.lib.gpx_memcpy_d2h(buf.ctypes.data_as(ctypes.c_void_p), dev_ptr, buf.nbytes, stream))
            self.dist.all_reduce(self.torch.from_numpy(buf),
                                 op=self.dist.ReduceOp.SUM if op == 0 else self.dist.ReduceOp.MAX)
            _lib.check(self.lib.gpx_memcpy_h2d(dev_ptr, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes, stream))
            return 0
        except Exception as exc:
            self.error = exc
            return 1


class NativeDistributedGP(object):
    """One rank's share of a GP whose factorisation schedule runs in C (csrc/gpx_mg.hip): HIP streams,
    events and the panel broadcasts are issued by libgpx, the collectives are RCCL (`backend="rccl"`,
    one process per GPU; the ncclUniqueId travels over `dist`, a torch.distributed CPU group) or host
    callbacks over `dist` (`backend="callbacks"`: several ranks may share a GPU).  `dist=None`: one rank."""

    TIMING_KEYS = ("kernel_build", "factor", "solve", "reduce", "chain_panel", "chain_pack", "chain_bcast",
                   "chain_update")

    def __init__(self, n, d, dtype_id=_lib.F64, kernel_id=_lib.KERNEL_GAUSSIAN, nb=None, dist=None,
                 backend="rccl", device=0):
        self.lib = _lib.load()
        self.n, self.d = int(n), int(d)
        self.rank, self.world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
        self.nb = int(nb or default_nb(n, self.world))
        _lib.check(self.lib.gpx_set_device(int(device)))
        self.h = ctypes.c_void_p()
        self._cb = None
        if backend == "rccl":
            import torch
            ident = np.zeros(_lib.MG_ID_BYTES, dtype=np.uint8)
            if self.rank == 0:
                _lib.check(self.lib.gpx_mg_unique_id(ident.ctypes.data_as(ctypes.c_void_p)))
            if dist is not None and self.world > 1:
                t = torch.from_numpy(ident)
                dist.broadcast(t, src=0)                      # out-of-band exchange over the CPU group
            _lib.check(self.lib.gpx_mg_create(ctypes.byref(self.h), dtype_id, kernel_id, self.n, self.d, self.nb,
                                              self.world, self.rank, ident.ctypes.data_as(ctypes.c_void_p)))
        elif backend == "callbacks":
            if dist is not None and self.world > 1:
                self._cb = GlooCallbacks(dist)
                b = ctypes.cast(self._cb.bcast, ctypes.c_void_p)
                a = ctypes.cast(self._cb.allreduce, ctypes.c_void_p)
            else:
                b = a = None
            _lib.check(self.lib.gpx_mg_create_cb(ctypes.byref(self.h), dtype_id, kernel_id, self.n, self.d, self.nb,
                                                 self.world, self.rank, b, a, None))
        else:
            raise ValueError("backend must be 'rccl' or 'callbacks'")
        self.log_lh = None
        self.info = None

    def _check(self, rc):
        if rc != 0 and self._cb is not None and self._cb.error is not None:
            err, self._cb.error = self._cb.error, None
            raise err
        _lib.check(rc)

    def set_data(self, x, y):
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(self.n, self.d))
        y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(self.n))
        self._check(self.lib.gpx_mg_set_data(self.h, _lib.dptr(x), _lib.dptr(y)))

    def fit(self, params, s):
        p = np.ascontiguousarray(params, dtype=np.float64)
        llh, info = ctypes.c_double(0.0), ctypes.c_int(0)
        self._check(self.lib.gpx_mg_fit(self.h, _lib.dptr(p), float(s), ctypes.byref(llh), ctypes.byref(info)))
        self.log_lh, self.info = llh.value, info.value
        return self.log_lh

    def mean(self, params, xo):
        p = np.ascontiguousarray(params, dtype=np.float64)
        xo = np.ascontiguousarray(np.asarray(xo, dtype=np.float64).reshape(-1, self.d))
        out = np.empty(xo.shape[0], dtype=np.float64)
        self._check(self.lib.gpx_mg_mean(self.h, _lib.dptr(p), _lib.dptr(xo), xo.shape[0], _lib.dptr(out)))
        return out

    @property
    def alpha(self):
        out = np.empty(self.n, dtype=np.float64)
        self._check(self.lib.gpx_mg_get_alpha(self.h, _lib.dptr(out)))
        return out

    @property
    def logdet(self):
        v = ctypes.c_double(0.0)
        self._check(self.lib.gpx_mg_scalars(self.h, ctypes.byref(v), None, None))
        return v.value

    def timing(self):
        ms = np.zeros(8)
        self._check(self.lib.gpx_mg_timing(self.h, _lib.dptr(ms)))
        return dict(zip(self.TIMING_KEYS, [float(v) for v in ms]))

    def close(self):
        if self.h:
            self.lib.gpx_mg_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ benchmark --
def bench_distributed(args, X, y, Xo, params, s, dtype_id, residual_check=None):
    """bench.py's N > 1 leg: the same workload as N = 1 (strong scaling), one rank per GPU.  Control plane:
    a torch.distributed gloo group (rendezvous, barriers, the ncclUniqueId, the max-over-ranks clock); data
    plane: libgpx's C schedule with RCCL collectives.  GPX_DIST_BACKEND=gloo swaps the data-plane
    collectives for host callbacks over the same gloo group (rehearsal with several ranks on one GPU)."""
    import sys
    import torch
    import torch.distributed as dist
    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    if os.environ.get("GPX_BENCH_SINGLE_DEVICE"):      # rehearsal on a 1-GPU box
        local_rank = 0
    if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # one node: do not depend on the hostname resolving
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    lib = _lib.load()
    backend = "callbacks" if os.environ.get("GPX_DIST_BACKEND", "nccl") == "gloo" else "rccl"
    fallback_note = None
    if backend == "rccl":
        # pre-flight, so that a rank without a usable RCCL does not leave the others waiting inside
        # ncclCommInitRank: every rank checks locally that its GPU is there and that librccl loads (an
        # ncclUniqueId can be made), the verdict is agreed on over the gloo group, and if any rank says no,
        # ALL ranks take the host-callback data plane (slower, same schedule, same result)
        ok, why = 1, ""
        try:
            _lib.check(lib.gpx_set_device(int(local_rank)))
            probe = np.zeros(_lib.MG_ID_BYTES, dtype=np.uint8)
            _lib.check(lib.gpx_mg_unique_id(probe.ctypes.data_as(ctypes.c_void_p)))
        except Exception as exc:                          # noqa: BLE001 -- any failure means "not usable here"
            ok, why = 0, "%s: %s" % (type(exc).__name__, exc)
        verdict = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
        if int(verdict.item()) == 0:
            reasons = [None] * world
            dist.all_gather_object(reasons, why)
            fallback_note = "RCCL data plane unavailable (%s): host callbacks over gloo instead" % "; ".join(
                "rank %d: %s" % (i, r) for i, r in enumerate(reasons) if r)
            if rank == 0:
                sys.stderr.write("bench: " + fallback_note + "\n")
            backend = "callbacks"
    N, d, m = args.n, args.d, args.m
    # RCCL prints a version banner on stdout when its communicator is created; keep stdout clean for the
    # single JSON line (stderr still shows everything)
    import sys
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        gp = NativeDistributedGP(N, d, dtype_id=dtype_id, dist=dist, backend=backend, device=local_rank)
        gp.set_data(X, y)

        def step():
            llh = gp.fit(params, s)                    # synchronous: returns when this rank's streams have drained
            mean = gp.mean(params, Xo)
            return llh, mean

        for _ in range(args.warmup):
            step()
        dist.barrier()
        _lib.check(lib.gpx_device_sync())
        if rank == 0 and not args.no_prof:
            _lib.check(lib.gpx_prof_enable(1))         # per-launch HIP events on this rank's streams
        chain = np.zeros(8)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            llh, mean_host = step()
            tm = gp.timing()
            chain += np.array([tm[k] for k in gp.TIMING_KEYS])
        _lib.check(lib.gpx_device_sync())
        dist.barrier()
        elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        sec = float(elapsed.item()) / args.steps
        chain /= args.steps
        assert np.isfinite(llh) and np.isfinite(mean_host).all()
        all_chain = [None] * world
        dist.all_gather_object(all_chain, {k: round(float(v), 3) for k, v in zip(gp.TIMING_KEYS, chain)})
        peak = 78.6 if dtype_id == _lib.F64 else 157.3
        tfl = (N ** 3 / 3.0) / sec / 1e12
        rank0_gemm = None
        if rank == 0 and not args.no_prof:
            a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            _lib.check(lib.gpx_prof_read(1, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
            _lib.check(lib.gpx_prof_enable(0))
            if b.value > 0:
                rank0_gemm = {"launches_per_step": a.value / args.steps, "ms_per_step": b.value / args.steps,
                              "tflops": c.value / (b.value * 1e-3) / 1e12}
        check = None
        if rank == 0 and residual_check is not None:
            res, nres = residual_check(gp.alpha)
            tol = 1e-9 if dtype_id == _lib.F64 else 2e-3
            assert res < tol, "sampled-row residual %.3e exceeds %.1e" % (res, tol)
            check = {"sampled_rows": nres, "max_abs_residual_K_alpha_minus_y_over_max_y": res, "tolerance": tol}
        result = {
            "metric": "GP fit+predict wall-clock (kernel build + Cholesky + solve + log_lh + posterior mean)",
            "value": round(sec, 4), "unit": "s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(sec * 1e3, 2), "higher_is_better": False, "scaling": "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "N=%d d=%d RBF(GaussianKernel) %s, m=%d test points, h=1 w=0.5*sqrt(d) s=1"
                                   % (N, d, args.dtype, m), "N": N, "d": d, "m": m,
                       "parallelism": "1-D block-cyclic block columns (nb=%d) over %d GPUs, %s panel broadcast, "
                                      "schedule in C (gpx_mg_*)"
                                      % (gp.nb, world, "RCCL" if backend == "rccl" else "host-callback (gloo)")},
            "log_lh": llh,
            "check": check,
            "data_plane_fallback": fallback_note,
            "whole_step_tflops_n3_over_3": round(tfl, 3),
            "whole_step_frac_of_peak_all_gpus": round(tfl / (peak * world), 4),
            # where each rank's step went (ms per step, HIP events on its own streams): the stages, and inside
            # the factorisation the owner's chain -- panels it factored, packs, broadcasts (including the wait
            # for the root), updates
            "stage_and_chain_ms_per_rank": all_chain,
            "roofline": ({"bound": "mfma",
                          "kernel": "gpx::gemm_nt_fast_kernel<T, 128, 1, 128> (trailing SYRK updates) on rank 0",
                          "achieved": round(rank0_gemm["tflops"], 3), "peak": peak, "unit": "TFLOP/s",
                          "frac": round(rank0_gemm["tflops"] / peak, 4), "traffic": None,
                          "launches_per_step": rank0_gemm["launches_per_step"],
                          "kernel_ms_per_step_rank0": round(rank0_gemm["ms_per_step"], 3)}
                         if rank0_gemm else None),
        }
        gp.close()
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    dist.barrier()
    dist.destroy_process_group()
    return result
