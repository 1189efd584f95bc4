"""Multi-GPU GP fit/predict: 1-D block-cyclic block columns, one process per GPU.

The reference has nothing distributed (SURVEY sections 2 and 5); this is the
north-star's scaling path.  The n x n kernel matrix is split into
block columns of width nb; global block column j lives on rank j % P as local
block j // P (row-major local matrix, n rows x ceil(nblk / P) * nb columns).

The schedule -- panel factorisation, pack, panel broadcast, trailing updates with
one-panel look-ahead, the distributed solves, HIP streams / events and the RCCL
collectives -- runs in C behind the ABI (csrc/gpx_mg.hip, ``gpx_mg_*``);
`NativeDistributedGP` is its Python face.  torch.distributed (a CPU group: gloo)
is the CONTROL plane only: rendezvous, the ncclUniqueId, agreeing that every rank's
local allocations succeeded before anybody enters ncclCommInitRank, barriers and
the max-over-ranks clock of the benchmark.  Nothing in this module touches
torch.cuda.  (The same schedule written out in Python over device-op objects, which
the CPU gloo tests use to exercise ownership maps and collective order without a
GPU, is test infrastructure: tests/_py_schedule.py.)
"""
import ctypes
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


# ------------------------------------------------- the C schedule (gpx_mg_*, RCCL) --
class GlooCallbacks(object):
    """Host-side collectives for gpx_mg_create_cb: device pointer -> host staging -> torch.distributed
    (any CPU backend) -> device.  Slow by construction; it exists so that the C schedule can be run with
    several ranks on ONE GPU (RCCL cannot put two ranks on a device) and verified in tests/"""

    _NP = {_lib.F64: np.float64, _lib.F32: np.float32, 2: np.int32}

    def __init__(self, dist):
        import torch
        from .mlii import _cpu_group
        self.torch, self.dist = torch, dist
        self.group = _cpu_group(dist)          # host buffers: never over an nccl default group
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
            self.dist.broadcast(self.torch.from_numpy(buf), src=root, group=self.group)
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
                                 op=self.dist.ReduceOp.SUM if op == 0 else self.dist.ReduceOp.MAX, group=self.group)
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
                 backend="rccl", device=0, callbacks=None):
        """`callbacks` (backend="callbacks" only): an object with ctypes function pointers `.bcast` / `.allreduce`
        (gpx_mg_bcast_fn / gpx_mg_allreduce_fn of include/gpx.h), an `.error` slot and `.rank` / `.world` -- any
        transport for the host-callback data plane; default: `GlooCallbacks(dist)`."""
        self.lib = _lib.load()
        self.n, self.d = int(n), int(d)
        if callbacks is not None:
            self.rank, self.world = int(callbacks.rank), int(callbacks.world)
        else:
            self.rank, self.world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
        self.nb = int(nb or default_nb(n, self.world))
        _lib.check(self.lib.gpx_set_device(int(device)))
        self.h = ctypes.c_void_p()
        self._cb = None
        if backend == "rccl":
            # two steps with an agreement in between: a rank whose local allocation fails (the local matrix and two
            # panel buffers in HBM, three streams) must not leave the others blocked inside ncclCommInitRank
            rc = self.lib.gpx_mg_create_local(ctypes.byref(self.h), dtype_id, kernel_id, self.n, self.d, self.nb,
                                              self.world, self.rank)
            err = _lib.last_error() if rc != 0 else ""
            ident = np.zeros(_lib.MG_ID_BYTES, dtype=np.uint8)
            if rc == 0 and self.rank == 0:
                rc = self.lib.gpx_mg_unique_id(ident.ctypes.data_as(ctypes.c_void_p))
                err = _lib.last_error() if rc != 0 else ""
            if dist is not None and self.world > 1:
                import torch
                from .mlii import _cpu_group
                cpu = _cpu_group(dist)           # CPU tensors: a gloo side group when the default group is nccl
                ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=cpu)
                if int(ok.item()) == 0:
                    self.close()
                    raise _lib.GpxError("multi-GPU set-up failed on %s before the communicator was created%s"
                                        % ("this rank" if rc != 0 else "another rank", (": " + err) if err else ""))
                dist.broadcast(torch.from_numpy(ident), src=0, group=cpu)    # the ncclUniqueId travels out of band
            else:
                _lib.check(rc)
            try:
                _lib.check(self.lib.gpx_mg_connect(self.h, ident.ctypes.data_as(ctypes.c_void_p)))
            except Exception:
                self.close()
                raise
        elif backend == "callbacks":
            if self.world > 1:
                self._cb = callbacks if callbacks is not None else GlooCallbacks(dist)
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

    def comm_info(self):
        """What the communicator itself reports: RCCL's rank count (0 for the callback back-end), this rank's
        index and HIP device as RCCL sees them, and the panel-broadcast algorithm in force."""
        nr, rk, dev, sag = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        self._check(self.lib.gpx_mg_comm_info(self.h, ctypes.byref(nr), ctypes.byref(rk), ctypes.byref(dev),
                                              ctypes.byref(sag)))
        return {"rccl_nranks": nr.value, "rank": rk.value, "device": dev.value,
                "panel_bcast": "scatter+allgather (send/recv)" if sag.value else "one collective per chunk"}

    def set_bcast(self, sag):
        """Panel broadcast algorithm for the following fits (collective: the same on every rank)."""
        self._check(self.lib.gpx_mg_set_bcast(self.h, 1 if sag else 0))

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
WATCHDOG_EXIT = 86      # exit status of a rank whose watchdog fired (bench.py's launcher recognises it)


class Watchdog(object):
    """Wall-clock bound around a phase that can only hang, never fail -- ncclCommInitRank with a peer that never
    arrives, the first collective over a link that is not there.  The guarded call sits in C with the GIL released,
    so a timer thread still runs: on expiry it says which phase on stderr and ends THIS process with status
    WATCHDOG_EXIT (os._exit: no unwinding through frames blocked in RCCL).  Every rank carries its own; peers of a
    hung rank are stuck in the same collective and expire with it, so the whole job fails loudly within `seconds`
    instead of holding its GPUs until the lease ends.  Nothing is re-executed in a process that touched the GPU."""

    def __init__(self, seconds, what, rank=0, on_expire=None):
        import threading
        self.what, self.rank, self.seconds = what, rank, float(seconds)
        self._on_expire = on_expire or (lambda: os._exit(WATCHDOG_EXIT))
        self._timer = threading.Timer(self.seconds, self._fire) if self.seconds > 0 else None
        if self._timer is not None:
            self._timer.daemon = True
            self._timer.start()

    def _fire(self):
        import sys
        try:
            sys.stderr.write("bench: WATCHDOG rank %d: '%s' did not finish within %.0f s -- exiting with status %d\n"
                             % (self.rank, self.what, self.seconds, WATCHDOG_EXIT))
            sys.stderr.flush()
        finally:
            self._on_expire()

    def cancel(self):
        if self._timer is not None:
            self._timer.cancel()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.cancel()
        return False


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
        # ncclCommInitRank: every rank checks locally that its GPU is there and that librccl loads
        # (gpx_mg_probe), the verdict is agreed on over the gloo group, and if any rank says no,
        # ALL ranks take the host-callback data plane (slower, same schedule, same result)
        ok, why = 1, ""
        try:
            _lib.check(lib.gpx_set_device(int(local_rank)))
            _lib.check(lib.gpx_mg_probe())                 # dlopen + symbols only: no bootstrap thread, no socket
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
        # communicator set-up and the first fit (the first real collectives) under a watchdog: see Watchdog
        wd_s = float(os.environ.get("GPX_BENCH_WATCHDOG_S", "300"))
        with Watchdog(wd_s, "communicator set-up (%s, world %d)" % (backend, world), rank):
            gp = NativeDistributedGP(N, d, dtype_id=dtype_id, dist=dist, backend=backend, device=local_rank)
        gp.set_data(X, y)

        def step():
            llh = gp.fit(params, s)                    # synchronous: returns when this rank's streams have drained
            mean = gp.mean(params, Xo)
            return llh, mean

        with Watchdog(wd_s, "first fit + predict (N=%d, world %d)" % (N, world), rank):
            first = step()
            dist.barrier()
        for _ in range(max(0, args.warmup - 1)):
            step()
        # what the communicator itself says (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), from every rank:
        # evidence that the data plane really spans `world` GPUs (a silent fallback would show rccl_nranks = 0)
        comm_infos = [None] * world
        dist.all_gather_object(comm_infos, gp.comm_info())
        # panel broadcast algorithm: measured, not guessed.  xGMI is point to point (7 links per GPU); whether RCCL's own
        # broadcast or the scatter + all-gather form moves a panel faster depends on the node.  One untimed fit each
        # (after the warm-up), the slower rank decides, every rank takes the same mode.
        bcast_tune = None
        if world > 2 and not os.environ.get("GPX_MG_BCAST") and not os.environ.get("GPX_BENCH_NO_BCAST_TUNE"):
            times = {}
            for mode in (0, 1):
                gp.set_bcast(mode)
                dist.barrier()
                t0 = time.perf_counter()
                gp.fit(params, s)
                el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
                dist.all_reduce(el, op=dist.ReduceOp.MAX)
                times[mode] = float(el.item())
            best = 1 if times[1] < times[0] else 0
            gp.set_bcast(best)
            bcast_tune = {"one_collective_s": round(times[0], 4), "scatter_allgather_s": round(times[1], 4),
                          "chosen": "scatter+allgather" if best else "one collective per chunk"}
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
            "rccl_nranks": comm_infos[0]["rccl_nranks"],
            "comm_info_per_rank": comm_infos,
            "panel_bcast_autotune": bcast_tune,
            "watchdog_s": wd_s,
            "first_fit_log_lh": first[0],
            "panel_bcast": gp.comm_info()["panel_bcast"],
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
