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
torch.cuda.  Round 5: the schedule's free parameters are measured in the run
(`tune_schedule`), and one rank's share of a P-rank run can be rehearsed at full size
on one GPU (`rehearse_rank`: measured compute, modelled transfer).
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
    # gpx_mg_timing_ex: how long the update stream sat idle in front of a panel (chunk) that had not arrived -- EXPOSED
    # chain time, the longest single wait, the number of waits above 20 us; rehearsal: the modelled transfer time
    TIMING_KEYS_EX = TIMING_KEYS + ("exposed_wait", "exposed_wait_max", "exposed_waits_over_20us", "modelled_transfer",
                                    "modelled_remote_chain")

    def __init__(self, n, d, dtype_id=_lib.F64, kernel_id=_lib.KERNEL_GAUSSIAN, nb=None, dist=None,
                 backend="rccl", device=0, callbacks=None, rehearsal=None, adopt_from=None):
        """`callbacks` (backend="callbacks" only): an object with ctypes function pointers `.bcast` / `.allreduce`
        (gpx_mg_bcast_fn / gpx_mg_allreduce_fn of include/gpx.h), an `.error` slot and `.rank` / `.world` -- any
        transport for the host-callback data plane; default: `GlooCallbacks(dist)`."""
        self.lib = _lib.load()
        self.n, self.d = int(n), int(d)
        if rehearsal is not None:
            # ONE rank of a `world`-rank run in this process (gpx_mg_create_rehearsal): rehearsal = dict(rank, world, L_ptr,
            # ldl, alpha_ptr, link_GBps, latency_us) -- the resident factor / solution stand in for what other ranks send
            self.rank, self.world = int(rehearsal["rank"]), int(rehearsal["world"])
            self.nb = int(nb or default_nb(n, self.world))
            _lib.check(self.lib.gpx_set_device(int(device)))
            self.h = ctypes.c_void_p()
            self._cb = None
            _lib.check(self.lib.gpx_mg_create_rehearsal(
                ctypes.byref(self.h), dtype_id, kernel_id, self.n, self.d, self.nb, self.world, self.rank,
                ctypes.c_void_p(rehearsal["L_ptr"]), int(rehearsal["ldl"]), ctypes.c_void_p(rehearsal["alpha_ptr"]),
                float(rehearsal.get("link_GBps", 100.0)), float(rehearsal.get("latency_us", 20.0))))
            self.log_lh = None
            self.info = None
            return
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
            if rc == 0 and self.rank == 0 and adopt_from is None:
                rc = self.lib.gpx_mg_unique_id(ident.ctypes.data_as(ctypes.c_void_p))
                err = _lib.last_error() if rc != 0 else ""
            adopt = adopt_from is not None
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
                if not adopt:
                    dist.broadcast(torch.from_numpy(ident), src=0, group=cpu)    # the ncclUniqueId travels out of band
            else:
                _lib.check(rc)
            try:
                if adopt:
                    # ONE ncclCommInitRank per process: a handle of another width takes over the communicator (round 5: the
                    # tuning pass tries several layouts; re-creating communicators in a running job is a risk nobody needs)
                    _lib.check(self.lib.gpx_mg_adopt_comm(self.h, adopt_from.h))
                else:
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

    def set_chunks(self, chunks):
        """Row chunks per panel broadcast for the following fits (collective: the same on every rank)."""
        self._check(self.lib.gpx_mg_set_chunks(self.h, int(chunks)))

    def set_owner_first(self, on):
        """The owner of the next panel factors it before it starts its own trailing update (default: on for world >= 2;
        GPX_MG_OWNER_FIRST=0 / 1 overrides; `schedule_info()` says what is in effect)."""
        self._check(self.lib.gpx_mg_set_owner_first(self.h, 1 if on else 0))

    def set_wait_timing(self, on):
        """Time the update stream's waits for panels that have not arrived (`timing(extended=True)`: exposed_wait ...).
        Off by default: two timing-enabled event records per wait on the stream that bounds the step."""
        self._check(self.lib.gpx_mg_set_wait_timing(self.h, 1 if on else 0))

    def schedule_info(self):
        """The schedule parameters IN EFFECT for the next fit, read back from the handle."""
        v = [ctypes.c_int(0) for _ in range(4)]
        self._check(self.lib.gpx_mg_schedule_info(self.h, *[ctypes.byref(x) for x in v]))
        return {"owner_first": bool(v[0].value), "chunks": v[1].value,
                "panel_bcast": "scatter+allgather" if v[2].value else "one collective", "wait_timing": bool(v[3].value)}

    def timing(self, extended=False):
        if extended:
            ms = np.zeros(len(self.TIMING_KEYS_EX))
            self._check(self.lib.gpx_mg_timing_ex(self.h, _lib.dptr(ms), ms.size))
            return dict(zip(self.TIMING_KEYS_EX, [float(v) for v in ms]))
        ms = np.zeros(8)
        self._check(self.lib.gpx_mg_timing(self.h, _lib.dptr(ms)))
        return dict(zip(self.TIMING_KEYS, [float(v) for v in ms]))

    def chain_by_panel(self):
        """ms per panel of the last fit that this rank spent on the owner's chain (factor + pack + the last chunk of the
        column update); 0 for panels it does not own."""
        nblk = -(-self.n // self.nb)
        ms = np.zeros(nblk)
        self._check(self.lib.gpx_mg_chain_by_panel(self.h, _lib.dptr(ms), nblk))
        return ms

    def close(self):
        if self.h:
            self.lib.gpx_mg_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------- schedule tuning --
def schedule_candidates(n, world, nbs=(256, 512, 1024), chunks=(2, 4, 8)):
    """The free parameters of the multi-GPU schedule, in ONE order on every rank: block-column width, row chunks per panel
    broadcast, broadcast form (scatter + all-gather needs more than two ranks).  Grouped by nb (a new width is a new local
    layout, i.e. a new handle)."""
    out = []
    for nb in nbs:
        if nb > n or -(-n // nb) < world:                        # (every rank should own a panel)
            continue
        for sag in ((0, 1) if world > 2 else (0,)):
            for ch in chunks:
                out.append({"nb": int(nb), "chunks": int(ch), "sag": int(sag)})
    return out


def tune_schedule(candidates, measure, dist=None, budget_s=60.0, clock=time.perf_counter):
    """Measure the candidates IN ORDER while the budget lasts and pick the fastest -- together.  `measure(c)` runs one
    untimed fit with candidate c on this rank and returns its seconds.  A candidate's time is the SLOWEST rank's (MAX
    all-reduce over `dist`, a torch.distributed CPU group), and so is the clock the budget is checked against, so every
    rank measures the same candidates, stops at the same one and chooses the same triple.  At least one candidate is
    always measured.  Returns (best, table): table rows are the candidates with `fit_s` (agreed) added."""
    t_start = clock()
    table = []
    for c in candidates:
        mine = float(measure(c))
        spent = clock() - t_start
        if dist is not None and dist.get_world_size() > 1:
            import torch
            v = torch.tensor([mine, spent], dtype=torch.float64)
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
            mine, spent = float(v[0].item()), float(v[1].item())
        row = dict(c)
        if np.isfinite(mine):
            row["fit_s"] = round(mine, 5)
        else:                                              # some rank could not set this candidate up: dropped on all
            row["fit_s"] = None
            row["skipped"] = "a rank could not set this layout up"
        table.append(row)
        if spent > budget_s:
            break
    done = [r for r in table if r["fit_s"] is not None]
    if not done:
        raise RuntimeError("tune_schedule: no candidate could be measured")
    best = min(done, key=lambda r: (r["fit_s"], r["nb"], r["chunks"], r["sag"]))
    return {k: best[k] for k in ("nb", "chunks", "sag")}, table


# ------------------------------------------------------------------ rehearsal --
def rehearse_rank(N, d, rank, world, X, y, params, s, dtype_id=_lib.F64, nb=None, chunks=None, sag=None, fits=3,
                  link_GBps=100.0, latency_us=20.0, device=0, owner_first=None):
    """ONE rank's share of a `world`-rank fit at full size on ONE GPU: its own panels, packs, column
    updates and trailing updates run exactly as they would (same kernels, same streams, same look-ahead), in situ beside
    each other; the panels it does not own come out of a resident single-GPU factor of the same matrix; every broadcast is
    a delay that MODELS the transfer (bytes / link rate for a ring, 2 bytes / (P link rate) for scatter + all-gather,
    + latency per collective), and a remote panel additionally waits for what THIS rank measured as its own chain for the
    nearest panel it owns in the fit before (symmetric ranks).  Measured compute, modelled transfer -- labelled as such in
    the returned dict.  The owned block columns of the result are checked against the resident factor."""
    from .gp import GP
    from .kernels import GaussianKernel
    lib = _lib.load()
    dtype = "float64" if dtype_id == _lib.F64 else "float32"
    g = GP(GaussianKernel(float(params[0]), float(params[1])), X, y, s=float(s), dtype=dtype)
    llh_ref = float(g.log_lh)
    st = g._fit_pd()
    A, lda, al = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_void_p()
    _lib.check(lib.gpx_gp_device_ptrs(st.handle, ctypes.byref(A), ctypes.byref(lda), None, None, ctypes.byref(al), None))
    # the rider row: row n of the resident matrix <- L^-1 y (what rides through the distributed factorisation as one more
    # row of every panel; the single-GPU fit's own work row does not survive its backward solve)
    from .device import DeviceBuffer
    es_ = 8 if dtype_id == _lib.F64 else 4
    tmp = DeviceBuffer.from_host(np.ascontiguousarray(y, dtype=np.float64 if es_ == 8 else np.float32))
    _lib.check(lib.gpx_d_trsv_lower(dtype_id, A, N, lda.value, tmp.ptr, ctypes.c_void_p(A.value + N * lda.value * es_), 0, None))
    _lib.check(lib.gpx_stream_sync(None))
    tmp.free()
    reh = {"rank": rank, "world": world, "L_ptr": A.value, "ldl": lda.value, "alpha_ptr": al.value,
           "link_GBps": link_GBps, "latency_us": latency_us}
    mg = NativeDistributedGP(N, d, dtype_id=dtype_id, nb=nb, device=device, rehearsal=reh)
    try:
        if chunks is not None:
            mg.set_chunks(chunks)
        if sag is not None:
            mg.set_bcast(sag)
        if owner_first is not None:
            mg.set_owner_first(owner_first)
        mg.set_wait_timing(True)
        sched = mg.schedule_info()                      # what is IN EFFECT (the arguments may have been None)
        mg.set_data(X, y)
        runs = []
        p = np.ascontiguousarray(params, dtype=np.float64)
        for i in range(fits):
            t0 = time.perf_counter()
            mg.fit(p, s)
            wall = time.perf_counter() - t0
            tm = mg.timing(extended=True)
            tm["wall_s"] = wall
            runs.append(tm)
        chain = mg.chain_by_panel()
        # the owned block columns against the resident factor (both in HBM: a few sampled rows of each owned panel)
        es = 8 if dtype_id == _lib.F64 else 4
        npdt = np.float64 if es == 8 else np.float32
        Aloc, ld_loc = ctypes.c_void_p(), ctypes.c_int64()
        _lib.check(lib.gpx_mg_device_ptrs(mg.h, ctypes.byref(Aloc), ctypes.byref(ld_loc)))
        worst = 0.0
        nbv = mg.nb
        owned = [j for j in range(-(-N // nbv)) if j % world == rank]
        for j in owned[:: max(1, len(owned) // 6)]:
            kb = min(nbv, N - j * nbv)
            # (not the rider row: the back substitution keeps its running right-hand side there; alpha covers it)
            for r in sorted({j * nbv, j * nbv + kb - 1, min(N - 1, j * nbv + kb + 777), N - 1}):
                w = kb if r >= j * nbv + kb else (r - j * nbv + 1)
                mine, ref = np.empty(w, dtype=npdt), np.empty(w, dtype=npdt)
                _lib.check(lib.gpx_memcpy_d2h(mine.ctypes.data_as(ctypes.c_void_p),
                                              ctypes.c_void_p(Aloc.value + (r * ld_loc.value + (j // world) * nbv) * es), w * es, None))
                _lib.check(lib.gpx_memcpy_d2h(ref.ctypes.data_as(ctypes.c_void_p),
                                              ctypes.c_void_p(A.value + (r * lda.value + j * nbv) * es), w * es, None))
                _lib.check(lib.gpx_stream_sync(None))
                scale = max(1e-300, float(np.abs(ref).max()))
                worst = max(worst, float(np.abs(mine.astype(np.float64) - ref.astype(np.float64)).max()) / scale)
        alpha = mg.alpha
        alpha_ref = np.array(g.inv_Kxx_y, dtype=np.float64)
        alpha_err = float(np.abs(alpha - alpha_ref).max() / max(1e-300, np.abs(alpha_ref).max()))
        last = runs[-1]
        nblk = -(-N // nbv)
        steps = nblk
        own = chain[chain > 0]
        return {
            "what": "rehearsal of rank %d of %d on one GPU: measured compute, MODELLED transfer" % (rank, world),
            "N": N, "d": d, "dtype": dtype, "nb": nbv, "chunks": sched["chunks"], "owner_first": sched["owner_first"],
            "panel_bcast": "scatter+allgather" if sched["panel_bcast"] == "scatter+allgather" else "one collective (ring)",
            "model": {"link_GBps_sustained_assumed": link_GBps, "latency_us_per_collective_assumed": latency_us,
                      "ring_us": "bytes / rate + latency", "scatter_allgather_us": "2 bytes / (P rate) + 2 latency",
                      "remote_owner_chain": "this rank's own measured chain for its nearest owned panel, from the fit before"},
            "fits": runs,
            "rank_step_s": last["wall_s"],
            "per_step_ms": {"update": last["chain_update"] / steps, "exposed_wait": last["exposed_wait"] / steps,
                            "modelled_transfer": last["modelled_transfer"] / steps,
                            "modelled_remote_chain": last["modelled_remote_chain"] / steps,
                            "own_chain_per_owned_panel_mean": float(own.mean()) if own.size else 0.0,
                            "own_chain_per_owned_panel_max": float(own.max()) if own.size else 0.0},
            "owned_panels": int(own.size),
            "single_gpu_log_lh": llh_ref,
            "check": {"owned_columns_vs_resident_factor_max_rel_err": worst, "alpha_vs_single_gpu_max_rel_err": alpha_err},
        }
    finally:
        mg.close()


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
        holder = {"gp": None}

        def make(nb):
            """A handle of block-column width nb (a new local layout: allocation, data, communicator) and its first fit --
            the first real collectives of that communicator -- under the watchdog."""
            old_gp = holder["gp"]
            if old_gp is not None and backend != "rccl":
                old_gp.close()                          # (callbacks: nothing to take over)
                old_gp = None
            with Watchdog(wd_s, "communicator set-up (%s, world %d, nb %s)" % (backend, world, nb), rank):
                # a further handle takes over the first one's RCCL communicator: ONE ncclCommInitRank per process
                g_ = NativeDistributedGP(N, d, dtype_id=dtype_id, nb=nb, dist=dist, backend=backend, device=local_rank,
                                         adopt_from=old_gp)
            if old_gp is not None:
                old_gp.close()
            g_.set_data(X, y)
            g_.set_wait_timing(True)                    # the bench reports exposed chain time per rank
            holder["gp"] = g_
            with Watchdog(wd_s, "first fit + predict (N=%d, world %d, nb %d)" % (N, world, g_.nb), rank):
                llh0 = g_.fit(params, s)
                mean0 = g_.mean(params, Xo)
                dist.barrier()
            return llh0, mean0

        def step():
            gp = holder["gp"]
            llh = gp.fit(params, s)                    # synchronous: returns when this rank's streams have drained
            mean = gp.mean(params, Xo)
            return llh, mean

        first = make(None)
        for _ in range(max(0, args.warmup - 1)):
            step()
        # what the communicator itself says (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), from every rank:
        # evidence that the data plane really spans `world` GPUs (a silent fallback would show rccl_nranks = 0)
        comm_infos = [None] * world
        dist.all_gather_object(comm_infos, holder["gp"].comm_info())
        # The schedule's free parameters are MEASURED, not guessed (round 5): block-column width x row chunks per panel
        # broadcast x broadcast form, one untimed fit each after the warm-up, under a time budget; a candidate's time is the
        # slowest rank's, every rank stops at the same candidate and takes the same triple (tune_schedule).  xGMI is point to
        # point (7 links per GPU): whether RCCL's own broadcast or scatter + all-gather moves a panel faster, and whether the
        # owner's chain or the update bounds a step at a given width, depends on the node.  Variables that pin a parameter
        # (GPX_POTRF_NB, GPX_MG_BCAST_CHUNKS, GPX_MG_BCAST) take it out of the search.
        sched_tune = None
        if (world > 1 or os.environ.get("GPX_BENCH_FORCE_TUNE")) and not os.environ.get("GPX_BENCH_NO_TUNE"):
            nb0 = holder["gp"].nb
            nbs = (nb0,) if os.environ.get("GPX_POTRF_NB") else tuple(sorted({256, 512, 1024, nb0}, key=lambda v: (v != nb0, v)))
            chunks = (int(os.environ["GPX_MG_BCAST_CHUNKS"]),) if os.environ.get("GPX_MG_BCAST_CHUNKS") else (2, 4, 8)
            cands = schedule_candidates(N, world, nbs=nbs, chunks=chunks)
            if os.environ.get("GPX_MG_BCAST") or os.environ.get("GPX_BENCH_NO_BCAST_TUNE"):
                want = 1 if os.environ.get("GPX_MG_BCAST") == "sag" else 0
                cands = [c for c in cands if c["sag"] == want]

            def measure(c):
                """One untimed-by-the-bench fit of candidate c: seconds, or inf when its layout cannot be set up here (the
                old handle plus the new layout are in HBM together while the communicator moves: a rank may run out).  The
                verdict is agreed on by tune_schedule's MAX all-reduce: inf on one rank drops the candidate on all."""
                if holder["gp"].nb != c["nb"]:
                    ok = 1
                    if backend == "rccl":
                        # can every rank hold the second layout?  agreed BEFORE anybody enters the collective set-up
                        need = (N + 1) * (-(-(-(-N // c["nb"])) // world)) * c["nb"] * (8 if dtype_id == _lib.F64 else 4) * 1.1
                        ok = 1 if _lib.mem_free() > need else 0
                        v = torch.tensor([ok], dtype=torch.int32)
                        dist.all_reduce(v, op=dist.ReduceOp.MIN)
                        ok = int(v.item())
                    if not ok:
                        return float("inf")
                    make(c["nb"])
                gp = holder["gp"]
                gp.set_chunks(c["chunks"])
                gp.set_bcast(c["sag"])
                dist.barrier()
                with Watchdog(wd_s, "tuning fit (nb %d, chunks %d, sag %d)" % (c["nb"], c["chunks"], c["sag"]), rank):
                    t0 = time.perf_counter()
                    gp.fit(params, s)
                    return time.perf_counter() - t0

            budget = float(os.environ.get("GPX_BENCH_TUNE_BUDGET_S", "90"))
            if cands:
                best, table = tune_schedule(cands, measure, dist=dist, budget_s=budget)
                if holder["gp"].nb != best["nb"]:
                    make(best["nb"])
                holder["gp"].set_chunks(best["chunks"])
                holder["gp"].set_bcast(best["sag"])
                step()                                  # one untimed step with the chosen triple
                sched_tune = {"chosen": best, "table": table, "budget_s": budget, "candidates": len(cands),
                              "measured": len(table), "rule": "one untimed fit each; a candidate's time = the slowest rank's"}
        gp = holder["gp"]
        dist.barrier()
        _lib.check(lib.gpx_device_sync())
        if rank == 0 and not args.no_prof:
            _lib.check(lib.gpx_prof_enable(1))         # per-launch HIP events on this rank's streams
        keys = gp.TIMING_KEYS_EX[:-2]                 # (the last two are the rehearsal's modelled delays)
        chain = np.zeros(len(keys))
        t0 = time.perf_counter()
        for _ in range(args.steps):
            llh, mean_host = step()
            tm = gp.timing(extended=True)
            chain += np.array([tm[k] for k in keys])
        _lib.check(lib.gpx_device_sync())
        dist.barrier()
        elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        sec = float(elapsed.item()) / args.steps
        chain /= args.steps
        assert np.isfinite(llh) and np.isfinite(mean_host).all()
        all_chain = [None] * world
        dist.all_gather_object(all_chain, {k: round(float(v), 3) for k, v in zip(keys, chain)})
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
            "schedule_autotune": sched_tune,
            "bcast_chunks": gp.schedule_info()["chunks"],
            "schedule_in_effect": gp.schedule_info(),
            "watchdog_s": wd_s,
            "first_fit_log_lh": first[0],
            "panel_bcast": gp.comm_info()["panel_bcast"],
            "whole_step_tflops_n3_over_3": round(tfl, 3),
            "whole_step_frac_of_peak_all_gpus": round(tfl / (peak * world), 4),
            # where each rank's step went (ms per step, HIP events on its own streams): the stages, and inside
            # the factorisation the owner's chain -- panels it factored, packs, broadcasts (including the wait
            # for the root), updates -- and exposed_wait: how long the rank's update stream sat IDLE in front of a
            # panel (chunk) that had not arrived (a hidden chain shows ~0 there whatever chain_* say)
            "stage_and_chain_ms_per_rank": all_chain,
            "roofline": ({"bound": "mfma",
                          "kernel": "gpx::gemm_nt_fast_kernel<T, 128, 1, 0> (trailing SYRK updates) on rank 0",
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
