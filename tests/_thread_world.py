"""Test infrastructure: the C multi-GPU schedule (gpx_mg_*) with `world` ranks as THREADS of one process.

A GPU box admits at most six processes on its card, so a world of eight -- the north star's rank count -- cannot be
rehearsed with one process per rank there.  The schedule itself does not care where its ranks live: every rank is a
`gpx_mg` handle with its own streams, the library's per-thread state (look-ahead stream, published blocks of the
resident panel kernel) is per host thread, and the host-callback data plane only needs a broadcast and an all-reduce.
Here those rendezvous on a threading.Barrier and pass the payload through a shared host buffer -- the same two-hop
staging as multi_gpu.GlooCallbacks, without the process boundary.  ctypes releases the GIL around every C call and a
Barrier wait releases it too, so the eight ranks really run their launches concurrently.
"""
import ctypes
import threading

import numpy as np

from gaussian_processes_amd import _lib, multi_gpu
from oracle import gp_oracle as orc


class _Shared(object):
    def __init__(self, world, timeout):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=timeout)
        self.buf = None
        self.slots = [None] * world


class ThreadCallbacks(object):
    """One rank's end of the in-process collectives (the interface NativeDistributedGP(callbacks=...) expects)."""

    _NP = {_lib.F64: np.float64, _lib.F32: np.float32, 2: np.int32}

    def __init__(self, shared, rank):
        self.shared, self.rank, self.world = shared, rank, shared.world
        self.lib = _lib.load()
        self.error = None
        self.bcast = _lib.MG_BCAST_FN(self._bcast)
        self.allreduce = _lib.MG_ALLREDUCE_FN(self._allreduce)

    def _fail(self, exc):
        self.error = exc
        self.shared.barrier.abort()          # nobody is left waiting for a rank that has given up
        return 1

    def _bcast(self, user, dev_ptr, nbytes, root, stream):
        try:
            sh = self.shared
            _lib.check(self.lib.gpx_stream_sync(stream))
            if self.rank == root:
                buf = np.empty(nbytes, dtype=np.uint8)
                _lib.check(self.lib.gpx_memcpy_d2h(buf.ctypes.data_as(ctypes.c_void_p), dev_ptr, nbytes, stream))
                sh.buf = buf
            sh.barrier.wait()
            if self.rank != root:
                src = sh.buf
                assert src is not None and src.nbytes == nbytes, "broadcast size mismatch between ranks"
                _lib.check(self.lib.gpx_memcpy_h2d(dev_ptr, src.ctypes.data_as(ctypes.c_void_p), nbytes, stream))
            sh.barrier.wait()                # the root may reuse the buffer
            return 0
        except Exception as exc:             # noqa: BLE001 -- an exception must not unwind through the C frames
            return self._fail(exc)

    def _allreduce(self, user, dev_ptr, count, dtype, op, stream):
        try:
            sh = self.shared
            buf = np.empty(count, dtype=self._NP[dtype])
            _lib.check(self.lib.gpx_stream_sync(stream))
            _lib.check(self.lib.gpx_memcpy_d2h(buf.ctypes.data_as(ctypes.c_void_p), dev_ptr, buf.nbytes, stream))
            sh.slots[self.rank] = buf
            sh.barrier.wait()
            acc = sh.slots[0].copy()
            for r in range(1, self.world):   # rank order: every rank forms the same sum
                acc = acc + sh.slots[r] if op == 0 else np.maximum(acc, sh.slots[r])
            _lib.check(self.lib.gpx_memcpy_h2d(dev_ptr, acc.ctypes.data_as(ctypes.c_void_p), acc.nbytes, stream))
            sh.barrier.wait()
            return 0
        except Exception as exc:             # noqa: BLE001
            return self._fail(exc)


def run_thread_world(world, N, d, nb, m, dtype_id=0, s=1.0, sag=False, timeout=300):
    """Fit + refit + predict on `world` thread-ranks sharing GPU 0; returns rank 0's results plus every rank's log_lh."""
    X, y, Xo = orc.synth_inputs(N, d, m)
    params = np.array([1.0, 0.5 * np.sqrt(d)])
    shared = _Shared(world, timeout)
    out = [None] * world
    errs = [None] * world
    _lib.route_reset()

    def rank_main(rank):
        g = None
        try:
            cb = ThreadCallbacks(shared, rank)
            g = multi_gpu.NativeDistributedGP(N, d, dtype_id=dtype_id, nb=nb, backend="callbacks", device=0, callbacks=cb)
            g.set_data(X, y)
            if sag:
                g.set_bcast(True)
            llh = g.fit(params, s)
            llh2 = g.fit(params, s)
            mean = g.mean(params, Xo)
            out[rank] = {"log_lh": llh, "log_lh2": llh2, "mean": mean, "alpha": g.alpha if rank == 0 else None,
                         "info": g.info, "logdet": g.logdet}
        except BaseException as exc:         # noqa: BLE001
            errs[rank] = exc
            shared.barrier.abort()
        finally:
            if g is not None:
                g.close()

    threads = [threading.Thread(target=rank_main, args=(r,), name="gpx-rank-%d" % r) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout + 60)
    alive = [t.name for t in threads if t.is_alive()]
    assert not alive, "ranks still running after the timeout: %s" % alive
    first = next((e for e in errs if e is not None and not isinstance(e, threading.BrokenBarrierError)), None) or \
        next((e for e in errs if e is not None), None)
    if first is not None:
        raise first
    res = dict(out[0])
    res["log_lh_per_rank"] = [o["log_lh"] for o in out]
    res["sag_routes"] = _lib.route_count(_lib.ROUTE_MG_BCAST_SAG)
    res["one_routes"] = _lib.route_count(_lib.ROUTE_MG_BCAST_ONE)
    return res
