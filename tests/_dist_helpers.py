"""Test infrastructure for the multi-GPU schedule (gaussian_processes_amd/multi_gpu.py): one rank of the C schedule
(gpx_mg_*) per process, several ranks sharing GPU 0 with host-callback collectives over gloo.  (The Python restatement of
the schedule and its CPU emulator of the device ops, which rounds 1 - 4 kept here, are gone: one schedule, the C one.)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import gp_oracle as orc  # noqa: E402


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
