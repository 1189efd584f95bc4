"""world_size > 1 tests of the multi-GPU schedule on CPU (gloo), with the device ops
emulated by tests/_dist_helpers.CpuOps.  Checks ownership maps, look-ahead order,
panel broadcasts and the distributed solves against the oracle."""
import numpy as np
import pytest

from gaussian_processes_amd.multi_gpu import BlockCyclic
from _py_schedule import DistributedGP, LocalComm
from oracle import gp_oracle as orc
from _dist_helpers import CpuOps, run_world


def test_block_cyclic_maps():
    lay = BlockCyclic(1000, 128, 3, 1)
    assert lay.nblk == 8 and lay.my_blocks == [1, 4, 7]
    assert lay.owner(5) == 2 and lay.local_col(4) == 128 and lay.kb(7) == 1000 - 7 * 128
    assert lay.first_local_block_after(1) == 1 and lay.first_local_block_after(7) is None
    cover = sorted(j for r in range(3) for j in BlockCyclic(1000, 128, 3, r).my_blocks)
    assert cover == list(range(8))


def _check(res, N, d, m, s=1.0):
    X, y, Xo = orc.synth_inputs(N, d, m)
    o = orc.OracleGP("gaussian", (1.0, 0.5 * np.sqrt(d)), X, y, s)
    assert int(res["info"]) == 0
    np.testing.assert_allclose(float(res["log_lh"]), o.log_lh, rtol=1e-10)
    np.testing.assert_allclose(res["alpha"], o.inv_Kxx_y, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(res["mean"], o.mean(Xo), rtol=1e-8, atol=1e-11)


def test_single_rank_schedule_matches_oracle():
    N, d, m = 700, 3, 40
    X, y, Xo = orc.synth_inputs(N, d, m)
    ops = CpuOps()
    g = DistributedGP(ops, LocalComm(), N, d, nb=128)
    g.set_data(X, y)
    params = np.array([1.0, 0.5 * np.sqrt(d)])
    llh = g.fit(params, 1.0)
    out = ops.empty((m,))
    g.mean(ops.from_host(Xo), m, params, out)
    _check({"info": g.info_host, "log_lh": llh, "alpha": ops.to_host(g.alpha), "mean": ops.to_host(out)},
           N, d, m)


@pytest.mark.parametrize("world,N,nb", [(2, 900, 128), (3, 1000, 64), (2, 513, 256)])
def test_gloo_world_matches_oracle(tmp_path, world, N, nb):
    res = run_world(world, "gloo", False, N, 2, nb, 33, str(tmp_path))
    _check(res, N, 2, 33)


def test_gloo_world4_scatter_allgather_panel_broadcast(tmp_path, monkeypatch):
    # the opt-in panel broadcast (root scatters 1/P to every peer, then all-gather) must give the same
    # factorisation: 4 ranks, every broadcast (panels and the solve's block vectors) through it
    monkeypatch.setenv("GPX_DIST_BCAST", "sag")
    monkeypatch.setenv("GPX_DIST_SAG_MIN", "1")
    res = run_world(4, "gloo", False, 1024, 2, 128, 20, str(tmp_path))
    _check(res, 1024, 2, 20)


def test_info_reduction_and_minus_inf():
    # a failed panel (LAPACK-style info > 0) must surface as info_host and log_lh = -inf
    N, d = 300, 2
    X, y, _ = orc.synth_inputs(N, d, 4)
    ops = CpuOps()
    g = DistributedGP(ops, LocalComm(), N, d, nb=128)
    g.set_data(X, y)
    assert np.isfinite(g.fit(np.array([1.0, 1.0]), 1.0)) and g.info_host == 0
    g.info[0] = 137
    g.reduce_scalars()
    assert g.info_host == 137 and g.log_lh == -np.inf


def _mlii_worker(rank, world, port, outdir):
    import os
    import torch.distributed as dist
    from gaussian_processes_amd import mlii, _lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # no GPU here: rows are all invalid (-> nan without touching the device) except that
        # the dealing / all-reduce / code path for nan must round-trip on every rank
        thetas = np.array([[1.0, 0.0, 1.0], [0.0, 1.0, 1.0], [1.0, 1.0, -1.0], [np.nan, 1.0, 1.0], [1.0, -3.0, 0.1]])
        try:
            out = mlii.log_lh_batch(np.zeros(8), np.zeros(8), thetas, dist=dist)
            np.save(os.path.join(outdir, "mlii_%d.npy" % rank), out)
        except _lib.GpxError:
            np.save(os.path.join(outdir, "mlii_%d.npy" % rank), np.array([-1.0]))
    finally:
        dist.destroy_process_group()


def test_mlii_round_robin_over_gloo(tmp_path):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_mlii_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        out = np.load(str(tmp_path / ("mlii_%d.npy" % r)))
        # without a GPU the handle cannot be created: the product fails loudly (no CPU fallback)
        assert (out.shape == (1,) and out[0] == -1.0) or np.isnan(out).all()
