"""world_size > 1 on the CPU (gloo): the HOST logic of the multi-GPU path -- the block-cyclic maps, the agreement protocol
with which the ranks of a run choose the schedule's free parameters together (multi_gpu.tune_schedule: block-column width
x row chunks x broadcast form, a candidate's time = the slowest rank's, one budget clock for all), and the round-robin
dealing of an ML-II table.  The schedule itself is C behind the ABI (csrc/gpx_mg.hip) and runs under -m gpu with 2 / 3 / 4
process-ranks and 8 thread-ranks on one GPU; its chunk / piece arithmetic is walked on the CPU in tests/test_round4_cpu.py.
(Rounds 1 - 4 also kept a Python restatement of the schedule here; it described a solve the product no longer has.)"""
import json
import os

import numpy as np
import pytest

from gaussian_processes_amd import multi_gpu
from gaussian_processes_amd.multi_gpu import BlockCyclic


def test_block_cyclic_maps():
    lay = BlockCyclic(1000, 128, 3, 1)
    assert lay.nblk == 8 and lay.my_blocks == [1, 4, 7]
    assert lay.owner(5) == 2 and lay.local_col(4) == 128 and lay.kb(7) == 1000 - 7 * 128
    assert lay.first_local_block_after(1) == 1 and lay.first_local_block_after(7) is None
    cover = sorted(j for r in range(3) for j in BlockCyclic(1000, 128, 3, r).my_blocks)
    assert cover == list(range(8))


def test_schedule_candidates_order_and_limits():
    c8 = multi_gpu.schedule_candidates(65536, 8)
    assert len(c8) == 3 * 2 * 3 and c8[0] == {"nb": 256, "chunks": 2, "sag": 0}
    assert [c["nb"] for c in c8] == sorted(c["nb"] for c in c8)          # grouped by width: one handle per group
    c2 = multi_gpu.schedule_candidates(65536, 2)
    assert len(c2) == 3 * 3 and {c["sag"] for c in c2} == {0}            # scatter + all-gather needs more than two ranks
    assert multi_gpu.schedule_candidates(1000, 8, nbs=(256, 512)) == []  # fewer panels than ranks: nothing to tune
    pinned = multi_gpu.schedule_candidates(65536, 4, nbs=(512,), chunks=(4,))
    assert pinned == [{"nb": 512, "chunks": 4, "sag": 0}, {"nb": 512, "chunks": 4, "sag": 1}]


def test_tune_schedule_single_rank_budget_and_tie_break():
    cands = multi_gpu.schedule_candidates(65536, 4, nbs=(512, 1024), chunks=(2, 4))
    times = {(512, 2, 0): 0.5, (512, 4, 0): 0.4, (512, 2, 1): 0.4, (512, 4, 1): 0.6,
             (1024, 2, 0): 0.3, (1024, 4, 0): 0.3, (1024, 2, 1): 0.9, (1024, 4, 1): 0.9}
    now = [0.0]

    def measure(c):
        t = times[(c["nb"], c["chunks"], c["sag"])]
        now[0] += t + 1.0                                   # (a fit and a second of set-up)
        return t

    best, table = multi_gpu.tune_schedule(cands, measure, budget_s=1e9, clock=lambda: now[0])
    assert len(table) == 8 and best == {"nb": 1024, "chunks": 2, "sag": 0}     # tie: the smaller chunk count
    now[0] = 0.0
    best, table = multi_gpu.tune_schedule(cands, measure, budget_s=4.0, clock=lambda: now[0])
    # the budget is looked at AFTER a candidate: 1.5, 2.9, 4.3 > 4.0 -> three measured, the best of those
    assert len(table) == 3 and best == {"nb": 512, "chunks": 2, "sag": 1}      # (0.4 twice: the smaller chunk count)
    now[0] = 0.0
    best, table = multi_gpu.tune_schedule(cands, measure, budget_s=0.0, clock=lambda: now[0])
    assert len(table) == 1 and best == {"nb": 512, "chunks": 2, "sag": 0}      # at least one candidate, always


def _tune_worker(rank, world, port, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cands = multi_gpu.schedule_candidates(65536, 4, nbs=(512, 1024), chunks=(2, 4))
        # rank-dependent fit times: rank 0 alone would pick (512, 2, 0), the last rank alone (1024, 4, 1); the slowest
        # rank's time decides: max over ranks is smallest for (512, 4, 1)
        def fit_time(c, r):
            key = (c["nb"], c["chunks"], c["sag"])
            base = {(512, 2, 0): 0.20, (512, 4, 0): 0.50, (512, 2, 1): 0.45, (512, 4, 1): 0.30,
                    (1024, 2, 0): 0.60, (1024, 4, 0): 0.55, (1024, 2, 1): 0.70, (1024, 4, 1): 0.65}[key]
            slow = {(512, 2, 0): 0.80, (512, 4, 0): 0.52, (512, 2, 1): 0.47, (512, 4, 1): 0.31,
                    (1024, 2, 0): 0.61, (1024, 4, 0): 0.56, (1024, 2, 1): 0.71, (1024, 4, 1): 0.25}[key]
            return base if r == 0 else (slow if r == world - 1 else min(base, slow))
        # every rank has its OWN clock (rank r's seconds run (1 + r) times as fast): the budget is checked against the
        # slowest rank's, so all ranks stop after the same candidate
        now = [0.0]
        def measure(c):
            t = fit_time(c, rank)
            now[0] += (1 + rank) * 1.0
            return t
        full = multi_gpu.tune_schedule(cands, measure, dist=dist, budget_s=1e9, clock=lambda: now[0])
        now[0] = 0.0
        cut = multi_gpu.tune_schedule(cands, measure, dist=dist, budget_s=2.5 * world, clock=lambda: now[0])
        with open(os.path.join(outdir, "tune_%d.json" % rank), "w") as f:
            json.dump({"full": full, "cut": cut}, f)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tune_schedule_ranks_agree_over_gloo(tmp_path, world):
    """world_size 2 and 3 over gloo: ranks whose own measurements (and whose own clocks) disagree still measure the same
    candidates, stop at the same one and choose the same triple -- the one whose SLOWEST rank is fastest."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_tune_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [json.load(open(str(tmp_path / ("tune_%d.json" % r)))) for r in range(world)]
    for o in outs[1:]:
        assert o == outs[0]                                  # the same table, the same choice, on every rank
    best, table = outs[0]["full"]
    assert len(table) == 8 and best == {"nb": 512, "chunks": 4, "sag": 1}
    row = [r for r in table if (r["nb"], r["chunks"], r["sag"]) == (512, 2, 0)][0]
    assert row["fit_s"] == 0.8                               # rank 0 measured 0.20: the slowest rank's time is what counts
    best_cut, table_cut = outs[0]["cut"]
    # the slowest clock runs `world` seconds per candidate; budget 2.5 * world -> three candidates, everywhere
    assert len(table_cut) == 3 and best_cut == {"nb": 512, "chunks": 2, "sag": 1}


def _mlii_worker(rank, world, port, outdir):
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
