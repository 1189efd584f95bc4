"""Round-4 host-side tests (no GPU): the panel-broadcast arithmetic of the C multi-GPU schedule at full size, and
the benchmark launcher's watchdog."""
import ctypes
import os
import shutil
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

from gaussian_processes_amd import _lib, multi_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plan(n, nb, world, chunks, j):
    lib = _lib.load()
    out = (ctypes.c_int64 * (6 * 32))()
    k = lib.gpx_debug_mg_plan(n, nb, world, chunks, j, out, 6 * 32)
    assert k >= 1, _lib.last_error()
    return [tuple(out[6 * i:6 * i + 6]) for i in range(k)]


@pytest.mark.parametrize("n", [65536, 65536 + 300, 32768, 8492])
@pytest.mark.parametrize("nb", [256, 512, 1024])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_mg_panel_broadcast_plan_alignment_and_coverage(n, nb, world):
    """mg_chunk_plan / mg_piece of csrc/gpx_mg.hip (what mg_factor_and_bcast and mg_bcast_panel execute), for every
    panel of a full-size problem: the row chunks tile the panel (rider row included) exactly once and start on
    128-row boundaries, the first chunk holds the next block column's diagonal rows, the scatter + all-gather pieces
    tile each chunk exactly once with 32-element alignment (128 / 256 bytes), the last piece takes the rest and is
    the only one that may be longer, and eligibility is what the kernel-side condition says."""
    nblk = -(-n // nb)
    for chunks in (1, 4, 16):
        for j in range(nblk):
            rows = n + 1 - j * nb
            plan = _plan(n, nb, world, chunks, j)
            assert len(plan) <= max(1, chunks)
            done = 0
            for c, (b, e, count, piece, last, sag) in enumerate(plan):
                assert b == done and e > b, (j, plan)
                assert b % 128 == 0, (j, plan)                      # chunk starts: 128-row aligned (byte offset b * nb * es)
                assert count == (e - b) * nb
                assert piece % 32 == 0 and piece * (world - 1) + last == count
                assert last >= piece and last - piece < 32 * world + nb * 128, (j, plan)
                assert sag == (1 if (world > 2 and piece >= 1024) else 0)
                done = e
            assert done == rows, (j, plan)                          # coverage, rider row included
            if len(plan) > 1:
                assert plan[0][1] >= min(rows, max(2 * nb, 1024))   # the next panel's B-operand rows travel first


@pytest.mark.parametrize("nb", [512, 1024])
def test_mg_panel_broadcast_plan_at_n262144_world8(nb):
    """The same walk at N = 262144, P = 8 (round 6): north_star's "N where K + factor exceed one GPU's HBM" -- 550 GB in
    fp64, 69 GB a rank.  A panel is up to 262145 x 1024 elements = 2.1 GB: every count and offset is checked as a 64-bit
    quantity (a 32-bit BYTE offset would wrap in the first panel's second chunk)."""
    n, world = 262144, 8
    nblk = -(-n // nb)
    for chunks in (4, 16):
        for j in list(range(0, nblk, 7)) + [nblk - 2, nblk - 1]:
            rows = n + 1 - j * nb
            plan = _plan(n, nb, world, chunks, j)
            done = 0
            for (b, e, count, piece, last, sag) in plan:
                assert b == done and e > b and b % 128 == 0
                assert count == (e - b) * nb and count * 8 < 2 ** 40
                assert piece % 32 == 0 and piece * (world - 1) + last == count and last >= piece
                done = e
            assert done == rows, (j, plan)
    j0 = _plan(n, nb, world, 4, 0)
    assert sum(c[2] for c in j0) * 8 >= 2 ** 30      # a panel of 1 - 2 GB: the later chunks' byte offsets do not fit 30 bits


def test_mg_plan_rejects_bad_arguments():
    lib = _lib.load()
    out = (ctypes.c_int64 * 6)()
    assert lib.gpx_debug_mg_plan(1000, 256, 4, 4, 4, out, 6) < 0        # panel 4 does not exist (4 * 256 >= 1000)
    assert lib.gpx_debug_mg_plan(65536, 512, 8, 4, 0, out, 6) < 0       # four chunks do not fit six values


def test_watchdog_fires_and_cancels():
    fired = []
    wd = multi_gpu.Watchdog(0.2, "a phase that hangs", rank=3, on_expire=lambda: fired.append(time.monotonic()))
    time.sleep(0.6)
    assert len(fired) == 1
    wd.cancel()
    fired2 = []
    with multi_gpu.Watchdog(5.0, "a phase that finishes", on_expire=lambda: fired2.append(1)):
        time.sleep(0.05)
    time.sleep(0.1)
    assert not fired2
    # 0 disables it
    multi_gpu.Watchdog(0, "disabled").cancel()


def test_watchdog_ends_a_hung_process_with_its_status():
    """A phase blocked in a call that never returns (here: a lock nobody releases, standing in for ncclCommInitRank
    without its peers): the process exits with WATCHDOG_EXIT and says which phase on stderr."""
    code = ("import sys, threading; sys.path.insert(0, %r)\n"
            "from gaussian_processes_amd import multi_gpu\n"
            "with multi_gpu.Watchdog(0.5, 'communicator set-up (test)', rank=1):\n"
            "    threading.Lock().acquire(); threading.Lock().acquire(); l = threading.Lock(); l.acquire(); l.acquire()\n"
            "print('not reached')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == multi_gpu.WATCHDOG_EXIT, (r.returncode, r.stderr[-500:])
    assert "WATCHDOG rank 1" in r.stderr and "communicator set-up (test)" in r.stderr
    assert "not reached" not in r.stdout


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_hazard_and_leaf_probes_still_build_for_gfx950(tmp_path):
    """tools/mfma_hazard_probe_gen.py (the measured wait-state table the asm MFMAs of csrc/gpx_leaf.h rely on) and
    tools/leaf_probe.hip (the leaf alone against a host Cholesky) are the evidence behind DESIGN section 3.2c: they must keep
    generating / compiling against the current gpx_leaf.h (cross-compilation, no GPU needed)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    gen = subprocess.run([sys.executable, os.path.join(root, "tools", "mfma_hazard_probe_gen.py")], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=120)
    assert gen.returncode == 0, gen.stderr[-1000:]
    for src, extra in ((os.path.join(root, "tools", "mfma_hazard_probe.hip"), ["-O2"]),
                       (os.path.join(root, "tools", "leaf_probe.hip"), ["-O3", "-std=c++17", "-I" + os.path.join(root, "gaussian_processes_amd", "csrc")]),
                       (os.path.join(root, "tools", "leaf_probe.hip"), ["-O3", "-std=c++17", "-DPROBE_RING=4", "-DSTAMP=5",
                                                                         "-I" + os.path.join(root, "gaussian_processes_amd", "csrc")])):
        out = str(tmp_path / (os.path.basename(src) + ".o"))
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-Wno-unused-value", "-Wno-unused-result", "-c", src, "-o", out] + extra,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]


def test_mlii_optimize_lock_step_rendezvous_with_a_host_evaluator():
    """mlii.optimize's rendezvous (round 6) without a GPU: R L-BFGS-B optimisers, each on its own thread, are served by
    ONE batched evaluation per step made on the calling thread; optimisers that finish early leave and the batch
    shrinks; a failing evaluation reaches the caller instead of hanging the threads.  The evaluator is a host stand-in
    with a known maximum; the walls (non-finite rows) behave as the -inf rows of a real table do."""
    import threading
    from gaussian_processes_amd import mlii
    centre = np.array([1.3, 2.0, 0.7])

    class Fake(object):
        def __init__(self):
            self.sizes, self.threads = [], set()

        def value_and_grad(self, thetas, clamp=True):
            t = np.atleast_2d(thetas)
            self.sizes.append(t.shape[0])
            self.threads.add(threading.get_ident())
            u = np.log(t) - np.log(centre)
            val = -(u ** 2).sum(1) * np.array([1.0, 3.0, 0.5]).sum()
            grad = -2.0 * u / t * np.array([1.0, 3.0, 0.5]).sum()
            wall = t[:, 2] > 50.0                                  # a region that "is not positive definite"
            val = np.where(wall, -np.inf, val)
            grad[wall] = np.nan
            return val, grad

    rs = np.random.RandomState(3)
    th0 = np.exp(rs.uniform(-1.5, 1.5, (7, 3)))
    th0[5, 2] = 80.0                                               # starts on the wall: stays where it is
    ev = Fake()
    res = mlii.optimize(None, None, th0, evaluator=ev, maxiter=40)
    assert ev.threads == {threading.get_ident()}                   # only the caller ever evaluates
    assert ev.sizes[0] == 7 and ev.sizes[-1] == 7 and max(ev.sizes) == 7 and min(ev.sizes) >= 1
    assert sorted(set(ev.sizes[1:-1]), reverse=True)[0] <= 7 and len(ev.sizes) == res["batched_calls"]
    fin = np.isfinite(res["log_lh0"])
    assert (res["log_lh"][fin] >= res["log_lh0"][fin]).all()
    np.testing.assert_allclose(res["theta"][fin], np.tile(centre, (fin.sum(), 1)), rtol=1e-4)
    np.testing.assert_array_equal(res["theta"][5], th0[5])
    assert res["best"] in np.nonzero(fin)[0]

    class Broken(Fake):
        def value_and_grad(self, thetas, clamp=True):
            if len(self.sizes) >= 2:
                raise MemoryError("out of HBM (test)")
            return Fake.value_and_grad(self, thetas, clamp)

    with pytest.raises((MemoryError, RuntimeError)):
        mlii.optimize(None, None, th0[:3], evaluator=Broken(), maxiter=10)
