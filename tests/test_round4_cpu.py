"""Round-4 host-side tests (no GPU): the panel-broadcast arithmetic of the C multi-GPU schedule at full size, and
the benchmark launcher's watchdog."""
import ctypes
import os
import shutil
import subprocess
import sys
import threading
import time

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
