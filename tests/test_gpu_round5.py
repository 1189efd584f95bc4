"""Round-5 GPU tests (``-m gpu``), all through the C ABI:

  * the TALL panel route (csrc/gpx_panel.hip, winv256_kernel): the rows below a 256-column panel's diagonal block as
    products with inv(L_256) on the GEMM kernel -- forced at sizes the oracle finishes in seconds (with and without the
    folded left update, ragged sizes, the riding right-hand side, fp64 / fp32) against the oracle's factor
    (scipy.linalg.cholesky, reference gp/gp.py:294) and against the resident route; at N = 20480 with its natural
    threshold against the resident route and sampled oracle kernel rows.  (Opt-in: measured neutral at N = 65536.)
"""
import ctypes
import os

import numpy as np
import pytest

import gaussian_processes_amd as gp
from gaussian_processes_amd import _lib
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu

# ------------------------------------------------------------------------- the asm leaf's self-check --
def test_asm_leaf_self_check_passes_on_this_device(monkeypatch):
    """The asm-scheduled leaf (csrc/gpx_leaf.h: wait states MEASURED on gfx950) is compared, on the device the process
    runs on, with the compiler-scheduled leaf before its first use -- alone and beside a product that loads every matrix
    pipe.  It must pass here (state 1); a fit then takes the asm leaf, and GPX_LEAF=1 still selects the other one: both
    against the oracle."""
    lib = _lib.load()
    st = ctypes.c_int(-1)
    _lib.check(lib.gpx_debug_leaf_selfcheck(1, ctypes.byref(st)))
    assert st.value == 1, st.value
    N, d = 1536, 3
    X, y, _ = orc.synth_inputs(N, d, 4)
    o = orc.OracleGP("gaussian", (1.0, 0.9), X, y, 0.7)
    out = {}
    for leaf in (None, "1", "5"):
        if leaf is None:
            monkeypatch.delenv("GPX_LEAF", raising=False)
        else:
            monkeypatch.setenv("GPX_LEAF", leaf)
        g = gp.GP(gp.GaussianKernel(1.0, 0.9), X, y, s=0.7)
        out[leaf] = float(g.log_lh)
        np.testing.assert_allclose(out[leaf], o.log_lh, rtol=1e-10)
        np.testing.assert_allclose(np.tril(g.Lxx), o.Lxx, rtol=1e-9, atol=1e-12)


def test_four_host_threads_factor_n8192_concurrently():
    """Every single-matrix panel of up to 8192 rows runs the one-workgroup-per-CU instantiation (LV = 4): with four host
    threads factoring n = 8192 at once, 4 x 128 resident workgroups that each want a whole CU compete for 256 CUs and spin
    on each other's flags.  Progress rests on in-order dispatch within a launch (a consumer is only ever placed after its
    producers); a workgroup left unplaced past the spin bound would surface as info = -7 -> GPX_ERR_INTERNAL.  Six rounds of
    four concurrent fits: every one succeeds and reproduces the single-threaded log_lh bit for bit."""
    import threading
    N, d = 8192, 8
    X, y, _ = orc.synth_inputs(N, d, 4)
    params, s = (1.0, 0.5 * np.sqrt(d)), 1.0

    def one_fit():
        g = gp.GP(gp.GaussianKernel(*params), X, y, s=s)
        return float(g.log_lh)

    ref = one_fit()
    assert np.isfinite(ref)
    res, errs = [], []

    def worker():
        try:
            for _ in range(6):
                res.append(one_fit())
        except Exception as e:                                   # noqa: BLE001 -- reported below
            errs.append(repr(e))

    ts = [threading.Thread(target=worker) for _ in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert len(res) == 24 and all(v == ref for v in res), (ref, sorted(set(res)))


def test_checkpoint_file_mode_follows_the_umask(tmp_path):
    """gpx_gp_save writes through mkstemp (0600) and then gives the file what fopen would have given it: 0666 & ~umask."""
    X, y, _ = orc.synth_inputs(300, 2, 4)
    g = gp.GP(gp.GaussianKernel(1.0, 1.0), X, y, s=0.5)
    old = os.umask(0o077)
    try:
        p1 = tmp_path / "a.gpx"
        g.save_fitted(p1)
        assert (os.stat(p1).st_mode & 0o777) == 0o600
        os.umask(0o022)
        p2 = tmp_path / "b.gpx"
        g.save_fitted(p2)
        assert (os.stat(p2).st_mode & 0o777) == 0o644
    finally:
        os.umask(old)
    g2 = gp.GP.load_fitted(p2)
    assert float(g2.log_lh) == float(g.log_lh)


# ------------------------------------------------------------------------------- multi-GPU rehearsal --
@pytest.mark.parametrize("world,rank,nb,sag", [(4, 1, 256, 0), (8, 7, 256, 1), (2, 0, 512, 0)])
def test_rehearsal_of_one_rank_reproduces_its_block_columns(world, rank, nb, sag):
    """gpx_mg_create_rehearsal: rank `rank` of a `world`-rank run in this one process -- its own panels, packs and updates
    through the product's C schedule, the panels it does not own copied out of a resident single-GPU factor of the same
    matrix behind modelled transfer delays.  What the rank computes must be its block columns of that factor (and the
    solution the single-GPU fit's), the timing must carry the exposed-wait and modelled-transfer classes."""
    from gaussian_processes_amd import multi_gpu
    N, d = 6000, 3
    X, y, _ = orc.synth_inputs(N, d, 4)
    params, s = np.array([1.0, 0.5 * np.sqrt(d)]), 1.0
    out = multi_gpu.rehearse_rank(N, d, rank, world, X, y, params, s, nb=nb, chunks=4, sag=sag, fits=2, link_GBps=100.0)
    assert out["check"]["owned_columns_vs_resident_factor_max_rel_err"] < 1e-10, out["check"]
    assert out["check"]["alpha_vs_single_gpu_max_rel_err"] < 1e-7, out["check"]
    o = orc.OracleGP("gaussian", params, X, y, s)
    np.testing.assert_allclose(out["single_gpu_log_lh"], o.log_lh, rtol=1e-10)
    last = out["fits"][-1]
    assert last["modelled_transfer"] > 0 and last["exposed_wait"] >= 0 and last["chain_update"] > 0
    assert out["owned_panels"] == len([j for j in range(-(-N // nb)) if j % world == rank])
    assert out["per_step_ms"]["own_chain_per_owned_panel_mean"] > 0


def test_one_rccl_communicator_serves_every_width(monkeypatch):
    """bench.py's tuning pass tries several block-column widths; a new width is a new local layout, i.e. a new handle -- but
    NOT a new communicator: the next handle takes over the previous one's (gpx_mg_adopt_comm: one ncclCommInitRank per
    process).  Three widths in a row on one RCCL communicator (one rank, every collective forced to be a real RCCL call),
    each fitting the same data to the oracle's log_lh, with the chunk count and the broadcast form changed between fits
    as the tuning pass does; the handle that gave its communicator away can still be closed."""
    from gaussian_processes_amd import multi_gpu
    monkeypatch.setenv("GPX_FORCE_COLLECTIVES", "1")
    N, d = 3000, 3
    X, y, Xo = orc.synth_inputs(N, d, 8)
    params, s = np.array([1.0, 0.5 * np.sqrt(d)]), 1.0
    o = orc.OracleGP("gaussian", params, X, y, s)
    prev = None
    try:
        for nb in (512, 256, 1024):
            mg = multi_gpu.NativeDistributedGP(N, d, nb=nb, backend="rccl", device=0, adopt_from=prev)
            if prev is not None:
                prev.close()
            prev = mg
            assert mg.comm_info()["rccl_nranks"] == 1
            mg.set_data(X, y)
            for chunks, sag in ((2, 0), (8, 1), (4, 0)):
                mg.set_chunks(chunks)
                mg.set_bcast(sag)
                np.testing.assert_allclose(mg.fit(params, s), o.log_lh, rtol=1e-10)
            tm = mg.timing(extended=True)
            assert tm["exposed_wait"] >= 0 and tm["modelled_transfer"] == 0 and tm["factor"] > 0
            np.testing.assert_allclose(mg.mean(params, Xo), o.mean(Xo), rtol=1e-8, atol=1e-11)
    finally:
        if prev is not None:
            prev.close()


def test_bench_tuning_pass_over_one_rccl_rank_adopts_the_communicator():
    """The flow the first real multi-GPU run will take, as far as one GPU can take it: bench.py's distributed leg over a
    real RCCL communicator (one rank, every collective forced), the tuning pass forced on: three block-column widths x
    three chunk counts, every new width a new handle that takes over the ONE communicator; the line carries the table,
    the chosen triple is the table's best, the result is the oracle's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GPX_MG_BCAST", "GPX_BENCH_NO_TUNE", "GPX_POTRF_NB",
              "GPX_MG_BCAST_CHUNKS", "GPX_DIST_BACKEND"):
        env.pop(k, None)
    env.update(GPX_BENCH_FORCE_DIST="1", GPX_FORCE_COLLECTIVES="1", GPX_BENCH_FORCE_TUNE="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29641")
    N, d = 6144, 4
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--problem-n", str(N), "--problem-d", str(d),
                        "--problem-m", "64", "--steps", "1", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][-1])
    assert out["rccl_nranks"] == 1
    tune = out["schedule_autotune"]
    assert tune["measured"] == 9 and {row["nb"] for row in tune["table"]} == {256, 512, 1024}
    best = min(tune["table"], key=lambda row: (row["fit_s"], row["nb"], row["chunks"], row["sag"]))
    assert tune["chosen"] == {k: best[k] for k in ("nb", "chunks", "sag")}
    assert ("nb=%d" % tune["chosen"]["nb"]) in out["config"]["parallelism"]
    X, y, _ = orc.synth_inputs(N, d, 64)
    o = orc.OracleGP("gaussian", (1.0, 0.5 * np.sqrt(d)), X, y, 1.0)
    np.testing.assert_allclose(out["log_lh"], o.log_lh, rtol=1e-10)
    assert out["stage_and_chain_ms_per_rank"][0]["exposed_wait"] >= 0


@pytest.mark.parametrize("owner_first", ["0", "1"])
def test_native_world3_with_and_without_owner_first(tmp_path, monkeypatch, owner_first):
    """The owner-first schedule (the owner of the next panel factors it before its own trailing update; default on) and the
    schedule without it: three ranks sharing GPU 0 over host callbacks, against the oracle -- the switch only moves a wait,
    both must give the same factorisation."""
    from _dist_helpers import run_native_world
    monkeypatch.setenv("GPX_MG_OWNER_FIRST", owner_first)
    N, d, m = 4100, 3, 24
    res = run_native_world(3, N, d, 256, m, str(tmp_path))
    X, y, Xo = orc.synth_inputs(N, d, m)
    o = orc.OracleGP("gaussian", (1.0, 0.5 * np.sqrt(d)), X, y, 1.0)
    assert int(res["info"]) == 0
    np.testing.assert_allclose(float(res["log_lh"]), o.log_lh, rtol=1e-10)
    np.testing.assert_allclose(float(res["log_lh2"]), float(res["log_lh"]), rtol=0, atol=0)
    np.testing.assert_allclose(res["alpha"], o.inv_Kxx_y, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(res["mean"], o.mean(Xo), rtol=1e-8, atol=1e-11)
