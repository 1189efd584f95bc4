"""Round 6 (``-m gpu``): orders ABOVE the headline's N = 65536, the batched value + gradient of ML-II and its optimiser,
the pipelined lock-step batch.

`scipy.linalg.cholesky` (gp/gp.py:294) takes any n; until this round no fit above N = 65536 had been run here, the exact
tile map of the trailing update stopped at 64 bands of 1024 rows, and the 64-bit index paths north_star's "N where
K + factor exceed one GPU's HBM" depends on had never been exercised.  N = 131072, d = 32, fp64 is 137 GB of the 288 GB.

Tolerances (fp64): sampled-row residual rtol 1e-9 / atol 1e-10, log_lh identity rtol 1e-12, mean rtol 1e-9, factor rows
L[r] . L[c] = K[r, c] rtol 1e-11 (as tests/test_gpu_configs.py at N = 65536).
"""
import ctypes
import gc
import time

import numpy as np
import pytest

import gaussian_processes_amd as gp
from gaussian_processes_amd import _lib
from oracle import gp_oracle as orc
from test_gpu_configs import _check_factor_rows, _device_diag, _krows

pytestmark = pytest.mark.gpu


def test_n131072_fp64_one_gpu_and_the_rccl_schedule(monkeypatch):
    """N = 131072, d = 32, fp64 (twice the headline's order; 128 bands of the exact tile map, row offsets beyond 2^32
    bytes everywhere): through `gp.GP` against oracle kernel rows (K alpha = y on sampled rows, the log_lh identity from a
    device-side fetch of diag(L), the mean on 8 test points, L[r] . L[c] = K[r, c] on sampled rows spread over all
    sixteen 8192-row bands), then through the C multi-GPU schedule with ONE RCCL rank and every collective forced
    (nb = 1024), the two routes against each other."""
    from gaussian_processes_amd import multi_gpu
    if _lib.device_info(0)["hbm_bytes"] < 200e9:
        pytest.skip("needs ~140 GB of HBM")
    N, d, m = 131072, 32, 64
    X, y, Xo = orc.synth_inputs(N, d, m)
    h, w, s = 1.0, 0.5 * np.sqrt(d), 1.0
    rows = np.unique(np.array([0, 1, 1023, 1024, 65535, 65536, 65537, 66559, 66560, 98303, 98304, 100000, N // 3,
                               N - 1025, N - 1024, N - 2, N - 1]))
    Krows = _krows(X, rows, h, w, s)
    Ko = orc.kernel_matrix("gaussian", "K", Xo[:8], X, (h, w))
    _lib.route_reset()
    t0 = time.perf_counter()
    g = gp.GP(gp.GaussianKernel(h, w), X, y, s=s)
    alpha = g.inv_Kxx_y
    t_fit = time.perf_counter() - t0
    assert _lib.route_count(_lib.ROUTE_SYRK_EXACT) > 0 and _lib.route_count(_lib.ROUTE_SYRK_PATCH) == 0
    np.testing.assert_allclose(Krows @ alpha, y[rows], rtol=1e-9, atol=1e-10)
    dg = _device_diag(g)
    assert (dg > 0).all()
    llh = float(g.log_lh)
    np.testing.assert_allclose(llh, -0.5 * y @ alpha - np.log(dg).sum() - 0.5 * N * np.log(2 * np.pi), rtol=1e-12)
    np.testing.assert_allclose(g.mean(Xo)[:8], Ko @ alpha, rtol=1e-9, atol=1e-11)
    _check_factor_rows(g, X, h, w, s, [0, 1024, 8191, 20000, 32768, 49151, 65535, 65536, 70001, 81920, 98303, 98304, 110000,
                                       N - 1025, N - 2, N - 1], rtol=1e-11, atol=1e-12)
    print("N=131072 fit + alpha through gp.GP: %.2f s (first call, allocation included)" % t_fit)
    del g
    gc.collect()

    monkeypatch.setenv("GPX_FORCE_COLLECTIVES", "1")
    params = np.array([h, w])
    mg = multi_gpu.NativeDistributedGP(N, d, nb=1024, backend="rccl", device=0)
    try:
        mg.set_data(X, y)
        llh_mg = mg.fit(params, s)
        assert mg.info == 0
        alpha_mg = mg.alpha
        np.testing.assert_allclose(Krows @ alpha_mg, y[rows], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(llh_mg, llh, rtol=1e-11)
        np.testing.assert_allclose(alpha_mg, alpha, rtol=1e-7, atol=1e-10)
    finally:
        mg.close()


# ------------------------------------------------------------------------- batched value + gradient, optimiser --
def _draws(d, count):
    """SURVEY 8(d) config-5 draws in the library's column order (h, w, s)."""
    rs = np.random.RandomState(2)
    w = rs.uniform(0.25, 2, 64) * np.sqrt(d)
    h = rs.uniform(0.5, 2, 64)
    s = rs.uniform(0.5, 2, 64)
    return np.column_stack([h, w, s])[:count]


def _raw_llh(o):
    """The oracle's log marginal likelihood from its own Cholesky factor, WITHOUT the reference's logdet < MIN clamp."""
    L, a = o.Lxx, o.inv_Kxx_y
    return float(-0.5 * o.y @ a - np.log(np.diag(L)).sum() - 0.5 * o.y.size * np.log(2 * np.pi))


def test_fit_batch_grad_vs_oracle_and_central_differences():
    """gpx_gp_fit_batch_grad (gp/gp.py:398-433 `dloglh_dtheta` + gp_c.pyx:34-49 for a whole table of restarts, on the
    lock-step factorisation): 8 SURVEY-8(d) draws at N = 2048, d = 8 against the ORACLE's dense-numpy dloglh_dtheta
    (rtol 1e-7), against central differences of the batched (unclamped) log_lh itself (rtol 2e-5: the reference's own
    check, gp/tests/test_gp.py:75-97), against the one-handle-at-a-time gradient (rtol 1e-10), and the conventions:
    a row the reference would refuse (w < EPS) is NaN in value and gradient and does not disturb its neighbours."""
    from gaussian_processes_amd import mlii
    N, d = 2048, 8
    X, y, _ = orc.synth_inputs(N, d, 4)
    th = _draws(d, 8)
    with mlii.BatchEvaluator(X, y) as ev:
        val_c, grad = ev.value_and_grad(th)                          # the reference's clamped value
        val, grad2 = ev.value_and_grad(th, clamp=False)
        np.testing.assert_array_equal(grad, grad2)
        np.testing.assert_array_equal(ev(th), val_c)                 # the value-only sweep is the same lock-step pass
        for i in range(8):
            o = orc.OracleGP("gaussian", (th[i, 0], th[i, 1]), X, y, th[i, 2])
            ref = np.asarray(o.dloglh_dtheta)
            np.testing.assert_allclose(grad[i], ref, rtol=1e-7, atol=1e-9 * np.abs(ref).max())
            np.testing.assert_allclose(val[i], _raw_llh(o), rtol=1e-10)
            if np.isfinite(val_c[i]):
                assert val_c[i] == val[i]
            else:
                assert np.isneginf(val_c[i]) and np.isneginf(float(o.log_lh))     # the reference's logdet < MIN clamp
        # central differences of the batched value: one call for all 8 x 3 x 2 perturbed rows
        rel = 1e-5
        tab = []
        for i in range(8):
            for k in range(3):
                for sgn in (-1.0, 1.0):
                    r = th[i].copy(); r[k] *= (1.0 + sgn * rel); tab.append(r)
        v, _ = ev.value_and_grad(np.array(tab), clamp=False)
        v = v.reshape(8, 3, 2)
        fd = (v[:, :, 1] - v[:, :, 0]) / (2 * rel * th)
        np.testing.assert_allclose(grad, fd, rtol=2e-5, atol=1e-6 * np.abs(grad).max())
        # a refused row between two good ones
        bad = th[:3].copy(); bad[1, 1] = 0.0
        vb, gb = ev.value_and_grad(bad)
        assert np.isnan(vb[1]) and np.isnan(gb[1]).all()
        np.testing.assert_array_equal(gb[[0, 2]], grad[[0, 2]])
        np.testing.assert_array_equal(vb[[0, 2]], val_c[[0, 2]])
    g1 = gp.GP(gp.GaussianKernel(th[3, 0], th[3, 1]), X, y, s=th[3, 2])
    np.testing.assert_allclose(grad[3], g1.dloglh_dtheta, rtol=1e-10)


def test_fit_batch_grad_periodic_and_fp32():
    """The same entry for the periodic kernel (d = 1, three kernel parameters) against the oracle, and in fp32 against
    the fp64 run (rtol 5e-3: the gradient is a difference of O(n) sums; SURVEY 8(d)'s fp32 tolerance for means is 1e-3)."""
    from gaussian_processes_amd import mlii
    x = np.sort(np.random.RandomState(3).uniform(-5, 5, 640))
    y = np.sin(x) + 0.05 * np.random.RandomState(4).randn(640)
    th = np.array([[1.1, 0.8, 2.3, 0.5], [0.7, 1.3, 3.1, 0.9], [1.5, 0.6, 1.7, 0.4]])
    with mlii.BatchEvaluator(x, y, kernel="periodic") as ev:
        val, grad = ev.value_and_grad(th, clamp=False)
    for i in range(3):
        o = orc.OracleGP("periodic", tuple(th[i, :3]), x, y, th[i, 3])
        np.testing.assert_allclose(grad[i], o.dloglh_dtheta, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(val[i], _raw_llh(o), rtol=1e-10)
    X, yy, _ = orc.synth_inputs(1536, 4, 4)
    t4 = _draws(4, 4)
    t4[:, 2] = np.maximum(t4[:, 2], 1.0)
    with mlii.BatchEvaluator(X, yy) as e64, mlii.BatchEvaluator(X, yy, dtype="float32") as e32:
        v64, g64 = e64.value_and_grad(t4, clamp=False)
        v32, g32 = e32.value_and_grad(t4, clamp=False)
    np.testing.assert_allclose(v32, v64, rtol=1e-4)
    np.testing.assert_allclose(g32, g64, rtol=5e-3, atol=5e-3 * np.abs(g64).max())


def test_mlii_optimize_improves_every_finite_restart():
    """mlii.optimize: L-BFGS-B from 8 SURVEY-8(d) draws at N = 2048, d = 8, all restarts in lock-step (one
    gpx_gp_fit_batch_grad call per step for every restart still running).  Every restart ends at least where it began,
    the finite ones strictly higher; the reported values are what the batched evaluator returns at the reported
    parameters; the best restart's gradient in log(theta) has shrunk by at least 10 x; far fewer batched calls than
    function evaluations."""
    from gaussian_processes_amd import mlii
    N, d = 2048, 8
    X, y, _ = orc.synth_inputs(N, d, 4)
    th0 = _draws(d, 8)
    res = mlii.optimize(X, y, th0, maxiter=12)
    assert res["theta"].shape == (8, 3) and (res["theta"] > 0).all()
    fin = np.isfinite(res["log_lh0"])
    assert fin.all()                                                   # unclamped values: every draw is positive definite
    assert (res["log_lh"] >= res["log_lh0"]).all() and (res["log_lh"][fin] > res["log_lh0"][fin]).all()
    assert res["batched_calls"] <= res["nfev"].max() + 3 and res["nfev"].sum() > 2 * res["batched_calls"]
    with mlii.BatchEvaluator(X, y) as ev:
        v, g = ev.value_and_grad(res["theta"], clamp=False)
        v0, g0 = ev.value_and_grad(th0, clamp=False)
    np.testing.assert_allclose(v, res["log_lh"], rtol=1e-12)
    b = res["best"]
    assert v[b] == v.max()
    assert np.abs(g[b] * res["theta"][b]).max() < 0.1 * np.abs(g0[b] * th0[b]).max()
    # the same answer as the reference's loop would give for the best restart: oracle value at the optimum
    o = orc.OracleGP("gaussian", (res["theta"][b, 0], res["theta"][b, 1]), X, y, res["theta"][b, 2])
    np.testing.assert_allclose(v[b], _raw_llh(o), rtol=1e-10)


# --------------------------------------------------------------------------------- the pair phase of the factorisation --
def _diag_any(g):
    """diag(L) from the handle's HBM matrix, fp64 or fp32 storage."""
    st = g._fit_pd()
    lib = _lib.load()
    A, lda = ctypes.c_void_p(), ctypes.c_int64()
    _lib.check(lib.gpx_gp_device_ptrs(st.handle, ctypes.byref(A), ctypes.byref(lda), None, None, None, None))
    es = 8 if g._dtype == _lib.F64 else 4
    out = np.empty(g._n, dtype=np.float64 if es == 8 else np.float32)
    _lib.check(lib.gpx_memcpy2d_d2h(out.ctypes.data_as(ctypes.c_void_p), es, A, (lda.value + 1) * es, es, g._n, None))
    return out.astype(np.float64)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_pair_phase_k2048_updates_vs_one_panel_per_update(monkeypatch, dtype):
    """potrf()'s pair phase (far trailing updates of depth K = 2048, one per TWO 1024-wide panels, both panels of the next
    pair factored on the side stream beside it; opt-in, GPX_POTRF_PAIR_ROWS: a shorter step at N = 65536 for a slower
    update kernel) forced at
    N = 13312 + 37 (ragged: the pairs, the exit step, then the tapering single-panel loop) against the default schedule of
    that size (one panel per update): both against oracle kernel rows (K alpha = y), the log_lh identity from diag(L), and
    each other (log_lh rtol 1e-12 / 1e-5: the same sums in a different association)."""
    N, d = 13312 + 37, 6
    X, y, Xo = orc.synth_inputs(N, d, 8)
    h, w, s = 1.0, 0.5 * np.sqrt(d), 1.0
    rows = np.unique(np.array([0, 1023, 1024, 2047, 2048, 4095, 4096, 5000, 8191, 8192, 12287, 12288, N - 2, N - 1]))
    Krows = _krows(X, rows, h, w, s)
    f64 = dtype == "float64"
    out = {}
    monkeypatch.setenv("GPX_POTRF_NB", "1024")             # (the taper would give 512-wide blocks at this size)
    monkeypatch.setenv("GPX_FIT_RIDE_MAX", "0")            # (no rider row: the pair phase is for sizes beyond it)
    for label, pr in (("pair", "2048"), ("single", "0")):
        monkeypatch.setenv("GPX_POTRF_PAIR_ROWS", pr)
        _lib.route_reset()
        g = gp.GP(gp.GaussianKernel(h, w), X, y, s=s, dtype=dtype)
        alpha = np.array(g.inv_Kxx_y, dtype=np.float64)
        assert _lib.route_count(_lib.ROUTE_POTRF_PAIR) == (1 if label == "pair" else 0), label
        np.testing.assert_allclose(Krows @ alpha, y[rows], rtol=1e-9 if f64 else 2e-3, atol=1e-10 if f64 else 2e-3, err_msg=label)
        dg = _diag_any(g)
        llh = float(g.log_lh)
        np.testing.assert_allclose(llh, -0.5 * y @ alpha - np.log(dg).sum() - 0.5 * N * np.log(2 * np.pi),
                                   rtol=1e-12 if f64 else 1e-5, err_msg=label)
        out[label] = (llh, alpha, np.array(g.mean(Xo)))
        del g
        gc.collect()
    np.testing.assert_allclose(out["pair"][0], out["single"][0], rtol=1e-12 if f64 else 1e-5)
    np.testing.assert_allclose(out["pair"][1], out["single"][1], rtol=1e-8 if f64 else 5e-3, atol=1e-11 if f64 else 5e-3)
    np.testing.assert_allclose(out["pair"][2], out["single"][2], rtol=1e-8 if f64 else 1e-3, atol=1e-11 if f64 else 1e-3)


def test_fit_batch_grad_chunked_equals_one_chunk_and_leaves_the_handle_alone(monkeypatch):
    """The table is processed in chunks of what fits in HBM (GPX_BATCH_MAX caps a chunk): chunks of 3 give the same values and
    gradients as one chunk of 8, bit for bit; the handle's own fitted state (a fit made before the sweep) is untouched, as
    gpx_gp_fit_batch promises; the value-only entry after a gradient sweep still agrees."""
    from gaussian_processes_amd import mlii
    N, d = 1536, 5                                         # (a multiple of 512: the gradient's lock-step group route)
    X, y, _ = orc.synth_inputs(N, d, 4)
    th = _draws(d, 8)
    with mlii.BatchEvaluator(X, y) as ev:
        lib = _lib.load()
        p0 = np.array([1.1, 0.9 * np.sqrt(d)])
        _lib.check(lib.gpx_gp_set_params(ev.h, _lib.dptr(p0), 0.8))
        _lib.check(lib.gpx_gp_fit(ev.h, None))
        v_own = ctypes.c_double(0.0)
        _lib.check(lib.gpx_gp_log_lh(ev.h, ctypes.byref(v_own)))
        monkeypatch.setenv("GPX_BATCH_MAX", "3")               # (first: an existing larger workspace would be reused whole)
        v3, g3 = ev.value_and_grad(th, clamp=False)
        monkeypatch.delenv("GPX_BATCH_MAX", raising=False)
        v1, g1 = ev.value_and_grad(th, clamp=False)
        np.testing.assert_array_equal(v1, v3)
        np.testing.assert_array_equal(g1, g3)
        monkeypatch.delenv("GPX_BATCH_MAX", raising=False)
        v_again = ctypes.c_double(0.0)
        _lib.check(lib.gpx_gp_log_lh(ev.h, ctypes.byref(v_again)))
        assert v_again.value == v_own.value
        o = orc.OracleGP("gaussian", (1.1, 0.9 * np.sqrt(d)), X, y, 0.8)
        np.testing.assert_allclose(v_own.value, float(o.log_lh), rtol=1e-10)
        np.testing.assert_allclose(ev(th), np.where(np.isfinite(ev(th)), v1, -np.inf))


# ------------------------------------------------------------- the fragment-register hazard of round 6, as a long soak --
@pytest.mark.parametrize("dtype,fold", [("float32", True), ("float32", False), ("float64", True)])
def test_long_soak_beside_a_neighbour(monkeypatch, dtype, fold):
    """1500 fits of n = 8192 (fp64: 600) beside a second host thread that factors n = 3000 in a loop, every log_lh and alpha
    equal to the first fit's bit for bit -- the folded route (the panel applies its predecessor itself) and the one with a
    block-column update per step.  An in-flight LDS read of the GEMM's k-loop used to land in a register its epilogue had
    been given about once in a thousand such fp32 fits ("The waits carry the fragment registers"; the 50
    neighbour fits of the round-4 soak caught it once in five suite runs)."""
    import threading
    from test_gpu_round4 import _Fit
    if not fold:
        monkeypatch.setenv("GPX_POTRF_FOLD_ROWS", "0")
    n, d = 8192, 3
    reps = 1500 if dtype == "float32" else 600
    X, y, _ = orc.synth_inputs(n, d, 4)
    params, s = np.array([1.0, 0.5 * np.sqrt(d)]), 0.9
    fit = _Fit(X, y, dtype)
    llh0, a0 = fit(params, s)
    stop, err = threading.Event(), []

    def neighbour():
        try:
            Xn, yn, _ = orc.synth_inputs(3000, d, 4, seed=5)
            other = _Fit(Xn, yn, dtype)
            l0, c0 = other(params, 1.1)
            while not stop.is_set():
                l1, c1 = other(params, 1.1)
                if l1 != l0 or not np.array_equal(c1, c0):
                    err.append("the neighbour's own fit changed: %r vs %r" % (l1, l0))
                    break
            other.close()
        except Exception as exc:       # noqa: BLE001
            err.append(repr(exc))

    t = threading.Thread(target=neighbour)
    t.start()
    try:
        for rep in range(reps):
            llh, a = fit(params, s)
            assert llh == llh0, "fit %d beside a neighbour: log_lh %r vs %r" % (rep, llh, llh0)
            assert np.array_equal(a, a0), "fit %d beside a neighbour: alpha differs in %d entries" % (rep, int((a != a0).sum()))
    finally:
        stop.set()
        t.join(120)
        fit.close()
    assert not err, err
    rows = [0, 1, n // 2, n - 2, n - 1, 777, 4097]
    from test_gpu_round4 import _residual
    assert _residual(X, y, a0, 1.0, 0.5 * np.sqrt(d), s, rows) < (1e-9 if dtype == "float64" else 2e-3)
