"""Round-4 GPU tests (``-m gpu``), all through the C ABI:

  * repeat-run determinism soak of the factorisation (the resident panel kernel's cross-XCD hand-offs): back-to-back
    fits must reproduce log_lh and alpha BIT FOR BIT, on the default route, without host pacing, with the formal
    release / acquire hand-off, and with a second handle's factorisation running on another host thread;
  * the reference's statistical property checks over its seeded parameter stream, restated for the GPU classes
    (gp/tests/test_gp.py:37-54 count_failures with the 95 % rule; :67-72 test_inv; :75-174 the finite-difference
    checks of dloglh / dlh / d2lh / dm);
  * the periodic kernel beyond toy size (periodic_c.pyx:18-30): a fit at N = 4096 (1-D) and K at d = 8;
  * the C multi-GPU schedule with EIGHT ranks (threads of this process sharing GPU 0, host-callback collectives):
    both panel-broadcast forms, ragged last block, N not a multiple of nb * P;
  * the plugin-kernel derivative glue (gp_c.pyx:34-131) on the device against its numpy restatement, with
    non-symmetric operands, and through gp.GP with a pure-Python kernel;
  * bench.py --gpus 4 over gloo: the multi-GPU line's schema.
"""
import ctypes
import json
import os
import threading

import numpy as np
import pytest

import gaussian_processes_amd as gp
from gaussian_processes_amd import _lib
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ soak --
class _Fit(object):
    """A gpx_gp handle driven directly (what bench.py times): set_params -> fit -> log_lh, alpha."""

    def __init__(self, X, y, dtype):
        self.lib = _lib.load()
        self.n, self.d = X.shape
        self.h = ctypes.c_void_p()
        dt = _lib.F64 if dtype == "float64" else _lib.F32
        _lib.check(self.lib.gpx_gp_create(ctypes.byref(self.h), dt, _lib.KERNEL_GAUSSIAN, self.n, self.d))
        xs, ys = np.ascontiguousarray(X, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
        _lib.check(self.lib.gpx_gp_set_data(self.h, _lib.dptr(xs), _lib.dptr(ys)))

    def __call__(self, params, s, want_alpha=True):
        p = np.ascontiguousarray(params, dtype=np.float64)
        _lib.check(self.lib.gpx_gp_set_params(self.h, _lib.dptr(p), float(s)))
        info = ctypes.c_int(0)
        _lib.check(self.lib.gpx_gp_fit(self.h, ctypes.byref(info)))
        assert info.value == 0
        out = ctypes.c_double(0.0)
        _lib.check(self.lib.gpx_gp_log_lh(self.h, ctypes.byref(out)))
        alpha = None
        if want_alpha:
            alpha = np.empty(self.n, dtype=np.float64)
            _lib.check(self.lib.gpx_gp_get_alpha(self.h, _lib.dptr(alpha)))
        return out.value, alpha

    def close(self):
        if self.h:
            self.lib.gpx_gp_destroy(self.h)
            self.h = ctypes.c_void_p()


def _residual(X, y, alpha, h, w, s, rows):
    """max |K[r, :] alpha - y[r]| / max|y| on a few rows, K from the kernel's definition (gaussian_c.pyx:27-35)."""
    c1, c2 = -0.5 / (w * w), 0.5 * np.sqrt(2.0 / np.pi) * h * h / w
    res = 0.0
    for r in rows:
        e = c1 * ((X - X[r]) ** 2).sum(1)
        k = np.where(e < _lib.MIN_LOG, 0.0, c2 * np.exp(e))
        k[r] += s * s
        res = max(res, abs(float(k @ alpha) - float(y[r])))
    return res / float(np.abs(y).max())


SOAK_REPS = int(os.environ.get("GPX_SOAK_REPS", "200"))
SOAK_REPS_NEIGHBOUR = int(os.environ.get("GPX_SOAK_REPS_NEIGHBOUR", "50"))


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("n", [257, 1990, 2048, 4171, 8192])
def test_soak_repeat_fits_are_bitwise_identical(monkeypatch, n, dtype):
    """200 back-to-back fits per route (default; GPX_POTRF_HOST_PACED=0; GPX_RES_STRICT=0: the relaxed hand-offs instead of
    the formal release / acquire pair) and 50 with another handle's factorisation running on a second host thread: every log_lh and alpha
    equal to the first fit's bit for bit, the routes equal to each other, and the first within tolerance of the
    oracle (n <= 4171) or of the sampled-row residual K alpha = y (n = 8192).  This is the test that would catch a
    rare ordering bug in the resident panel kernel's cross-XCD hand-offs (the event of round 3)."""
    d = 3
    X, y, _ = orc.synth_inputs(n, d, 4)
    h, w, s = 1.0, 0.5 * np.sqrt(d), 0.9
    params = np.array([h, w])
    f64 = dtype == "float64"
    first = {}
    for route, env in (("default", {}), ("not_host_paced", {"GPX_POTRF_HOST_PACED": "0"}), ("relaxed", {"GPX_RES_STRICT": "0"})):
        for k in ("GPX_POTRF_HOST_PACED", "GPX_RES_STRICT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        fit = _Fit(X, y, dtype)
        llh0, a0 = fit(params, s)
        first[route] = (llh0, a0)
        for rep in range(1, SOAK_REPS):
            llh, a = fit(params, s)
            assert llh == llh0, "n=%d %s %s: log_lh of fit %d differs: %r vs %r" % (n, dtype, route, rep, llh, llh0)
            assert np.array_equal(a, a0), "n=%d %s %s: alpha of fit %d differs in %d entries (max %.3e)" % (
                n, dtype, route, rep, int((a != a0).sum()), float(np.abs(a - a0).max()))
        fit.close()
    for k in ("GPX_POTRF_HOST_PACED", "GPX_RES_STRICT"):
        monkeypatch.delenv(k, raising=False)
    # the formal (default) and the relaxed hand-off run the same kernels in the same order: bitwise equal.  Without host pacing the launch order and the
    # instantiation a panel takes (whole CUs or shared ones) may differ; the arithmetic inside the leaves is the same in both
    # instantiations, the comparison stays at a tolerance because the update's tile shapes may differ with the route
    assert first["relaxed"][0] == first["default"][0]
    assert np.array_equal(first["relaxed"][1], first["default"][1])
    np.testing.assert_allclose(first["not_host_paced"][0], first["default"][0], rtol=1e-12 if f64 else 1e-6)
    np.testing.assert_allclose(first["not_host_paced"][1], first["default"][1], rtol=1e-9 if f64 else 1e-3, atol=1e-11 if f64 else 1e-4)
    # a neighbour: another handle (other size) factoring in a loop on a second host thread (its own look-ahead stream,
    # its own published blocks), while this thread repeats the fit
    stop = threading.Event()
    err = []

    def neighbour():
        try:
            Xn, yn, _ = orc.synth_inputs(3000, d, 4, seed=5)
            other = _Fit(Xn, yn, dtype)
            l0, a0n = other(params, 1.1)
            while not stop.is_set():
                l1, a1 = other(params, 1.1)
                if l1 != l0 or not np.array_equal(a1, a0n):
                    err.append("the neighbour's own fit changed: %r vs %r" % (l1, l0))
                    break
            other.close()
        except Exception as exc:       # noqa: BLE001
            err.append(repr(exc))

    t = threading.Thread(target=neighbour)
    t.start()
    try:
        fit = _Fit(X, y, dtype)
        for rep in range(SOAK_REPS_NEIGHBOUR):
            llh, a = fit(params, s)
            assert llh == first["default"][0], "n=%d %s beside a neighbour: log_lh of fit %d differs" % (n, dtype, rep)
            assert np.array_equal(a, first["default"][1]), "n=%d %s beside a neighbour: alpha of fit %d differs" % (n, dtype, rep)
        fit.close()
    finally:
        stop.set()
        t.join(120)
    assert not err, err
    # ... and the value that is reproduced is the right one
    llh0, a0 = first["default"]
    if n <= 4171:
        o = orc.OracleGP("gaussian", (h, w), X, y, s)
        np.testing.assert_allclose(llh0, o.log_lh, rtol=1e-10 if f64 else 1e-4)
        if f64:
            np.testing.assert_allclose(a0, o.inv_Kxx_y, rtol=1e-8, atol=1e-11)
        else:
            np.testing.assert_allclose(a0, o.inv_Kxx_y, rtol=2e-3, atol=2e-4)
    else:
        rows = [0, 1, n // 2, n - 2, n - 1, 777, 4097]
        assert _residual(X, y, a0, h, w, s, rows) < (1e-9 if f64 else 2e-3)


# ------------------------------------------------- the round-4 leaf (two waves, MFMAs in a fixed order) --
@pytest.mark.parametrize("N", [64, 130, 700, 1990, 4171])
def test_fp64_leaves_vs_oracle_and_each_other(monkeypatch, N):
    """The fp64 resident panel kernel has two leaves for the 64 x 64 diagonal block (gpx_leaf.h, GPX_LEAF):
      4 / 5        factor64_wave: wave 0 factors the TRANSPOSE held in accumulator tiles -- a strip's registers serve as MFMA
                   A and B operands as they are -- its MFMAs issued from asm in a fixed order between the stages of the pivot
                   arithmetic; wave 1 follows with the inverse from operands left in LDS.  4: one workgroup a CU, tiles in
                   AGPRs, a slot per step (single matrices up to 8192 rows); 5: two workgroups a CU, tiles in VGPRs, a ring
                   of four slots (everything else in fp64);
      1            factor64_mfma (round 3): four waves, three barriers and five LDS round trips a step.
    All three against the oracle's factor, log_lh, alpha and explicit inverse (W = inv(L_jj) of every leaf feeds it), and
    against each other (they differ in rounding only)."""
    d = 3
    X, y, Xo = orc.synth_inputs(N, d, 16)
    h, w, s = 1.0, 0.5 * np.sqrt(d), 0.9
    o = orc.OracleGP("gaussian", (h, w), X, y, s)
    out = {}
    routes = (("two_waves_lv4", {}), ("two_waves_lv5", {"GPX_LEAF4_ROWS": "0", "GPX_PANEL_EXCL_ROWS": "0"}), ("four_waves", {"GPX_LEAF": "1"}))
    for label, env in routes:
        for k in ("GPX_LEAF", "GPX_LEAF4_ROWS", "GPX_PANEL_EXCL_ROWS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = gp.GP(gp.GaussianKernel(h, w), X, y, s=s)
        out[label] = (float(g.log_lh), np.array(g.Lxx), np.array(g.inv_Kxx_y), np.array(g.inv_Kxx) if N <= 700 else None)
        np.testing.assert_allclose(out[label][0], o.log_lh, rtol=1e-10, err_msg=label)
        np.testing.assert_allclose(np.tril(out[label][1]), o.Lxx, rtol=1e-9, atol=1e-12, err_msg=label)
        np.testing.assert_allclose(out[label][2], o.inv_Kxx_y, rtol=1e-8, atol=1e-11, err_msg=label)
        if N <= 700:
            np.testing.assert_allclose(out[label][3], o.inv_Kxx, rtol=1e-7, atol=1e-10, err_msg=label)
    for k in ("GPX_LEAF", "GPX_LEAF4_ROWS", "GPX_PANEL_EXCL_ROWS"):
        monkeypatch.delenv(k, raising=False)
    np.testing.assert_allclose(np.tril(out["two_waves_lv4"][1]), np.tril(out["four_waves"][1]), rtol=1e-11, atol=1e-13)
    # the two instantiations of the two-wave leaf do the same arithmetic in the same order
    np.testing.assert_array_equal(np.tril(out["two_waves_lv4"][1]), np.tril(out["two_waves_lv5"][1]))
    # lock-step batches take the LV = 5 instantiation by default; GPX_LEAF = 1 / 4 force the others: same values
    from gaussian_processes_amd import mlii
    thetas = np.array([[h, w, s], [0.8, 1.1, 1.2]])
    b5 = mlii.log_lh_batch(X, y, thetas)
    monkeypatch.setenv("GPX_LEAF", "1")
    b1 = mlii.log_lh_batch(X, y, thetas)
    monkeypatch.setenv("GPX_LEAF", "4")
    b4 = mlii.log_lh_batch(X, y, thetas)
    monkeypatch.delenv("GPX_LEAF", raising=False)
    np.testing.assert_allclose(b5[0], o.log_lh, rtol=1e-10)
    np.testing.assert_allclose(b1, b5, rtol=1e-12)
    np.testing.assert_array_equal(b4, b5)


@pytest.mark.parametrize("N", [64, 700, 4171])
def test_fp32_panels_on_the_fp64_leaf_vs_oracle(monkeypatch, N):
    """fp32 panels hand their 64 x 64 diagonal blocks to the same fp64 two-wave leaf (it converts on the way in and out of LDS;
    GPX_LEAF = 1 keeps the round-3 fp32 MFMA leaf): all routes against the oracle at the fp32 tolerances, the two instantiations
    of the new leaf bit for bit, and the new leaf no further from the fp64 answer than the fp32 one."""
    d = 3
    X, y, Xo = orc.synth_inputs(N, d, 16)
    h, w, s = 1.0, 0.5 * np.sqrt(d), 0.9
    o = orc.OracleGP("gaussian", (h, w), X, y, s)
    out = {}
    routes = (("lv4", {}), ("lv5", {"GPX_LEAF4_ROWS": "0", "GPX_PANEL_EXCL_ROWS": "0"}), ("fp32_leaf", {"GPX_LEAF": "1"}))
    for label, env in routes:
        for k in ("GPX_LEAF", "GPX_LEAF4_ROWS", "GPX_PANEL_EXCL_ROWS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = gp.GP(gp.GaussianKernel(h, w), X, y, s=s, dtype="float32")
        out[label] = (float(g.log_lh), np.array(g.Lxx, dtype=np.float64), np.array(g.inv_Kxx_y, dtype=np.float64))
        np.testing.assert_allclose(out[label][0], o.log_lh, rtol=1e-4, err_msg=label)
        np.testing.assert_allclose(np.tril(out[label][1]), o.Lxx, rtol=2e-3, atol=2e-4, err_msg=label)
        np.testing.assert_allclose(out[label][2], o.inv_Kxx_y, rtol=2e-3, atol=2e-4, err_msg=label)
    for k in ("GPX_LEAF", "GPX_LEAF4_ROWS", "GPX_PANEL_EXCL_ROWS"):
        monkeypatch.delenv(k, raising=False)
    np.testing.assert_array_equal(np.tril(out["lv4"][1]), np.tril(out["lv5"][1]))
    err = {k: np.abs(np.tril(v[1]) - o.Lxx).max() for k, v in out.items()}
    assert err["lv4"] <= 1.5 * err["fp32_leaf"] + 1e-7, err


def test_async_fits_of_several_handles_on_one_thread_take_turns():
    """Handles fitted asynchronously back to back from ONE host thread run on their own streams but share that thread's
    scratch buffers (block inverses of the solves, solve operators, the panels' hand-off blocks): a call on another stream
    waits for the call before it (StreamTurn, csrc/gpx_common.h).  Four handles with different data and sizes, fits enqueued
    without a synchronisation in between, every log_lh and alpha equal to the value of the same handle fitted alone."""
    lib = _lib.load()
    sizes = [256, 1000, 1990, 640]
    hs, alone = [], []
    try:
        for k, N in enumerate(sizes):
            d = 2 + k
            X, y, _ = orc.synth_inputs(N, d, 4, seed=30 + k)
            prm = np.array([1.0 + 0.1 * k, 0.6 * np.sqrt(d)])
            h = ctypes.c_void_p()
            _lib.check(lib.gpx_gp_create(ctypes.byref(h), _lib.F64, _lib.KERNEL_GAUSSIAN, N, d))
            _lib.check(lib.gpx_gp_set_data(h, _lib.dptr(np.ascontiguousarray(X)), _lib.dptr(np.ascontiguousarray(y))))
            _lib.check(lib.gpx_gp_set_params(h, _lib.dptr(prm), 0.9))
            hs.append(h)
            _lib.check(lib.gpx_gp_fit(h, None))
            v = ctypes.c_double(0.0)
            _lib.check(lib.gpx_gp_log_lh(h, ctypes.byref(v)))                     # (synchronises: this handle alone)
            np.testing.assert_allclose(v.value, orc.OracleGP("gaussian", prm, X, y, 0.9).log_lh, rtol=1e-10)
            alone.append(v.value)
        for rep in range(40):
            for h in hs:
                _lib.check(lib.gpx_gp_fit(h, None))                               # enqueue only: four streams
            for h, r in zip(hs, alone):
                v = ctypes.c_double(0.0)
                _lib.check(lib.gpx_gp_log_lh(h, ctypes.byref(v)))
                assert v.value == r, (rep, v.value, r)
    finally:
        for h in hs:
            lib.gpx_gp_destroy(h)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("N,nb", [(4700, "1024"), (3072, "1024"), (2400, "768"), (5120 + 37, "1024")])
def test_wide_panel_recursive_halving_vs_oracle(monkeypatch, dtype, N, nb):
    """A 768 / 1024-wide outer block at small n (GPX_POTRF_NB): the panel halves recursively down to 256-column resident
    launches with one MFMA product between the halves.  Against the oracle: ragged last block, three and four sub-panels,
    the rider row, a lock-step batch.  (Round 4's alternative -- the wide panel as a nested factorisation with its own
    look-ahead -- measured no faster anywhere and was removed in round 6.)"""
    from gaussian_processes_amd import mlii
    d = 3
    X, y, Xo = orc.synth_inputs(N, d, 16)
    h, w, s = 1.0, 0.5 * np.sqrt(d), 0.9
    o = orc.OracleGP("gaussian", (h, w), X, y, s)
    monkeypatch.setenv("GPX_POTRF_NB", nb)
    thetas = np.array([[h, w, s], [0.8, 1.1, 1.2]])
    g = gp.GP(gp.GaussianKernel(h, w), X, y, s=s, dtype=dtype)
    llh = float(g.log_lh)
    L, alpha = np.array(g.Lxx, dtype=np.float64), np.array(g.inv_Kxx_y, dtype=np.float64)
    batch = mlii.log_lh_batch(X, y, thetas, dtype=dtype) if N <= 4700 else None
    f64 = dtype == "float64"
    np.testing.assert_allclose(llh, o.log_lh, rtol=1e-10 if f64 else 1e-4)
    np.testing.assert_allclose(np.tril(L), o.Lxx, **(dict(rtol=1e-9, atol=1e-12) if f64 else dict(rtol=2e-3, atol=2e-4)))
    np.testing.assert_allclose(alpha, o.inv_Kxx_y, **(dict(rtol=1e-8, atol=1e-11) if f64 else dict(rtol=2e-3, atol=2e-4)))
    if batch is not None:
        np.testing.assert_allclose(batch[0], o.log_lh, rtol=1e-10 if f64 else 1e-4)


# ------------------------------------- the reference's property checks, seeded stream --
DTHETA = 1e-5            # gp/tests/util.py:9 'dtheta'
PFAIL = 5                # 'pct_allowed_failures'


def _allclose(x, y):     # gp/tests/util.py:51-52
    return np.allclose(x, y, rtol=1e-5)


def _make_xy():          # gp/tests/util.py:35-38
    x = np.linspace(-2 * np.pi, 2 * np.pi, 16)
    return x, np.sin(x)


def _make_xo():          # gp/tests/util.py:41-43
    return np.linspace(-2 * np.pi, 2 * np.pi, 32)


def _make_random_gp():   # gp/tests/test_gp.py:29-34 with util.rand_params('h', 'w', 's')
    x, y = _make_xy()
    h = np.random.uniform(0, 2)
    w = np.random.uniform(np.pi / 32., np.pi / 2.)
    s = np.random.uniform(0, 0.5)
    return gp.GP(gp.GaussianKernel(h, w), x, y, s=s)


def _count_failures(check, n):   # gp/tests/test_gp.py:37-54
    np.random.seed(2348)
    failures = []
    for _ in range(n):
        g = _make_random_gp()
        try:
            check(g)
        except AssertionError as err:
            failures.append((tuple(g.params), str(err)[:200]))
    pfail = 100.0 * len(failures) / n
    assert pfail < PFAIL, "%s failed %d/%d (%.1f%%): %s" % (check.__name__, len(failures), n, pfail, failures[:3])


def _central(g, prop, i):
    """(prop(theta + dtheta e_i) - prop(theta - dtheta e_i)) / 2 / dtheta  (util.approx_deriv with gp.copy())"""
    vals = []
    for sign in (-1.0, 1.0):
        p = np.array(g.params)
        p[i] += sign * DTHETA
        c = g.copy()
        c.params = p
        vals.append(prop(c))
    return (vals[1] - vals[0]) / 2.0 / DTHETA


def test_reference_property_inv():
    def check_inv(g):            # test_gp.py:67-72
        I = np.dot(g.Kxx, g.inv_Kxx)
        assert _allclose(I, np.eye(I.shape[0]))
    _count_failures(check_inv, 10)


def test_reference_property_dloglh():
    def check_dloglh(g):         # test_gp.py:75-97
        jac = g.dloglh_dtheta
        approx = np.array([_central(g, lambda c: c.log_lh, i) for i in range(len(g.params))])
        assert _allclose(jac, approx)
    _count_failures(check_dloglh, 100)


def test_reference_property_dlh():
    def check_dlh(g):            # test_gp.py:100-122
        jac = g.dlh_dtheta
        approx = np.array([_central(g, lambda c: c.lh, i) for i in range(len(g.params))])
        assert _allclose(jac, approx)
    _count_failures(check_dlh, 100)


def test_reference_property_d2lh():
    def check_d2lh(g):           # test_gp.py:125-147
        hess = g.d2lh_dtheta2
        approx = np.empty(hess.shape)
        for i in range(len(g.params)):
            approx[:, i] = _central(g, lambda c: c.dlh_dtheta, i)
        assert _allclose(hess, approx)
    _count_failures(check_d2lh, 100)


def test_reference_property_dm():
    xo = _make_xo()

    def check_dm(g):             # test_gp.py:150-174
        jac = g.dm_dtheta(xo)
        approx = np.array([_central(g, lambda c: c.mean(xo), i) for i in range(len(g.params))])
        assert _allclose(jac, approx)
    _count_failures(check_dm, 100)


# ------------------------------------------------------------- periodic at size --
def test_periodic_fit_n4096_1d_vs_oracle():
    """periodic_c.pyx:18-30 through the whole hot path at a size where the factorisation takes every route
    (resident panels, tapered outer block, riding right-hand side): N = 4096, 1-D, against the oracle."""
    N, m = 4096, 200
    X, y, Xo = orc.synth_inputs(N, 1, m)
    prm, s = (1.1, 0.8, 2.3), 0.5
    g = gp.GP(gp.PeriodicKernel(*prm), X.ravel(), y, s=s)
    o = orc.OracleGP("periodic", prm, X.ravel(), y, s)
    np.testing.assert_allclose(g.log_lh, o.log_lh, rtol=1e-10)
    np.testing.assert_allclose(g.inv_Kxx_y, o.inv_Kxx_y, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(g.mean(Xo.ravel()), o.mean(Xo.ravel()), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(np.diag(g.Lxx), np.diag(o.Lxx), rtol=1e-9)
    # fp32 route at the SURVEY 8(d) tolerances
    g32 = gp.GP(gp.PeriodicKernel(*prm), X.ravel(), y, s=s, dtype="float32")
    np.testing.assert_allclose(g32.log_lh, o.log_lh, rtol=1e-4)
    np.testing.assert_allclose(g32.mean(Xo.ravel()), o.mean(Xo.ravel()), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("n,m,d", [(3000, 2500, 8), (4096, 4096, 8), (1025, 3001, 3)])
def test_periodic_kernel_matrix_at_size_vs_oracle(n, m, d):
    """K only at d > 1 (the reference is 1-D; the oracle restates periodic_c.pyx:27-29 with the distance taken over
    the d inputs the way the build does): interior-tile fast path and ragged edges."""
    rng = np.random.RandomState(n + m + d)
    a = rng.uniform(-3, 3, (n, d))
    b = rng.uniform(-3, 3, (m, d))
    p = gp.PeriodicKernel(1.1, 0.8, 2.3)
    np.testing.assert_allclose(p(a, b), orc.kernel_matrix("periodic", "K", a, b, p.params), rtol=1e-11, atol=1e-300)


# ------------------------------------------ the C schedule with eight ranks (threads) --
@pytest.mark.parametrize("N,nb,sag,dtype_id", [(8492, 512, False, 0), (8492, 512, True, 0), (12288 + 77, 512, True, 0),
                                               (9000, 256, True, 1)])
def test_native_mg_world8_threads_on_one_gpu_vs_oracle(N, nb, sag, dtype_id):
    """gpx_mg_* with world = 8 -- the north star's rank count -- as eight threads of this process sharing GPU 0 (a GPU
    box admits at most six processes on its card), collectives through host callbacks that rendezvous on a
    threading.Barrier.  N is not a multiple of nb * 8 and the last block is ragged; `sag`: panels travel as scatter +
    all-gather (eight pieces, the last one longer), the route counters say which form ran.  Against the oracle."""
    from _thread_world import run_thread_world
    d, m = 3, 40
    res = run_thread_world(8, N, d, nb, m, dtype_id=dtype_id, sag=sag)
    X, y, Xo = orc.synth_inputs(N, d, m)
    o = orc.OracleGP("gaussian", (1.0, 0.5 * np.sqrt(d)), X, y, 1.0)
    assert res["info"] == 0
    if dtype_id == 0:
        np.testing.assert_allclose(res["log_lh"], o.log_lh, rtol=1e-10)
        np.testing.assert_allclose(res["alpha"], o.inv_Kxx_y, rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(res["mean"], o.mean(Xo), rtol=1e-8, atol=1e-11)
    else:
        np.testing.assert_allclose(res["log_lh"], o.log_lh, rtol=1e-4)
        np.testing.assert_allclose(res["mean"], o.mean(Xo), rtol=1e-3, atol=1e-3)
    assert res["log_lh2"] == res["log_lh"]                 # a second fit reuses buffers, events and streams
    # every rank reports the same scalars
    assert len(set(res["log_lh_per_rank"])) == 1, res["log_lh_per_rank"]
    if sag:
        assert res["sag_routes"] > 0
    else:
        assert res["sag_routes"] == 0 and res["one_routes"] > 0


# ------------------------------------------------------ plugin-kernel derivative glue --
def _glue_numpy(y, Ki, Kj, Kh, Kiy, s, lh):
    """gp_c.pyx:34-111 in numpy, term by term (test infrastructure)."""
    n, m = Kj.shape[0], Kj.shape[1]
    P = n + 1
    dK = [Kj[i] if i < n else np.eye(m) * 2 * s for i in range(P)]
    dloglh, dlh = np.empty(P), np.empty(P)
    for i in range(P):
        k = Ki @ dK[i]
        dloglh[i] = 0.5 * y @ (k @ Kiy) - 0.5 * np.trace(k)
        dlh[i] = 0.5 * lh * (y @ (k @ Kiy) - np.trace(k))
    dKi = [-Ki @ (dK[i] @ Ki) for i in range(P)]
    d2 = np.empty((P, P))
    for i in range(P):
        KidK = Ki @ dK[i]
        tr_i = y @ (KidK @ Kiy) - np.trace(KidK)
        for j in range(P):
            d2k = Kh[i, j] if (i < n and j < n) else (np.eye(m) * 2 if (i == n and j == n) else np.zeros((m, m)))
            q = dKi[j] @ dK[i]
            t1 = lh * (y @ (q @ Kiy) + Kiy @ (d2k @ Kiy) + Kiy @ (dK[i] @ (dKi[j] @ y)) - np.trace(q + Ki @ d2k))
            d2[i, j] = 0.5 * (dlh[j] * tr_i + t1)
    return dloglh, dlh, d2


@pytest.mark.parametrize("n,npar", [(1, 1), (37, 2), (300, 3), (777, 2)])
def test_gp_c_glue_on_device_vs_numpy_with_nonsymmetric_operands(n, npar):
    """ext.gp_c.{dloglh_dtheta, dlh_dtheta, d2lh_dtheta2, dm_dtheta} (gpx_gp_c_*: matrices uploaded once, matrix-vector
    work + trace reductions + one GEMM per parameter) against the reference's own dense-product formulas in numpy, on
    RANDOM operands -- nothing symmetric, Kiy not equal to Ki y -- so that every transposition in the device
    formulation is exercised."""
    from gaussian_processes_amd.ext import gp_c
    rng = np.random.RandomState(n * 7 + npar)
    sc = 1.0 / np.sqrt(n)
    y = rng.randn(n); Kiy = rng.randn(n)
    Ki = rng.randn(n, n) * sc
    Kj = rng.randn(npar, n, n) * sc
    Kh = rng.randn(npar, npar, n, n) * sc
    s, lh = 0.7, 0.37
    ref_dloglh, ref_dlh, ref_d2 = _glue_numpy(y, Ki, Kj, Kh, Kiy, s, lh)
    out = np.empty(npar + 1)
    gp_c.dloglh_dtheta(y, Ki, Kj, Kiy, s, out)
    np.testing.assert_allclose(out, ref_dloglh, rtol=1e-11, atol=1e-12)
    out2 = np.empty(npar + 1)
    gp_c.dlh_dtheta(y, Ki, Kj, Kiy, s, lh, out2)
    np.testing.assert_allclose(out2, ref_dlh, rtol=1e-11, atol=1e-12)
    d2 = np.empty((npar + 1, npar + 1))
    gp_c.d2lh_dtheta2(y, Ki, Kj, Kh, Kiy, s, lh, ref_dlh, d2)
    np.testing.assert_allclose(d2, ref_d2, rtol=1e-10, atol=1e-11)
    m = 23
    Kjxo = rng.randn(npar, m, n) * sc
    Kxox = rng.randn(m, n) * sc
    dm = np.empty((npar + 1, m))
    gp_c.dm_dtheta(y, Ki, Kj, Kjxo, Kxox, s, dm)
    ref_dm = np.empty((npar + 1, m))
    for i in range(npar + 1):                                  # gp_c.pyx:121-131
        dKxox = Kjxo[i] if i < npar else np.zeros((m, n))
        dKxx = Kj[i] if i < npar else np.eye(n) * 2 * s
        ref_dm[i] = dKxox @ (Ki @ y) - Kxox @ ((Ki @ (dKxx @ Ki)) @ y)
    np.testing.assert_allclose(dm, ref_dm, rtol=1e-11, atol=1e-12)
    # the reference's buffer errors
    with pytest.raises(ValueError):
        gp_c.dloglh_dtheta(y.astype(np.float32), Ki, Kj, Kiy, s, out)
    with pytest.raises(ValueError):
        gp_c.dloglh_dtheta(y, np.asfortranarray(Ki) if n > 1 else Ki[:, :0], Kj, Kiy, s, out)


class _RBFWithDerivs(gp.kernels.Kernel):
    """A pure-Python plugin kernel with jacobian / hessian (no native id): K = h^2 exp(-r^2 / (2 l^2))."""

    def __init__(self, h, ell):
        self.h, self.ell = float(h), float(ell)

    @property
    def params(self):
        return np.array([self.h, self.ell])

    @params.setter
    def params(self, val):
        self.h, self.ell = float(val[0]), float(val[1])

    def _r2(self, x1, x2):
        a = np.asarray(x1, dtype=np.float64).reshape(len(x1), -1)
        b = np.asarray(x2, dtype=np.float64).reshape(len(x2), -1)
        return ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)

    def K(self, x1, x2, out=None):
        return self.h ** 2 * np.exp(-0.5 * self._r2(x1, x2) / self.ell ** 2)

    def jacobian(self, x1, x2, out=None):
        r2 = self._r2(x1, x2)
        K = self.K(x1, x2)
        return np.stack([2.0 * K / self.h, K * r2 / self.ell ** 3])

    def hessian(self, x1, x2, out=None):
        r2 = self._r2(x1, x2)
        K = self.K(x1, x2)
        l = self.ell
        H = np.empty((2, 2) + K.shape)
        H[0, 0] = 2.0 * K / self.h ** 2
        H[0, 1] = H[1, 0] = 2.0 * K * r2 / (self.h * l ** 3)
        H[1, 1] = K * (r2 * r2 / l ** 6 - 3.0 * r2 / l ** 4)
        return H


def test_plugin_kernel_derivative_properties_through_gp():
    """gp.GP with a pure-Python kernel: dloglh / dlh / d2lh / dm go host K, jacobian, hessian -> ext.gp_c (device)
    and must equal the reference's formulas evaluated in numpy on the same matrices, and the central differences of
    log_lh (the reference's own check, test_gp.py:75-97)."""
    rng = np.random.RandomState(4)
    N, m = 120, 17
    X = rng.uniform(-3, 3, N); y = np.sin(X) + 0.1 * rng.randn(N)
    Xo = rng.uniform(-3, 3, m)
    k = _RBFWithDerivs(1.3, 0.9)
    s = 0.6
    g = gp.GP(k, X, y, s=s)
    assert getattr(k, "_native_kernel", None) is None
    Kxx = k(X, X) + s * s * np.eye(N)
    Ki = np.linalg.inv(Kxx)
    Kiy = Ki @ y
    llh = -0.5 * y @ Kiy - 0.5 * np.linalg.slogdet(Kxx)[1] - 0.5 * N * np.log(2 * np.pi)
    lh = np.exp(llh)
    ref_dloglh, ref_dlh, ref_d2 = _glue_numpy(y, Ki, k.jacobian(X, X), k.hessian(X, X), Kiy, s, lh)
    np.testing.assert_allclose(g.dloglh_dtheta, ref_dloglh, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(g.dlh_dtheta, ref_dlh, rtol=1e-7, atol=1e-300)
    np.testing.assert_allclose(g.d2lh_dtheta2, ref_d2, rtol=1e-6, atol=1e-300)
    approx = np.array([_central(g, lambda c: c.log_lh, i) for i in range(3)])
    np.testing.assert_allclose(g.dloglh_dtheta, approx, rtol=1e-5, atol=1e-6)
    Kjxo = k.jacobian(Xo, X)
    Kxox = k(Xo, X)
    ref_dm = np.empty((3, m))
    for i in range(3):
        dKxox = Kjxo[i] if i < 2 else np.zeros((m, N))
        dKxx = k.jacobian(X, X)[i] if i < 2 else np.eye(N) * 2 * s
        ref_dm[i] = dKxox @ Kiy - Kxox @ (Ki @ (dKxx @ Kiy))
    np.testing.assert_allclose(g.dm_dtheta(Xo), ref_dm, rtol=1e-7, atol=1e-9)


# ------------------------------------------------------------- bench: multi-GPU line --
def test_bench_four_ranks_line_schema_over_gloo():
    """`python bench.py --gpus 4` from a plain invocation, four ranks sharing GPU 0 with the host-callback data plane
    (a GPU box admits six processes on its card, so the 8-rank line cannot be rehearsed here): the fields the driver
    and the reviewer read from the multi-GPU line -- per-rank communicator info, the broadcast autotune (world > 2),
    per-rank stage and chain times, the watchdog -- are present and consistent, and the result is the oracle's."""
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(GPX_DIST_BACKEND="gloo", GPX_BENCH_SINGLE_DEVICE="1", GPX_BENCH_TUNE_BUDGET_S="5")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--problem-n", "6144",
                        "--problem-d", "4", "--problem-m", "64", "--steps", "1", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["scaling"] == "strong" and out["unit"] == "s"
    assert len(out["comm_info_per_rank"]) == 4 and [c["rank"] for c in out["comm_info_per_rank"]] == [0, 1, 2, 3]
    assert out["rccl_nranks"] == 0                           # the callback data plane says so: no silent claim of RCCL
    assert len(out["stage_and_chain_ms_per_rank"]) == 4
    for row in out["stage_and_chain_ms_per_rank"]:
        assert set(row) >= {"kernel_build", "factor", "solve", "chain_panel", "chain_bcast", "chain_update"}
        assert row["factor"] > 0
    # (round 5) the schedule's free parameters are measured in the run: nb x chunks x broadcast form, under a budget
    tune = out["schedule_autotune"]
    assert tune is not None and tune["measured"] >= 1 and tune["measured"] == len(tune["table"]) <= tune["candidates"]
    assert {r["nb"] for r in tune["table"]} <= {256, 512, 1024} and all(r["fit_s"] > 0 for r in tune["table"])
    best = min(tune["table"], key=lambda r: (r["fit_s"], r["nb"], r["chunks"], r["sag"]))
    assert tune["chosen"] == {k: best[k] for k in ("nb", "chunks", "sag")}
    assert out["panel_bcast"].startswith("scatter") == bool(tune["chosen"]["sag"]) and out["bcast_chunks"] == tune["chosen"]["chunks"]
    # (round 6) what the handle says is IN EFFECT, read back after the tuning pass
    eff = out["schedule_in_effect"]
    assert eff["chunks"] == tune["chosen"]["chunks"] and eff["owner_first"] is True and eff["wait_timing"] is True
    assert eff["panel_bcast"].startswith("scatter") == bool(tune["chosen"]["sag"])
    assert ("nb=%d" % tune["chosen"]["nb"]) in out["config"]["parallelism"]
    for row in out["stage_and_chain_ms_per_rank"]:
        assert row["exposed_wait"] >= 0 and row["exposed_wait_max"] >= 0 and row["exposed_waits_over_20us"] >= 0
    assert out["watchdog_s"] > 0 and out["check"]["max_abs_residual_K_alpha_minus_y_over_max_y"] < 1e-9
    X, y, _ = orc.synth_inputs(6144, 4, 64)
    o = orc.OracleGP("gaussian", (1.0, 0.5 * np.sqrt(4)), X, y, 1.0)
    np.testing.assert_allclose(out["log_lh"], o.log_lh, rtol=1e-10)
    np.testing.assert_allclose(out["first_fit_log_lh"], o.log_lh, rtol=1e-10)


def test_roctx_ranges_are_emitted_when_asked_for_and_change_nothing(monkeypatch):
    """GPX_ROCTX=1: every gpx_gp_* call and every launch class below it pushes a roctx range (SURVEY section 5's tracing
    suggestion; `rocprofv3 --marker-trace` shows them, profiles/r04_roctx_marker_trace.txt).  Off by default: no range is pushed;
    on: ranges are counted and the results are bit for bit the same."""
    lib = _lib.load()
    def count():
        v = ctypes.c_int64(0)
        _lib.check(lib.gpx_debug_roctx_ranges(ctypes.byref(v)))
        return v.value
    N, d = 700, 3
    X, y, Xo = orc.synth_inputs(N, d, 16)
    monkeypatch.delenv("GPX_ROCTX", raising=False)
    c0 = count()
    g = gp.GP(gp.GaussianKernel(1.0, 0.9), X, y, s=0.9)
    ref = (float(g.log_lh), np.array(g.mean(Xo)))
    assert count() == c0
    monkeypatch.setenv("GPX_ROCTX", "1")
    g2 = gp.GP(gp.GaussianKernel(1.0, 0.9), X, y, s=0.9)
    got = (float(g2.log_lh), np.array(g2.mean(Xo)))
    monkeypatch.delenv("GPX_ROCTX", raising=False)
    assert count() >= c0 + 4                                   # the fit call, its kernel build, panels, solves ...
    assert got[0] == ref[0]
    np.testing.assert_array_equal(got[1], ref[1])
