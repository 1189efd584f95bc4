"""CPU tests (no GPU needed): host-side logic of the drop-in classes, and that
the C-ABI library loads and exports every symbol include/gpx.h declares.
Mirrors gp/tests/test_gp.py:245-432 and gp/tests/test_kernels.py:13-26 where no
device compute is involved."""
import os
import pickle
import re
from copy import copy, deepcopy

import numpy as np
import pytest

import gaussian_processes_amd as gp
from gaussian_processes_amd import _lib
from conftest import ROOT


def make_xy():
    x = np.linspace(-2 * np.pi, 2 * np.pi, 16)
    return x, np.sin(x)


def make_gp():
    x, y = make_xy()
    return gp.GP(gp.GaussianKernel(1, 1), x, y, s=1)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gpx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gpx_[a-z0-9_A-Z]+)\s*\(", hdr))
    declared.discard("gpx_gp")
    assert len(declared) > 50
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), "libgpx.so does not export %s" % name
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert lib.gpx_version() == 100


def test_library_never_calls_getenv():
    """Every GPX_* switch is read through ONE snapshot per API call (csrc/gpx_tune.h: a pass over `environ` at the entry
    point), never by a getenv at its point of use -- round 4 had 66 such sites, a dozen of them per panel launch.  The
    library does not even import getenv; every name in the table is documented in DESIGN.md, and no source file reads a
    switch that is not in the table."""
    import subprocess
    und = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und, [l for l in und.splitlines() if "getenv" in l]
    csrc = os.path.join(ROOT, "gaussian_processes_amd", "csrc")
    table = open(os.path.join(csrc, "gpx_tune.h")).read()
    names = set(re.findall(r'"(GPX_[A-Z0-9_]+)"', table)) | {"GPX_POTRF_WIDTHS", "GPX_MG_BCAST", "GPX_RCCL_LIB"}
    assert 30 <= len(names) <= 45, len(names)       # round 6: pruned from 61 (the routes two rounds of measurement left neutral or slower)
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    missing = [n for n in sorted(names) if ("`%s`" % n) not in design and ("`%s=" % n) not in design]
    assert not missing, missing
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")) and f != "gpx_tune.h":
            src = open(os.path.join(csrc, f)).read()
            assert not re.search(r"\bgetenv\s*\(|env_i64\s*\(|env_set\s*\(", src), f
            for n in re.findall(r'"(GPX_[A-Z0-9_]+)"', src):
                assert n in names, (f, n)
    # a snapshot without a GPU: the counter moves with every entry point (here one that fails for lack of arguments)
    lib = _lib.load()
    import ctypes
    a, b = ctypes.c_int64(), ctypes.c_int64()
    assert lib.gpx_debug_tune_refreshes(ctypes.byref(a)) == 0
    lib.gpx_d_potrf(0, None, -1, 0, None, None)
    assert lib.gpx_debug_tune_refreshes(ctypes.byref(b)) == 0
    assert b.value == a.value + 1


def test_no_cpu_fallback_without_gpu():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    k = gp.GaussianKernel(1, 1)
    with pytest.raises(_lib.GpxError):
        k(np.zeros(3), np.zeros(3))
    with pytest.raises(_lib.GpxError):
        make_gp().log_lh


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "gaussian_processes_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("the oracle's", ""), os.path.join(dirpath, f)


@pytest.mark.parametrize("cls,good", [(gp.GaussianKernel, (0.7, 1.3)), (gp.PeriodicKernel, (0.7, 1.3, 2.0))])
def test_kernel_params_and_validation(cls, good):
    k = cls(*good)
    assert (k.params == np.array(good)).all()
    assert k.params.dtype == np.float64
    k.params = good
    assert (k.params == np.array(good)).all()
    for i in range(len(good)):
        bad = list(good)
        bad[i] = 0
        with pytest.raises(ValueError):
            cls(*bad)
        with pytest.raises(ValueError):
            k.params = bad
    with pytest.raises(ValueError):
        k.set_param("nope", 1.0)
    assert type(k.h) is np.float64


def test_kernel_copy_pickle():
    k1 = gp.GaussianKernel(0.3, 0.9)
    for k2 in (k1.copy(), copy(k1), deepcopy(k1), pickle.loads(pickle.dumps(k1))):
        assert k1.h == k2.h and k1.w == k2.w and k1 is not k2
    k3 = deepcopy(k1)
    assert k1.h is not k3.h
    p1 = gp.PeriodicKernel(0.3, 0.9, 1.7)
    p2 = pickle.loads(pickle.dumps(p1))
    assert (p1.params == p2.params).all()


def test_kernel_buffer_errors():
    from gaussian_processes_amd.ext import gaussian_c
    x = np.linspace(0, 1, 4)
    out = np.empty((4, 4))
    with pytest.raises(ValueError, match="dtype mismatch"):
        gaussian_c.K(out, x.astype(np.float32), x, 1.0, 1.0)
    with pytest.raises(ValueError, match="not C-contiguous"):
        gaussian_c.K(np.empty((4, 8))[:, ::2], x, x, 1.0, 1.0)
    with pytest.raises(ValueError, match="wrong number of dimensions"):
        gaussian_c.K(out, x[:, None], x, 1.0, 1.0)
    with pytest.raises(ValueError, match="shape"):
        gaussian_c.K(np.empty((4, 5)), x, x, 1.0, 1.0)


def test_gp_inputs_are_readonly_float64_copies():
    x, y = make_xy()
    g = gp.GP(gp.GaussianKernel(1, 1), x, y, s=1)
    assert g.x.dtype == np.float64 and g.y.dtype == np.float64 and type(g.s) is np.float64
    assert g.x is not x and not g.x.flags.writeable and not g.y.flags.writeable
    assert g.params.dtype == np.float64 and (g.params == [1, 1, 1]).all()


def test_set_y_shape_and_negative_s():
    g = make_gp()
    with pytest.raises(ValueError):
        g.y = g.y.copy()[:, None]
    x, y = make_xy()
    with pytest.raises(ValueError):
        gp.GP(gp.GaussianKernel(1, 1), x, y, s=-1)


def test_reset_memoized_on_any_setter():
    g = make_gp()
    for prop, val in (("x", g.x.copy() + 1), ("y", g.y.copy() + 1), ("s", g.s + 1),
                      ("params", g.params + 1)):
        g._memoized["sentinel"] = 1
        setattr(g, prop, val)
        assert g._memoized == {}


def test_memoprop_del():
    g = make_gp()
    g._memoized["Kxx"] = "cached"
    assert g.Kxx == "cached"
    del g.Kxx
    assert "Kxx" not in g._memoized


def test_set_params_and_unknown_name():
    g = make_gp()
    g.set_param("h", 1)
    g.set_param("w", 0.2)
    g.set_param("s", 0.01)
    assert g.get_param("w") == 0.2 and g.get_param("s") == 0.01
    with pytest.raises(AttributeError):
        g.set_param("p", 1.1)


def test_copy_semantics():
    g1 = make_gp()
    for g2 in (g1.copy(deep=False), copy(g1)):
        assert g1 is not g2 and g1._x is g2._x and g1._y is g2._y and g1._s is g2._s and g1.K is g2.K
    for g2 in (g1.copy(deep=True), deepcopy(g1), pickle.loads(pickle.dumps(g1))):
        assert g1 is not g2 and g1._x is not g2._x and g1._y is not g2._y and g1.K is not g2.K
        assert g1._s is not g2._s
        assert (g1._x == g2._x).all() and (g1._y == g2._y).all() and g1._s == g2._s
        assert (g1.K.params == g2.K.params).all()


def test_pickle_carries_memoized_host_arrays():
    g1 = make_gp()
    g1._memoized["Kxx"] = np.eye(16)
    g2 = pickle.loads(pickle.dumps(g1))
    assert (g2._memoized["Kxx"] == np.eye(16)).all()


def test_nd_inputs_accepted():
    X = np.random.RandomState(0).randn(10, 3)
    g = gp.GP(gp.GaussianKernel(1, 1), X, np.zeros(10), s=1)
    assert g._n == 10 and g._d == 3
    with pytest.raises(ValueError):
        gp.GP(gp.GaussianKernel(1, 1), X, np.zeros((10, 3)), s=1)


# ---- bench.py as the driver calls it: `python bench.py --gpus P` must start P ranks itself ----
def _run_bench(*argv):
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)


def test_bench_self_launches_ranks_from_a_plain_invocation():
    import json
    r = _run_bench("--gpus", "2", "--launch-check")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                      # exactly ONE JSON line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["launch_check"] is True and out["rank_sum"] == 3.0


def test_bench_self_launch_propagates_a_rank_failure():
    # without a GPU the ranks refuse to run (no CPU fallback): the launcher must exit non-zero
    # and print no result line
    from gaussian_processes_amd import _lib
    if _lib.device_count() >= 1:
        pytest.skip("a GPU is present: the ranks would run")
    r = _run_bench("--gpus", "2", "--problem-n", "512", "--no-cpu-baseline")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_pickled_state_has_exactly_the_reference_keys():
    # gp/gp.py:78-92: K, _x, _y, _s, _memoized -- nothing else for a GP built the reference's way
    g = make_gp()
    assert sorted(g.__getstate__()) == ["K", "_memoized", "_s", "_x", "_y"]
    g2 = pickle.loads(pickle.dumps(g))
    assert g2._dtype == g._dtype and g2._device is None
    # the opt-in extensions (fp32 device path, explicit device) survive a round trip
    x, y = make_xy()
    g32 = gp.GP(gp.GaussianKernel(1, 1), x, y, s=1, dtype="float32", device=0)
    g3 = pickle.loads(pickle.dumps(g32))
    assert g3._dtype == _lib.F32 and g3._device == 0
