#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

TEST INFRASTRUCTURE -- runs only in the authoring container, never on the GPU
box and never from the product path.

What it does
------------
1. Copies /root/reference/{gp,setup.py,VERSION.txt,README.md} into a scratch
   directory under /tmp and runs the reference's own
   ``setup.py build_ext --inplace`` there (Cython -> C, unmodified sources).
   Nothing from the reference (source, C, bytecode, .so) is written into this
   repository.
2. Imports the built ``gp`` package.  The reference is Python-2 code whose
   package __init__ files use implicit relative imports
   (gp/ext/__init__.py:1-3, gp/kernels/__init__.py:1-3); a meta-path alias
   finder maps the bare names onto the dotted modules so that no reference
   source has to be edited.
3. Evaluates the hot path (and the derivative stack) on the inputs used by the
   reference's own tests (gp/tests/util.py:15-48, gp/tests/test_gp.py:299-318)
   plus a few larger seeded cases, and stores inputs + outputs as .npz files.

Usage:  python3 oracle/make_golden.py [--scratch /tmp/gpref_build]
"""
import argparse
import importlib
import importlib.abc
import importlib.util
import json
import os
import shutil
import subprocess
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

ALIASES = {
    "gaussian_c": "gp.ext.gaussian_c",
    "periodic_c": "gp.ext.periodic_c",
    "gp_c": "gp.ext.gp_c",
    "base": "gp.kernels.base",
    "periodic": "gp.kernels.periodic",
    "gaussian": "gp.kernels.gaussian",
}


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real):
        self.real = real

    def create_module(self, spec):
        return importlib.import_module(self.real)

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path=None, target=None):
        if name in ALIASES:
            return importlib.util.spec_from_loader(name, _AliasLoader(ALIASES[name]))
        return None


def build_reference(scratch):
    if os.path.isdir(scratch):
        shutil.rmtree(scratch)
    os.makedirs(scratch)
    for item in ("gp", "setup.py", "VERSION.txt", "README.md"):
        src = os.path.join(REF, item)
        dst = os.path.join(scratch, item)
        if os.path.isdir(src):
            shutil.copytree(src, dst)
        else:
            shutil.copy(src, dst)
    subprocess.check_call([sys.executable, "setup.py", "build_ext", "--inplace"],
                          cwd=scratch, stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)


def import_reference(scratch):
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.meta_path.insert(0, _AliasFinder())
    sys.path.insert(0, scratch)
    import gp  # noqa
    return gp


# ---- the reference's own test helpers, restated (gp/tests/util.py) ----------
def rand_params(*args):
    out = []
    for p in args:
        if p == "h":
            out.append(np.random.uniform(0, 2))
        elif p == "w":
            out.append(np.random.uniform(np.pi / 32.0, np.pi / 2.0))
        elif p == "p":
            out.append(np.random.uniform(0.33, 3))
        elif p == "s":
            out.append(np.random.uniform(0, 0.5))
    return tuple(out)


def make_xy():
    x = np.linspace(-2 * np.pi, 2 * np.pi, 16).astype(np.float64)
    return x, np.sin(x)


def make_xo():
    return np.linspace(-2 * np.pi, 2 * np.pi, 32).astype(np.float64)


def seed():
    np.random.seed(2348)


def gp_record(g, xo, derivs=True):
    """Every public hot-path quantity of one GP instance."""
    rec = {
        "x": g.x, "y": g.y, "s": np.float64(g.s), "params": g.params, "xo": xo,
        "Kxx": g.Kxx, "Lxx": g.Lxx, "inv_Kxx": g.inv_Kxx,
        "inv_Kxx_y": g.inv_Kxx_y, "log_lh": np.float64(g.log_lh),
        "lh": np.float64(g.lh),
        "Kxoxo": g.Kxoxo(xo), "Kxxo": g.Kxxo(xo), "Kxox": g.Kxox(xo),
        "mean": g.mean(xo), "cov": g.cov(xo),
    }
    if derivs:
        rec.update({
            "Kxx_J": g.Kxx_J, "Kxx_H": g.Kxx_H,
            "dloglh_dtheta": g.dloglh_dtheta, "dlh_dtheta": g.dlh_dtheta,
            "d2lh_dtheta2": g.d2lh_dtheta2, "dm_dtheta": g.dm_dtheta(xo),
        })
    return rec


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print("wrote", path, "%.1f KiB" % (os.path.getsize(path) / 1024.0))


def flatten(prefix, rec):
    return {"%s__%s" % (prefix, k): np.asarray(v) for k, v in rec.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scratch", default="/tmp/gpref_build")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)

    build_reference(args.scratch)
    gp = import_reference(args.scratch)
    GP, GaussianKernel, PeriodicKernel = gp.GP, gp.GaussianKernel, gp.PeriodicKernel

    meta = {"reference_version": open(os.path.join(REF, "VERSION.txt")).read().strip(),
            "numpy": np.__version__}
    import scipy
    meta["scipy"] = scipy.__version__

    # -- 0. the seeded parameter stream (util.py:15-28,47-48) ------------------
    seed()
    meta["first_hws"] = list(rand_params("h", "w", "s"))

    # -- 1. kernel matrices + derivative stacks on the kernel-test inputs ------
    # test_gaussian_kernel.py:44-64 (x = linspace(-2,2,10), 100 seeded (h,w))
    seed()
    x = np.linspace(-2, 2, 10)
    n_k = 24
    g_params, g_K, g_J, g_H = [], [], [], []
    for _ in range(n_k):
        h, w = rand_params("h", "w")
        k = GaussianKernel(h, w)
        g_params.append(k.params)
        g_K.append(k(x, x))
        g_J.append(k.jacobian(x, x))
        g_H.append(k.hessian(x, x))
    save("gaussian_kernel.npz", x=x, params=np.array(g_params), K=np.array(g_K),
         J=np.array(g_J), H=np.array(g_H))

    # test_periodic_kernel.py:47-64 (x = linspace(-2pi,2pi,16), seeded (h,w,p))
    seed()
    x = np.linspace(-2 * np.pi, 2 * np.pi, 16)
    p_params, p_K, p_J, p_H = [], [], [], []
    for _ in range(n_k):
        h, w, p = rand_params("h", "w", "p")
        k = PeriodicKernel(h, w, p)
        p_params.append(k.params)
        p_K.append(k(x, x))
        p_J.append(k.jacobian(x, x))
        p_H.append(k.hessian(x, x))
    save("periodic_kernel.npz", x=x, params=np.array(p_params), K=np.array(p_K),
         J=np.array(p_J), H=np.array(p_H))

    # rectangular + clamp (gaussian_c.pyx:31-34): e < MIN -> exactly 0
    k = GaussianKernel(1.0, 0.01)
    x1 = np.array([0.0]); x2 = np.array([0.3, 0.38])
    k2 = GaussianKernel(0.7, 0.05)
    xa = np.linspace(-3, 3, 37); xb = np.linspace(-1, 2, 23)
    save("gaussian_clamp.npz", x1=x1, x2=x2, params=k.params, K=k(x1, x2),
         xa=xa, xb=xb, params2=k2.params, K2=k2(xa, xb), J2=k2.jacobian(xa, xb),
         H2=k2.hessian(xa, xb))

    # -- 2. GP records on the GP-test inputs (test_gp.py:24-35) ----------------
    xo = make_xo()
    x, y = make_xy()
    recs = {}
    recs.update(flatten("fixed", gp_record(GP(GaussianKernel(1, 1), x, y, s=1), xo)))
    recs.update(flatten("periodic", gp_record(GP(PeriodicKernel(1, 1, 1), x, y, s=0.1), xo)))
    seed()
    n_r = 16
    for i in range(n_r):
        h, w, s = rand_params("h", "w", "s")
        recs.update(flatten("rand%02d" % i, gp_record(GP(GaussianKernel(h, w), x, y, s=s), xo)))
    seed()
    for i in range(6):
        h, w, p, s = rand_params("h", "w", "p", "s")
        recs.update(flatten("prand%02d" % i, gp_record(GP(PeriodicKernel(h, w, p), x, y, s=s), xo)))
    meta["n_rand_gaussian_gp"] = n_r
    meta["n_rand_periodic_gp"] = 6
    save("gp_small.npz", **recs)

    # test_mean (test_gp.py:59-64): with s=0, mean(x) ~ y.  Store the reference's
    # own verdict per seeded GP so the build's 95% rule can be compared.
    seed()
    verdict = []
    for i in range(100):
        h, w, s = rand_params("h", "w", "s")
        g = GP(GaussianKernel(h, w), x, y, s=s)
        g.s = 0
        try:
            ok = bool(np.allclose(g.mean(g.x), g.y, rtol=1e-5))
        except np.linalg.LinAlgError:
            ok = False
        verdict.append(ok)
    meta["test_mean_pass_count_of_100"] = int(sum(verdict))

    # -- 3. the non-PD known-answer case (test_gp.py:298-333) ------------------
    h, w, s = 0.53356762, 2.14797803, 0
    xn = np.array([0.0, 0.3490658503988659, 0.6981317007977318,
                   1.0471975511965976, 1.3962634015954636, 1.7453292519943295,
                   2.0943951023931953, 0.41968261, 0.97349106, 1.51630532,
                   1.77356282, 2.07011378, 2.87018553, 3.70955074, 3.96680824,
                   4.50962249, 4.80617345, 5.06343095, 5.6062452])
    yn = np.array([-5.297411814764175e-16, 2.2887507861169e-16,
                   1.1824308893126911e-15, 1.9743321560961036e-15,
                   3.387047586844716e-15, 3.2612801348363973e-15,
                   2.248201624865942e-15, -3.061735126365188e-05,
                   2.1539042816804896e-05, -3.900581031467468e-05,
                   4.603140942399664e-05, 0.00014852070373963522,
                   -0.011659908151004955, -0.001060998167383152,
                   -0.0002808538329216448, -8.057870658869265e-06,
                   -7.668984947838558e-07, -7.910215881378919e-08,
                   -3.2649468298271893e-10])
    g = GP(GaussianKernel(h, w), xn, yn, s=s)
    raised = {}
    for prop in ("Lxx", "inv_Kxx", "inv_Kxx_y"):
        try:
            getattr(g, prop)
            raised[prop] = None
        except np.linalg.LinAlgError as e:
            raised[prop] = str(e)
    meta["nonpd"] = {"raised": raised, "log_lh": float(g.log_lh), "lh": float(g.lh),
                     "dloglh_all_nan": bool(np.isnan(g.dloglh_dtheta).all())}
    save("gp_nonpd.npz", x=xn, y=yn, params=np.array([h, w, s]), Kxx=g.Kxx)

    # -- 4. larger seeded 1-D cases (BASELINE config 1 and SURVEY section 6) ---
    big = {}
    for n in (256, 1024):
        rng = np.random.RandomState(0)
        xs = np.sort(rng.uniform(-10, 10, n))
        ys = np.sin(xs) + 0.1 * rng.randn(n)
        xos = np.random.RandomState(1).uniform(-10, 10, 64)
        g = GP(GaussianKernel(1.0, 0.5), xs, ys, s=1.0)
        big["n%d__x" % n] = xs
        big["n%d__y" % n] = ys
        big["n%d__xo" % n] = xos
        big["n%d__params" % n] = g.params
        big["n%d__inv_Kxx_y" % n] = g.inv_Kxx_y
        big["n%d__log_lh" % n] = np.float64(g.log_lh)
        big["n%d__mean" % n] = g.mean(xos)
        big["n%d__cov_diag" % n] = np.diag(g.cov(xos)).copy()
        big["n%d__Lxx_diag" % n] = np.diag(g.Lxx).copy()
        big["n%d__Lxx_lastrow" % n] = g.Lxx[-1].copy()
        gpk = GP(PeriodicKernel(1.3, 0.9, 2.1), xs, ys, s=0.7)
        big["n%d__per_params" % n] = gpk.params
        big["n%d__per_inv_Kxx_y" % n] = gpk.inv_Kxx_y
        big["n%d__per_log_lh" % n] = np.float64(gpk.log_lh)
        big["n%d__per_mean" % n] = gpk.mean(xos)
    # the F6 clamp: small noise at N=256 drives logdet below MIN -> -inf
    rng = np.random.RandomState(0)
    xs = np.sort(rng.uniform(-10, 10, 256))
    ys = np.sin(xs) + 0.1 * rng.randn(256)
    g = GP(GaussianKernel(1.0, 0.5), xs, ys, s=0.1)
    meta["n256_s0.1_log_lh"] = float(g.log_lh)
    meta["n256_s0.1_lh"] = float(g.lh)
    save("gp_seeded_1d.npz", **big)

    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
