"""tools/fuzz_fit.py [cases] [seed] -- random sizes / dtypes / kernels / switches through gp.GP against the oracle (diagnostic;
the parametrised tests in tests/ are the contract, this looks for the case nobody wrote down)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gaussian_processes_amd as gp
from gaussian_processes_amd import mlii
from oracle import gp_oracle as orc
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
SW = [{}, {"GPX_FIT_OPS_AHEAD_MIN": "1024", "GPX_FIT_OPS_TAIL": "1"}, {"GPX_POTRF_TWO_PART_ROWS": "0", "GPX_POTRF_TWO_PART_BATCH": "0"},
      {"GPX_FIT_RIDE_MAX": "0", "GPX_TRSV_OPS_MIN": "1024"}, {"GPX_POTRF_NB": "512"}, {"GPX_POTRF_HOST_PACED": "0"},
      {"GPX_POTRF_FOLD_ROWS": "0"}, {"GPX_LEAF": "1"}, {"GPX_LEAF": "5"}]
bad = 0
for c in range(cases):
    n = int(rng.choice([rng.randint(1, 300), rng.randint(300, 2200), 512 * rng.randint(1, 9), rng.randint(2200, 5200)]))
    d = int(rng.randint(1, 6))
    dtype = "float64" if rng.rand() < 0.6 else "float32"
    per = rng.rand() < 0.2 and d == 1
    sw = SW[rng.randint(len(SW))]
    for k in list(os.environ):
        if k.startswith("GPX_") and k not in ("GPX_TRACE",): del os.environ[k]
    os.environ.update(sw)
    X, y, Xo = orc.synth_inputs(n, d, 8)
    h, w, s = rng.uniform(0.5, 2), rng.uniform(0.3, 1.5) * np.sqrt(d), rng.uniform(0.8, 2.0)
    if per:
        kern, ok = gp.PeriodicKernel(h, w, 3.0), orc.OracleGP("periodic", (h, w, 3.0), X, y, s)
    else:
        kern, ok = gp.GaussianKernel(h, w), orc.OracleGP("gaussian", (h, w), X, y, s)
    g = gp.GP(kern, X, y, s=s, dtype=dtype)
    rt, rta = (1e-9, 1e-7) if dtype == "float64" else (2e-4, 5e-3)
    try:
        ll, llo = float(g.log_lh), float(ok.log_lh)
        if np.isneginf(llo):        # logdet below MIN (gp_c.pyx:22-25): both must say so
            assert np.isneginf(ll), ("log_lh", ll, llo)
        else:
            assert abs(ll - llo) <= rt * max(1.0, abs(llo)), ("log_lh", ll, llo)
        a, ao = np.asarray(g.inv_Kxx_y, dtype=np.float64), ok.inv_Kxx_y
        assert np.abs(a - ao).max() <= rta * max(1e-30, np.abs(ao).max()), ("alpha", np.abs(a - ao).max(), np.abs(ao).max())
        m, mo = np.asarray(g.mean(Xo), dtype=np.float64), ok.mean(Xo)
        assert np.abs(m - mo).max() <= rta * max(1.0, np.abs(mo).max()), ("mean", np.abs(m - mo).max())
        if n >= 2 and rng.rand() < 0.3:
            th = np.array([list(kern.params) + [s]] * 3, dtype=np.float64); th[1, 0] *= 1.1; th[2, -1] *= 0.9
            if not per:
                lb = mlii.log_lh_batch(X, y, th, dtype=dtype)
                assert (np.isneginf(lb[0]) and np.isneginf(llo)) or abs(lb[0] - llo) <= rt * max(1.0, abs(llo)), ("batch", lb[0], llo)
        print("ok   n=%5d d=%d %-7s %-8s %s" % (n, d, dtype, "periodic" if per else "gaussian", sw))
    except AssertionError as e:
        bad += 1
        print("FAIL n=%5d d=%d %-7s %-8s %s: %s" % (n, d, dtype, "periodic" if per else "gaussian", sw, e))
print("%d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
