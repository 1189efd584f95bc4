"""tools/r6_gemm_probe.py -- the trailing update alone (M = 32768) with two workgroups per CU and with one (padded LDS),
with timing-only ablations: where does the k-loop lose its cycles, and what does a lone workgroup reach?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib
from gaussian_processes_amd.device import DeviceBuffer, Event
lib = _lib.load()
def run(M, K, env, reps=3, dtype=_lib.F64):
    for k in list(os.environ):
        if k.startswith("GPX_GEMM_"):
            del os.environ[k]
    os.environ.update(env)
    npdt = np.float64 if dtype == _lib.F64 else np.float32
    n = M + K
    A = DeviceBuffer((n, n), npdt).zero()
    best = 1e9
    for r in range(reps + 1):
        e0, e1 = Event(), Event()
        e0.record()
        _lib.check(lib.gpx_d_syrk_bc(dtype, n, K, A.ptr, n, K, n, A.ptr, n, 0, K, 1024, 1, 0, None))
        e1.record(); e1.sync()
        if r > 0:
            best = min(best, e0.elapsed_ms(e1))
    fl = M * (M + 1) * K
    peak = 78.6 if dtype == _lib.F64 else 157.3
    print("%s M=%6d K=%5d %-45s %8.3f ms %7.2f TF/s %.4f" % ("f64" if dtype == _lib.F64 else "f32", M, K, " ".join("%s=%s" % (k[9:], v) for k, v in sorted(env.items())), best, fl / best / 1e9, fl / best / 1e9 / peak), flush=True)
    A.free()
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
for dt in (_lib.F64, _lib.F32):
    for M in (16384, 32768):
        for K in (1024,) if quick else (512, 1024, 3072):
            run(M, K, {}, dtype=dt)
            run(M, K, {"GPX_GEMM_PAD_LDS": "40000"}, dtype=dt)
if not quick:
    for ab in ("8", "1", "2", "4", "7", "15"):
        run(32768, 1024, {"GPX_GEMM_ABLATE": ab})
        run(32768, 1024, {"GPX_GEMM_ABLATE": ab, "GPX_GEMM_PAD_LDS": "40000"})
