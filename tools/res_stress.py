"""tools/res_stress.py -- the resident panel kernel over odd shapes: every number of steps (64 .. 256 columns), rows that
end inside / exactly at a workgroup, a last panel with nothing below it, panels narrower than the matrix; fp64 and fp32,
against scipy (diagnostic; the same cases are in tests/test_gpu_parity.py in a smaller number)."""
import os, sys
import numpy as np, scipy.linalg
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib
from gaussian_processes_amd.device import DeviceBuffer, sync
lib = _lib.load()
rng = np.random.RandomState(0)
worst = {}
cases = 0
for nb in (64, 128, 192, 256, 320, 512):
    os.environ["GPX_POTRF_NB"] = str(nb)
    for n in (64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 319, 320, 321, 383, 384, 448, 511, 512, 513, 640, 767, 768, 769,
              1000, 1024, 1025, 1280, 1663, 1664, 2047, 2048, 2049, 2240, 3000):
        ld = (n + 15) // 16 * 16
        B = rng.randn(n, n)
        A = B @ B.T + n * np.eye(n)
        Lref = scipy.linalg.cholesky(A, lower=True)
        for dt, npdt, tol in ((_lib.F64, np.float64, 1e-11), (_lib.F32, np.float32, 3e-4)):
            Ap = np.zeros((n, ld), dtype=npdt); Ap[:, :n] = A
            dA = DeviceBuffer.from_host(Ap)
            info = DeviceBuffer((4,), np.int32).zero()
            _lib.check(lib.gpx_d_potrf(dt, dA.ptr, n, ld, info.ptr, None))
            L = np.tril(dA.to_host()[:, :n].astype(np.float64))
            i = int(info.to_host()[0])
            err = np.abs(L - Lref).max() / np.abs(Lref).max()
            cases += 1
            key = ("f64" if dt == _lib.F64 else "f32")
            worst[key] = max(worst.get(key, 0.0), err)
            if i != 0 or not (err < tol):
                print("FAIL nb=%d n=%d %s info=%d err=%.2e" % (nb, n, key, i, err), flush=True)
            dA.free(); info.free()
print("%d cases; worst relative error f64 %.2e, f32 %.2e" % (cases, worst["f64"], worst["f32"]))
