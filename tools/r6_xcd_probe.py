"""tools/r6_xcd_probe.py -- does blockIdx % 8 still name the XCD when the trailing update runs IN SITU (beside the panel stream)?
Per-workgroup stamps (gpx_debug_gemm_stamps: XCC_ID of every tile) of one N = 65536 fit; the block indices that only the
first, largest update reaches are looked at."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_processes_amd as gp
from gaussian_processes_amd import _lib
from gaussian_processes_amd.device import DeviceBuffer, sync
from bench import synth
lib = _lib.load()
lib.gpx_debug_gemm_stamps.argtypes = [ctypes.c_void_p]
N, d = 65536, 32
X, y, _ = synth(N, d, 4, np.float64)
g = gp.GP(gp.GaussianKernel(1.0, 0.5 * np.sqrt(d)), X, y, s=1.0)
float(g.log_lh)                                     # warm
nblocks = 1 << 21
S = DeviceBuffer((nblocks * 8,), np.uint64).zero()
lib.gpx_debug_gemm_stamps(S.ptr)
g.set_param("h", 1.0 + 1e-13)
llh = float(g.log_lh)
lib.gpx_debug_gemm_stamps(None)
sync()
st = S.to_host().reshape(-1, 8).astype(np.int64)
lo, hi = 119808, 123392                              # block indices only the first update (123392 tiles) has
sel = st[lo:hi]
sel_ok = sel[:, 3] > 0
xcc = (sel[:, 6] >> 32) & 0xF
bid = np.arange(lo, hi)
print("tiles looked at: %d (stamped %d), log_lh %.6f" % (hi - lo, sel_ok.sum(), llh))
tab = np.zeros((8, 8), dtype=int)
for b, x, ok in zip(bid, xcc, sel_ok):
    if ok:
        tab[b & 7, x & 7] += 1
print("rows: blockIdx % 8, columns: XCC_ID")
print(tab)
perm_ok = all((tab[r] > 0).sum() == 1 for r in range(8))
print("blockIdx % 8 names ONE XCD for every residue:", perm_ok)
dur = (sel[sel_ok][:, 5] - sel[sel_ok][:, 4]) / 100.0
print("tile duration in situ [us]: mean %.1f p10 %.1f p50 %.1f p90 %.1f" % (dur.mean(), np.percentile(dur, 10), np.percentile(dur, 50), np.percentile(dur, 90)))
