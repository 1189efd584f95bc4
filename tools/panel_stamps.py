"""tools/panel_stamps.py -- per-workgroup timeline of ONE resident-panel launch inside a fit: N [launch index]
(s_memrealtime, 100 MHz: dispatch delay, pre-update, the four steps, end; diagnostic)."""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib, gp, kernels
from gaussian_processes_amd.device import DeviceBuffer, sync
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
at = int(sys.argv[2]) if len(sys.argv) > 2 else 12
d = 8
rng = np.random.RandomState(0)
X = rng.uniform(-10, 10, (N, d)); y = np.sin(X.sum(1) / np.sqrt(d)) + 0.1 * rng.randn(N)
g = gp.GP(kernels.GaussianKernel(1.0, 0.5 * np.sqrt(d)), X, y, s=1.0)
_ = g.log_lh                                           # warm
st = DeviceBuffer((2048 * 16,), np.uint64).zero()
lib.gpx_debug_panel_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.gpx_debug_panel_stamps(st.ptr, at)
g.s = 1.0001
_ = g.log_lh
sync()
lib.gpx_debug_panel_stamps(None, -1)
s = st.to_host().reshape(-1, 16).astype(np.int64)
live = np.nonzero(s[:, 0])[0]
live = live[live < 2000]; t0 = s[live, 0].min()
print("launch %d of the fit: %d workgroups; times in us after the first workgroup's start" % (at, len(live)))
print("  wg   start  preupd   step0   step1   step2   step3     end   xcc/cu")
for w in live:
    r = s[w]
    f = lambda v: "%7.1f" % ((v - t0) / 100.0) if v else "      -"
    if w < 8 or w % 8 == 0 or w == live[-1]:
        print("%4d %s %s %s %s %s %s %s   %x" % (w, f(r[0]), f(r[1]), f(r[2]), f(r[3]), f(r[4]), f(r[5]), f(r[6]), r[7]))
print("chain hand-offs (workgroup j+1 at step j): flag seen, W staged, X done (+LDS), own update done, X flag raised")
for w in live[:4]:
    r = s[w]
    if r[8]: print("  wg %d: %s" % (w, "  ".join("%7.1f" % ((r[k] - t0) / 100.0) for k in range(8, 13))))
ls = st.to_host().reshape(-1, 16).astype(np.int64)[2040]
if ls[0] and os.environ.get("GPX_LEAF", "4") == "1":
    print("MFMA leaf, step 9 of workgroup 0 (s_memtime cycles): S2 pivot %d, wait %d, S3+S4 %d, wait %d, S5 %d; whole step %d"
          % (ls[1] - ls[0], ls[2] - ls[1], ls[3] - ls[2], ls[4] - ls[3], ls[5] - ls[4], ls[6] - ls[5]))
elif ls[0] and os.environ.get("GPX_LEAF", "4") in ("2", "3"):
    # factor64_mfma2: [0] B1 passed, [1] pivot lane has D(jt+1), [2] wave 3 done with S3 + S4, [3] B2 passed, [4] pivot lane has
    # published the next inverse, [5] wave 3 done with S5 + hand-over, [6] the same point one step later (thread 0)
    print("MFMA leaf v2, step 9 of workgroup 0 (s_memtime cycles after B1): pivot D(jt+1) %d | wave 3 S3+S4 %d | B2 %d | pivot factor+publish %d "
          "(+%d after B2) | wave 3 S5+hand-over %d (+%d after B2) | whole step %d"
          % (ls[1] - ls[0], ls[2] - ls[0], ls[3] - ls[0], ls[4] - ls[0], ls[4] - ls[3], ls[5] - ls[0], ls[5] - ls[3], ls[7] - ls[0]))
lsx = st.to_host().reshape(-1, 16).astype(np.int64)[2040:2048].ravel()
if lsx[16] and lsx[17]:
    # wave 0 of the leaf (factor64_wave): [16] / [17] = s_memrealtime (100 MHz) at its start / end, [32] / [33] = s_memtime (core clock)
    us = (lsx[17] - lsx[16]) / 100.0
    print("leaf of workgroup 0 (wave 0, factor64_wave): %.2f us for 16 steps = %.2f us a step; %d core cycles a step; clock %.2f GHz" % (
        us, us / 16, (lsx[33] - lsx[32]) // 16, (lsx[33] - lsx[32]) / (us * 1e3)))
ends = (s[live, 6] - t0) / 100.0; starts = (s[live, 0] - t0) / 100.0
print("last start %.1f us, last end %.1f us (workgroup %d)" % (starts.max(), ends.max(), live[ends.argmax()]))
