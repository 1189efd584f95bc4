"""tools/leaf_time.py -- the 64 x 64 leaf alone: time per call and (pivot-wave kernel) s_memtime stamps of every step
of every wave: [0] after barrier 1, [1] end of B, [2] after barrier 2, [3] end of C / pivot work (diagnostic)."""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib
from gaussian_processes_amd.device import DeviceBuffer, Event, sync
lib = _lib.load()
n = 64
rng = np.random.RandomState(0)
B = rng.randn(n, n); A = B @ B.T + n * np.eye(n)
info = DeviceBuffer((4,), np.int32).zero()
dA = DeviceBuffer.from_host(A)
e0, e1 = Event(), Event(); e0.record()
for i in range(200):
    lib.gpx_d_potrf(_lib.F64, dA.ptr, n, n, info.ptr, None)
e1.record(); e1.sync()
print("ABLATE=%s PIPE=%s: %.2f us per 64x64 potrf call (incl. memset + launch)" % (os.environ.get("GPX_LEAF_ABLATE"), os.environ.get("GPX_LEAF_PIPE"), e0.elapsed_ms(e1) * 1e3 / 200))
if os.environ.get("GPX_LEAF_PIPE", "1") != "0":
    st = DeviceBuffer((5 * 16 * 4,), np.uint64).zero()
    lib.gpx_debug_leaf_stamps.argtypes = [ctypes.c_void_p]
    lib.gpx_debug_leaf_stamps(st.ptr)
    dA = DeviceBuffer.from_host(A)
    lib.gpx_d_potrf(_lib.F64, dA.ptr, n, n, info.ptr, None); sync()
    lib.gpx_debug_leaf_stamps(None)
    s = st.to_host().reshape(5, 16, 4).astype(np.int64)
    t0 = s[:, 0, 0].min()
    print("cycles (s_memtime, 100 MHz units x ?): per step, wave 0 / wave 3 / pivot wave: B, wait2, C-or-pivot, wait1(next)")
    for jt in range(16):
        row = []
        for w in (0, 3, 4):
            b = s[w, jt, 1] - s[w, jt, 0]; w2 = s[w, jt, 2] - s[w, jt, 1]; c = s[w, jt, 3] - s[w, jt, 2]
            w1 = (s[w, jt + 1, 0] - s[w, jt, 3]) if jt < 15 else 0
            row.append("%5d %5d %5d %5d" % (b, w2, c, w1))
        print("step %2d | %s | %s | %s | step total %d" % (jt, row[0], row[1], row[2], (s[0, jt + 1, 0] - s[0, jt, 0]) if jt < 15 else 0))
