"""tools/gemm_stamps.py -- where does a tile of the fast GEMM spend its cycles? (diagnostic)"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib
from gaussian_processes_amd.device import DeviceBuffer, sync
lib = _lib.load()
lib.gpx_debug_gemm_stamps.argtypes = [ctypes.c_void_p]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
rng = np.random.RandomState(0)
A = DeviceBuffer.from_host(rng.randn(M, K))
C = DeviceBuffer((M, M)).zero()
nblocks = 4 * 1024 * 1024
S = DeviceBuffer((nblocks * 6,), np.uint64).zero()
for rep in range(2):
    _lib.check(lib.gpx_d_gemm_nt(_lib.F64, M, M, K, -1.0, A.ptr, K, A.ptr, K, C.ptr, M, 1, 0, 0, None))
sync()
S.zero()
from gaussian_processes_amd.device import Event
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # sustained load before the stamped launch
for rep in range(REPS - 1):
    _lib.check(lib.gpx_d_gemm_nt(_lib.F64, M, M, K, -1.0, A.ptr, K, A.ptr, K, C.ptr, M, 1, 0, 0, None))
lib.gpx_debug_gemm_stamps(S.ptr)
e0, e1 = Event(), Event()
e0.record(None)
_lib.check(lib.gpx_d_gemm_nt(_lib.F64, M, M, K, -1.0, A.ptr, K, A.ptr, K, C.ptr, M, 1, 0, 0, None))
e1.record(None)
sync()
wall_ms = e0.elapsed_ms(e1)
lib.gpx_debug_gemm_stamps(None)
st = S.to_host().reshape(-1, 6).astype(np.int64)
bids = np.nonzero(st[:, 3] > 0)[0]
st = st[st[:, 3] > 0]
t_begin = st[:, 4].min()
print("per XCD (= blockIdx % 8): tiles, busy ms (sum of tile times / 32 CUs), first start, last end [ms, 100 MHz clock]")
for xcd in range(8):
    sel = (bids & 7) == xcd
    if not sel.any():
        continue
    t = st[sel]
    print("  xcd %d: %5d tiles  busy %.3f  start %.3f  end %.3f" % (xcd, sel.sum(), (t[:, 5] - t[:, 4]).sum() / 32 / 1e5, (t[:, 4].min() - t_begin) / 1e5, (t[:, 5].max() - t_begin) / 1e5))
pro, loop, epi = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
print("tiles", len(st))
clk = (st[:, 3] - st[:, 0]) / np.maximum(st[:, 5] - st[:, 4], 1) * 100.0
print("shader clock from s_memtime / s_memrealtime per tile: mean %.0f MHz  p10 %.0f  p90 %.0f" % (clk.mean(), np.percentile(clk, 10), np.percentile(clk, 90)))
print("realtime span of the launch: %.3f ms" % ((st[:, 5].max() - st[:, 4].min()) / 1e5))
print("wall %.3f ms  %.2f TF/s" % (wall_ms, M * (M + 256.0) * K / wall_ms / 1e9))
for nm, v in (("prologue", pro), ("k-loop", loop), ("epilogue", epi), ("total", st[:, 3] - st[:, 0])):
    print("%-9s mean %9.0f  p10 %9.0f  p50 %9.0f  p90 %9.0f  max %9.0f" % (nm, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
nk = K // 16
print("ideal k-loop cycles at 2 waves/SIMD: %d (=%d k-steps x 8192)" % (nk * 8192, nk))
busy = (st[:, 3] - st[:, 0]).sum() / 256.0
print("sum(tile total)/256 CUs = %.0f cycles = %.3f ms at the measured clock" % (busy, busy / clk.mean() / 1e3))
