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
NF = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # > 0: FULL product M x NF instead of the lower triangle
NN, TRI = (NF, 0) if NF else (M, 1)
rng = np.random.RandomState(0)
A = DeviceBuffer.from_host(rng.randn(M, K))
C = DeviceBuffer((M, NN)).zero()
nblocks = 4 * 1024 * 1024
S = DeviceBuffer((nblocks * 8,), np.uint64).zero()
for rep in range(2):
    _lib.check(lib.gpx_d_gemm_nt(_lib.F64, M, NN, K, -1.0, A.ptr, K, A.ptr, K, C.ptr, NN, TRI, 0, 0, None))
sync()
S.zero()
from gaussian_processes_amd.device import Event
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # sustained load before the stamped launch
for rep in range(REPS - 1):
    _lib.check(lib.gpx_d_gemm_nt(_lib.F64, M, NN, K, -1.0, A.ptr, K, A.ptr, K, C.ptr, NN, TRI, 0, 0, None))
lib.gpx_debug_gemm_stamps(S.ptr)
e0, e1 = Event(), Event()
e0.record(None)
_lib.check(lib.gpx_d_gemm_nt(_lib.F64, M, NN, K, -1.0, A.ptr, K, A.ptr, K, C.ptr, NN, TRI, 0, 0, None))
e1.record(None)
sync()
wall_ms = e0.elapsed_ms(e1)
lib.gpx_debug_gemm_stamps(None)
st = S.to_host().reshape(-1, 8).astype(np.int64)
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
print("wall %.3f ms  %.2f TF/s" % (wall_ms, (M * (M + 256.0) if TRI else 2.0 * M * NN) * K / wall_ms / 1e9))
for nm, v in (("prologue", pro), ("k-loop", loop), ("epilogue", epi), ("total", st[:, 3] - st[:, 0])):
    print("%-9s mean %9.0f  p10 %9.0f  p50 %9.0f  p90 %9.0f  max %9.0f" % (nm, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
nk = K // 16
print("ideal k-loop cycles at 2 waves/SIMD: %d (=%d k-steps x 8192)" % (nk * 8192, nk))
busy = (st[:, 3] - st[:, 0]).sum() / 256.0
print("sum(tile total)/256 CUs = %.0f cycles = %.3f ms at the measured clock" % (busy, busy / clk.mean() / 1e3))

# per-CU timeline: HW_ID bits cu_id[11:8] sh_id[12] se_id[15:13]; XCC_ID[3:0]
hw = st[:, 6] & 0xFFFFFFFF
xcc = (st[:, 6] >> 32) & 0xF
cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
ids = np.unique(cu)
busy_frac, gaps, conc = [], [], []
t0, t1 = st[:, 4].min(), st[:, 5].max()
for c in ids:
    t = st[cu == c]
    # union / sum of [start, end) in the 100 MHz clock
    ev = sorted([(a, 1) for a in t[:, 4]] + [(b, -1) for b in t[:, 5]])
    depth, last, hist = 0, t0, {}
    for (x, d) in ev:
        hist[depth] = hist.get(depth, 0) + (x - last)
        depth += d; last = x
    hist[0] = hist.get(0, 0) + (t1 - last)
    tot = float(t1 - t0)
    conc.append([hist.get(k, 0) / tot for k in range(4)])
conc = np.array(conc)
print("distinct CUs %d; fraction of the launch a CU holds 0 / 1 / 2 / 3 workgroups (mean over CUs): %s" % (len(ids), " / ".join("%.3f" % v for v in conc.mean(0))))

# refill latency of a workgroup slot: for every tile end, the next tile start on the same CU
refill = []
for c in ids:
    t = st[cu == c]
    starts = np.sort(t[:, 4]); ends = np.sort(t[:, 5])
    j = np.searchsorted(starts, ends, side="left")
    ok = j < len(starts)
    refill.extend(((starts[j[ok]] - ends[ok]) / 100.0).tolist())      # us
refill = np.array(refill)
print("slot refill latency (tile end -> next tile start on that CU) [us]: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  p99 %.2f" % (refill.mean(), np.percentile(refill, 10), np.percentile(refill, 50), np.percentile(refill, 90), np.percentile(refill, 99)))
dur = (st[:, 5] - st[:, 4]) / 100.0
print("tile duration [us]: mean %.1f p10 %.1f p50 %.1f p90 %.1f; tiles shorter than 5 us: %d" % (dur.mean(), np.percentile(dur, 10), np.percentile(dur, 50), np.percentile(dur, 90), (dur < 5).sum()))
