// gpx_potrf.hip -- blocked lower Cholesky, in place, row-major (gfx950).
//
// Replaces scipy.linalg.cholesky(Kxx, lower=True) of gp/gp.py:294 (LAPACK
// dpotrf; the arithmetic is not in the reference tree).  Non-positive-definite
// input is reported LAPACK-style through `info` (first failing leading minor,
// 1-based), which the host maps to numpy.linalg.LinAlgError exactly as the
// reference's callers expect (gp/gp.py:362-365, gp/tests/test_gp.py:322-327).
//
// Algorithm (N x N, outer block nb, inner block IB = 64):
//   for each block column k0 (width kb <= nb)                      -- right-looking
//     panel: recursive halving down to 64 columns (see potrf_panel_t)
//       (a) A[j0:, j0:j0+64] -= A[j0:, k0:j0] * A[j0:j0+64, k0:j0]^T    MFMA gemm_nt
//       (b) factor the 64 x 64 diagonal block in registers (one workgroup, 4-column steps)
//       (c) rows below: A[r, j0:j0+64] <- A[r, j0:j0+64] * Ljj^-T       (one lane per row, or for
//           short panels a fused 64 x 64 inverse + one MFMA product)
//     trailing update A[k0+kb:, k0+kb:] -= P * P^T, P = A[k0+kb:, k0:k0+kb]  MFMA gemm_nt, lower
// >= 97 % of the flops at N = 65536 are in the trailing update (K = nb deep).
#include "gpx_common.h"
#include "gpx_leaf.h"
#include <vector>

namespace gpx {

constexpr int IBP = IB + 1;   // LDS pitch (elements): conflict-free column walks
constexpr int LEAN_TMAX = 3;  // lean panel route: at most 3 blocks of 64 columns to the right of a step (panels <= 256 wide)

// ---- (b) diagonal block: right-looking Cholesky in 4-column steps, register resident ----
// 256 threads hold the 64 x 64 block as 16 x 16 register tiles of 4 x 4 (thread
// (tr, tc) owns rows 4tr.., columns 4tc..).  Step jt of 16, two barriers:
//   A  thread (jt, jt) factors its own 4 x 4 tile (four dependent rsqrt chains: the
//      critical path of the kernel) and publishes it with the reciprocal pivots;
//   B  the 15 - jt threads below it in tile column jt solve their tiles against it
//      (P <- P L_dd^-T) and publish the 4 finished columns of L;
//   C  every tile below and to the right gets its rank-4 update from LDS.
// (The unblocked form -- one pivot, one barrier per column -- took 24 us per block.)
// Scaling uses the reciprocal of the pivot's square root, as LAPACK's dpotf2 does.
// Blocks smaller than 64 are padded with the identity (pivot 1, no effect).
// When `inv` is non-null the same sweep also carries X = L^-1 (forward elimination of
// the identity: in step jt the four rows of X are solved against the diagonal tile and
// then subtracted, times L, from the rows below) and stores it as a dense 64 x 64
// row-major block: the row substitution below the leaf then becomes one small MFMA
// product X_rows * inv(L)^T.
// lean panel route: the leaf also saves the rows below its block that are the NEXT diagonal blocks of the panel
// (rd_rows of them, 64 columns) as they are BEFORE the substitution, so that every workgroup of the row kernel
// can recompute their substituted values without racing against the workgroup that stores them in place
template <typename T>
__device__ __forceinline__ void copy_rows_below(const T *__restrict__ blk, int64_t lda, T *__restrict__ rd, int rd_rows)
{
    if (!rd || rd_rows <= 0) return;
    rd += (int64_t)blockIdx.x * (LEAN_TMAX * IB * IB);
    for (int idx = threadIdx.x; idx < rd_rows * IB; idx += blockDim.x) {
        const int r = idx / IB, c = idx - r * IB;
        rd[idx] = blk[(int64_t)(IB + r) * lda + c];
    }
}

template <typename T, bool INV>
__global__ __launch_bounds__(256) void potrf_diag_kernel(T *__restrict__ blk, int64_t lda, int64_t j0,
                                                         int jb, int *__restrict__ info, T *__restrict__ inv,
                                                         int64_t sblk, T *__restrict__ rd = nullptr, int rd_rows = 0)
{
    // batched launches: workgroup b factors the block of matrix b (stride sblk), own info word and inverse
    blk += (int64_t)blockIdx.x * sblk;
    info += blockIdx.x;
    if (INV) inv += (int64_t)blockIdx.x * (IB * IB);
    copy_rows_below<T>(blk, lda, rd, rd_rows);
    // blk: the jb x jb diagonal block; j0: its global (0-based) index, for `info`
    // this one workgroup is the critical path of the whole panel and usually shares its CU with
    // trailing-update workgroups of the other stream: take the instruction arbiter's top priority
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x;
    const int tr = tid >> 4, tc = tid & 15;
    // an earlier block already failed: the factor is garbage from there on, do not spend the chain on it
    __shared__ int s_abort;
    if (tid == 0) s_abort = *info;
    __syncthreads();
    if (s_abort != 0) return;
    T a[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = 4 * tr + r, col = 4 * tc + c;
            T v = (row == col) ? (T)1 : (T)0;
            if (row < jb && col <= row) v = blk[(int64_t)row * lda + col];
            a[r][c] = v;
        }
    T x[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) x[r][c] = (4 * tr + r == 4 * tc + c) ? (T)1 : (T)0;
    factor64<T, INV>(a, x, jb, j0, info);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = 4 * tr + r, col = 4 * tc + c;
            if (row < jb && col <= row) blk[(int64_t)row * lda + col] = a[r][c];
            if (INV) inv[row * IB + col] = (col <= row) ? x[r][c] : (T)0;
        }
}

static thread_local bool g_leaf_pipe = false;            // set by potrf(): CUs are reserved for the panel stream
static unsigned long long *g_leaf_stamps = nullptr;     // diagnostic (gpx_debug_leaf_stamps): 5 waves x 16 steps x 4 stamps
// the leaf with the pivot wave (gpx_leaf.h): 320 threads
template <typename T, bool INV>
__global__ __launch_bounds__(320) void potrf_diag_pipe_kernel(T *__restrict__ blk, int64_t lda, int64_t j0, int jb,
                                                              int *__restrict__ info, T *__restrict__ inv, int64_t sblk,
                                                              int nsteps, unsigned long long *stamps,
                                                              T *__restrict__ rd = nullptr, int rd_rows = 0)
{
    blk += (int64_t)blockIdx.x * sblk;
    info += blockIdx.x;
    if (INV) inv += (int64_t)blockIdx.x * (IB * IB);
    copy_rows_below<T>(blk, lda, rd, rd_rows);
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x;
    __shared__ int s_abort;
    if (tid == 0) s_abort = *info;
    __syncthreads();
    if (s_abort != 0) return;
    const bool pivot = tid >= 256;
    // tile coordinates: waves 0-3 own (tr, tc); wave 4 lane t owns (t, t) (lanes >= 16 idle)
    const int tr = pivot ? (tid - 256) : (tid >> 4), tc = pivot ? (tid - 256) : (tid & 15);
    const bool live = !pivot || (tid - 256) < IB / 4;
    T a[4][4], x[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = 4 * tr + r, col = 4 * tc + c;
            T v = (row == col) ? (T)1 : (T)0;
            if (live && row < jb && col <= row) v = blk[(int64_t)row * lda + col];
            a[r][c] = v;
            x[r][c] = (row == col) ? (T)1 : (T)0;
        }
    factor64_pipe<T, INV>(a, x, jb, j0, info, leaf_role_320(tid), nsteps, stamps);
    if (!live) return;
    const bool own_a = pivot || tc < tr;              // waves 0-3 own the strictly-lower tiles, wave 4 the diagonal ones
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = 4 * tr + r, col = 4 * tc + c;
            if (own_a && row < jb && col <= row) blk[(int64_t)row * lda + col] = a[r][c];
            if (INV && !pivot) inv[row * IB + col] = (col <= row) ? x[r][c] : (T)0;
        }
}

// ---- fused panel step: one launch per 64 columns of a panel ---------------------------------------
// Replaces, per 64-column step of a panel, the chain  leaf -> row substitution -> in-panel update(s)  (3 - 4
// dependent launches of 12 - 30 us each beside a running trailing update) by ONE launch, left-looking inside
// the panel.  Step s of the panel at (r0p, c0p), columns [c0, c0 + 64), kin = 64 s columns to its left:
//   workgroup 0      D' = D - Lp Lp^T (Lp = the diagonal block's rows of the panel columns to the left, MFMA),
//                    factors D' and forms X = inv(L) in registers (factor64), stores L, publishes X and
//                    raises a flag (agent-scope release);
//   workgroup w > 0  64 rows below: R' = R - (their rows of the panel columns to the left) Lp^T on the MFMA
//                    pipe WHILE workgroup 0 factors, then waits for the flag (one lane polls, bounded spin,
//                    agent-scope acquire) and stores Y = R' X^T.
// Operands go global/L2 -> registers directly in MFMA fragment layout (each is used by one wave only).
// Workgroup 0 is dispatched first, so the producer is resident before any consumer can spin; should that
// ever fail the spin is bounded and the step reports info = -7 instead of hanging.
constexpr int PS_SPIN = 1 << 21;          // polls of ~0.3 us: gives up after ~0.6 s

template <typename T>
__global__ __launch_bounds__(256) void panel_step_kernel(T *__restrict__ A, int64_t lda, int64_t n, int64_t r0,
                                                         int64_t c0, int kin, int *__restrict__ info,
                                                         T *__restrict__ xinv, int *__restrict__ flag, int serial,
                                                         int64_t sM)
{
    typedef PM<T> M;
    typedef typename M::v4 v4;
    constexpr int EPK = M::EPK, SUB = M::SUB;
    A += (int64_t)blockIdx.y * sM; info += blockIdx.y; xinv += (int64_t)blockIdx.y * (IB * IB); flag += blockIdx.y;
    __shared__ T sT[IB][IB + 2];          // D' (workgroup 0) / R' : accumulator layout -> rows
    __shared__ int s_ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const bool diag = blockIdx.x == 0;
    if (diag) __builtin_amdgcn_s_setprio(3);
    if (tid == 0) s_ok = *info;
    __syncthreads();
    if (s_ok != 0) {                       // an earlier step failed: nothing to wait for, nothing to compute
        if (diag && tid == 0) __hip_atomic_store(flag, serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int64_t wr0 = (diag ? r0 : r0 + IB + (int64_t)(blockIdx.x - 1) * IB) + 16 * wave;   // this wave's 16 rows
    // ---- P = (own rows of the panel columns to the left) . Lp^T ----
    v4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] = (T)0;
    {
        const T *arow = A + min(wr0 + li, n - 1) * lda + (c0 - kin) + lq * SUB;
        const T *brow = A + (r0 + li) * lda + (c0 - kin) + lq * SUB;
        for (int k = 0; k < kin; k += EPK) {
            T fa[SUB], fb[4][SUB];
            load_frag32<T>(arow + k, fa);
#pragma unroll
            for (int j = 0; j < 4; ++j) load_frag32<T>(brow + (int64_t)(16 * j) * lda + k, fb[j]);
#pragma unroll
            for (int ss = 0; ss < SUB; ++ss)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = M::mfma(fa[ss], fb[j][ss], acc[j]);
        }
    }
    // ---- R' = R - P, into LDS by rows ----
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int lr = M::row(lane, r);
            const int64_t gr = wr0 + lr;
            const T v = (gr < n) ? A[gr * lda + c0 + 16 * j + li] : (T)0;
            sT[16 * wave + lr][16 * j + li] = v - acc[j][r];
        }
    __syncthreads();
    if (diag) {
        const int tr = tid >> 4, tc = tid & 15;
        T a[4][4], x[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int row = 4 * tr + r, col = 4 * tc + c;
                a[r][c] = (col <= row) ? sT[row][col] : (T)0;
                x[r][c] = (row == col) ? (T)1 : (T)0;
            }
        factor64<T, true>(a, x, IB, r0, info);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int row = 4 * tr + r, col = 4 * tc + c;
                if (col <= row) A[(r0 + row) * lda + c0 + col] = a[r][c];
                xinv[row * IB + col] = (col <= row) ? x[r][c] : (T)0;
            }
        // publish: every storing wave drains, the workgroup meets, ONE lane releases at agent scope and raises the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(flag, serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    // ---- consumers: wait for X (one relaxed poll loop, then ONE agent-scope acquire, then the barrier) ----
    if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != serial && spins < PS_SPIN) {
            __builtin_amdgcn_s_sleep(16);
            ++spins;
        }
        s_ok = spins < PS_SPIN;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (!s_ok) {
        if (tid == 0) atomicCAS(info, 0, -7);
        return;
    }
    // ---- Y = R' X^T ----
    v4 y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) y[j][r] = (T)0;
#pragma unroll
    for (int k = 0; k < IB; k += EPK) {
        T fa[SUB], fb[4][SUB];
#pragma unroll
        for (int ss = 0; ss < SUB; ++ss) fa[ss] = sT[16 * wave + li][k + lq * SUB + ss];
#pragma unroll
        for (int j = 0; j < 4; ++j) load_frag32<T>(xinv + (16 * j + li) * IB + k + lq * SUB, fb[j]);
#pragma unroll
        for (int ss = 0; ss < SUB; ++ss)
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = M::mfma(fa[ss], fb[j][ss], y[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gr = wr0 + M::row(lane, r);
            if (gr < n) A[gr * lda + c0 + 16 * j + li] = y[j][r];
        }
}

// ---- lean panel route: two launches per 64 panel columns, right-looking inside the panel ------------------
// After the leaf (L_ss, X = inv(L_ss), and the saved rows Rd of the next diagonal blocks), ONE row kernel does for 64
// rows per workgroup what used to be the substitution launch plus one to three in-panel update launches:
//   Y = R X^T                              (its rows of the step's 64 columns; stored in place)
//   for every block t of 64 columns to the right within the panel:
//       Yd_t = Rd_t X^T                    (the step's columns of that block's DIAGONAL rows -- recomputed by every
//                                           workgroup from the saved copy: 64^3 flops instead of a launch boundary)
//       A[rows, block t] -= Y Yd_t^T
// Operands go global / L2 -> registers in MFMA fragment layout; Y and Yd_t pass through LDS to become operands.
template <typename T>
__global__ __launch_bounds__(256) void panel_rows_kernel(T *__restrict__ A, int64_t lda, int64_t n, int64_t rb,
                                                         int64_t c0, int tblocks, const T *__restrict__ xinv,
                                                         const T *__restrict__ rd, int64_t sM)
{
    typedef PM<T> M;
    typedef typename M::v4 v4;
    constexpr int EPK = M::EPK, SUB = M::SUB, NCH = IB / EPK;
    A += (int64_t)blockIdx.y * sM;
    xinv += (int64_t)blockIdx.y * (IB * IB);
    rd += (int64_t)blockIdx.y * (LEAN_TMAX * IB * IB);
    __shared__ T sY[IB][IB + 2];
    __shared__ T sYd[IB][IB + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int64_t wr0 = rb + (int64_t)blockIdx.x * IB + 16 * wave;          // this wave's 16 rows
    // X as B operand: rows 16 j + li, kept in registers for all products of the workgroup
    T xb[NCH][4][SUB];
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc)
#pragma unroll
        for (int j = 0; j < 4; ++j) load_frag32<T>(xinv + (16 * j + li) * IB + kc * EPK + lq * SUB, xb[kc][j]);
    // ---- Y = R X^T ----
    v4 y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) y[j][r] = (T)0;
    {
        const T *arow = A + min(wr0 + li, n - 1) * lda + c0 + lq * SUB;
        T fa[NCH][SUB];
#pragma unroll
        for (int kc = 0; kc < NCH; ++kc) load_frag32<T>(arow + kc * EPK, fa[kc]);
#pragma unroll
        for (int kc = 0; kc < NCH; ++kc)
#pragma unroll
            for (int ss = 0; ss < SUB; ++ss)
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = M::mfma(fa[kc][ss], xb[kc][j][ss], y[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int lr = M::row(lane, r);
            const int64_t gr = wr0 + lr;
            if (gr < n) A[gr * lda + c0 + 16 * j + li] = y[j][r];
            sY[16 * wave + lr][16 * j + li] = y[j][r];
        }
    // ---- in-panel updates ----
    for (int t = 0; t < tblocks; ++t) {
        v4 yd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) yd[j][r] = (T)0;
        {
            const T *drow = rd + (int64_t)(IB * t + 16 * wave + li) * IB + lq * SUB;
            T fa[NCH][SUB];
#pragma unroll
            for (int kc = 0; kc < NCH; ++kc) load_frag32<T>(drow + kc * EPK, fa[kc]);
#pragma unroll
            for (int kc = 0; kc < NCH; ++kc)
#pragma unroll
                for (int ss = 0; ss < SUB; ++ss)
#pragma unroll
                    for (int j = 0; j < 4; ++j) yd[j] = M::mfma(fa[kc][ss], xb[kc][j][ss], yd[j]);
        }
        __syncthreads();                                   // the previous block's sYd has been consumed; sY is complete
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) sYd[16 * wave + M::row(lane, r)][16 * j + li] = yd[j][r];
        __syncthreads();
        v4 u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) u[j][r] = (T)0;
#pragma unroll
        for (int kc = 0; kc < NCH; ++kc) {
            T fa[SUB], fb[4][SUB];
#pragma unroll
            for (int ss = 0; ss < SUB; ++ss) fa[ss] = sY[16 * wave + li][kc * EPK + lq * SUB + ss];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ss = 0; ss < SUB; ++ss) fb[j][ss] = sYd[16 * j + li][kc * EPK + lq * SUB + ss];
#pragma unroll
            for (int ss = 0; ss < SUB; ++ss)
#pragma unroll
                for (int j = 0; j < 4; ++j) u[j] = M::mfma(fa[ss], fb[j][ss], u[j]);
        }
        const int64_t cu = c0 + (int64_t)IB * (t + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gr = wr0 + M::row(lane, r);
                if (gr < n) A[gr * lda + cu + 16 * j + li] -= u[j][r];
            }
    }
}

// OPT-IN (GPX_POTRF_LEAN=256).  Measured (n = 8192, nb = 256): the row kernel takes 56 / 36 / 26 / 14 us alone for
// 3 / 2 / 1 / 0 blocks to the right (its LDS fragment reads are element-wise and nothing is software-pipelined:
// 4 x the matrix-pipe time) and up to 155 us beside the trailing update, against 12 - 17 us + 14 - 30 us for the
// substitution and update launches of the tuned GEMM kernel it replaces: potrf 10.3 -> 12.1 ms.  Off by default.
static int64_t lean_max()
{
    const int64_t v = getenv("GPX_POTRF_LEAN") ? atoll(getenv("GPX_POTRF_LEAN")) : 0;       // (read per call: tests switch it)
    return std::min<int64_t>(v, (LEAN_TMAX + 1) * IB);
}
static int64_t lean_rows_max()
{
    const int64_t v = getenv("GPX_POTRF_LEAN_ROWS") ? atoll(getenv("GPX_POTRF_LEAN_ROWS")) : 16384;
    return v;
}

// scratch of the fused steps: per matrix of a batch one 64 x 64 inverse and one flag word; the flag only ever
// grows (`serial` is bumped per launch), so nothing is reset between launches
struct FusedScratch { void *p = nullptr; size_t bytes = 0; int device = -1; int serial = 0; };
static thread_local FusedScratch g_fscr;
static int fused_scratch(int nbatch, size_t es, void **xinv, int **flag)
{
    int dev = 0;
    GPX_HIP(hipGetDevice(&dev));
    const size_t need = (size_t)nbatch * (IB * IB * es + 256);
    if (g_fscr.device != dev || g_fscr.bytes < need) {
        if (g_fscr.p && g_fscr.device == dev) { GPX_HIP(hipDeviceSynchronize()); (void)hipFree(g_fscr.p); }
        g_fscr.p = nullptr; g_fscr.bytes = 0; g_fscr.device = dev;
        GPX_HIP(hipMalloc(&g_fscr.p, need));
        GPX_HIP(hipMemset(g_fscr.p, 0, need));
        g_fscr.bytes = need;
        g_fscr.serial = 0;
    }
    *flag = (int *)g_fscr.p;                                       // nbatch words (256-byte slab per 64 matrices is plenty)
    *xinv = (char *)g_fscr.p + (((size_t)nbatch * sizeof(int) + 255) / 256) * 256;
    return GPX_OK;
}

// OPT-IN (GPX_POTRF_FUSED=<max panel width, e.g. 256>).  Measured on one MI355X (n = 8192, nb = 256): a step
// takes 50 / 55 / 68 / 74 us alone (kin = 0 .. 192: workgroup 0's D' update reads its operands with
// un-pipelined global loads, 8 - 12 us per 64 columns) and 95 - 130 us beside the trailing update, against
// ~50 / ~75 us for the leaf + substitution + update launches it replaces: fit 10.55 -> 11.3 ms, the 64-restart
// lock-step sweep 0.293 -> 0.322 s.  Off by default until the operand loads are pipelined and the leaf itself
// (26 us of every step) is faster.
static int64_t fused_max()
{
    const int64_t v = getenv("GPX_POTRF_FUSED") ? atoll(getenv("GPX_POTRF_FUSED")) : 0;
    return v;
}

template <typename T>
static int potrf_panel_fused(T *A, int64_t lda, int64_t n, int64_t r0, int64_t c0, int64_t kb, int *info_dev,
                             hipStream_t st, const Batch *bt)
{
    const int nbatch = bt ? bt->count : 1;
    void *xinv = nullptr; int *flag = nullptr;
    GPX_TRY(fused_scratch(nbatch, sizeof(T), &xinv, &flag));
    for (int64_t s = 0; s < kb; s += IB) {
        const int64_t below = n - (r0 + s + IB);
        dim3 grid((unsigned)(1 + cdiv(std::max<int64_t>(below, 0), IB)), (unsigned)nbatch);
        ProfScope prof(PC_POTRF_DIAG, ((double)IB * IB * IB / 3.0 + (double)std::max<int64_t>(below, 0) * IB * (2.0 * s + IB)) * nbatch, st);
        hipLaunchKernelGGL((panel_step_kernel<T>), grid, dim3(256), 0, st, A, lda, n, r0 + s, c0 + s, (int)s, info_dev,
                           (T *)xinv, flag, ++g_fscr.serial, bt ? bt->sA : (int64_t)0);
    }
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}

// ---- (c) X[r, 0:jb] <- X[r, 0:jb] * Ljj^-T : one lane per row --------------
// Forward substitution along the row: x_c = (a_c - sum_{t<c} x_t L[c,t]) / L[c,c].
template <typename T, bool FULL64>
__global__ __launch_bounds__(256) void trsm_rows_kernel(T *__restrict__ X, int64_t ldx, int64_t rows,
                                                        const T *__restrict__ Ljj, int64_t ldl, int jb,
                                                        int64_t sX, int64_t sLjj)
{
    X += (int64_t)blockIdx.y * sX;
    Ljj += (int64_t)blockIdx.y * sLjj;
    __shared__ T sL[IB * IBP];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < jb * jb; idx += 256) {
        const int i = idx / jb, c = idx - i * jb;
        sL[i * IBP + c] = (c <= i) ? Ljj[(int64_t)i * ldl + c] : (T)0;
    }
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * 256 + tid;
    if (r >= rows) return;
    T *xr = X + r * ldx;
    if (FULL64) {
        T x[IB];
        constexpr int CH = 16 / sizeof(T);
#pragma unroll
        for (int c = 0; c < IB; c += CH) {
            struct alignas(16) V { T e[CH]; } v = *reinterpret_cast<const V *>(xr + c);
#pragma unroll
            for (int e = 0; e < CH; ++e) x[c + e] = v.e[e];
        }
#pragma unroll
        for (int c = 0; c < IB; ++c) {
            T v = x[c];
#pragma unroll
            for (int t = 0; t < c; ++t) v = fma(-x[t], sL[c * IBP + t], v);
            x[c] = v / sL[c * IBP + c];
        }
#pragma unroll
        for (int c = 0; c < IB; c += CH) {
            struct alignas(16) V { T e[CH]; } v;
#pragma unroll
            for (int e = 0; e < CH; ++e) v.e[e] = x[c + e];
            *reinterpret_cast<V *>(xr + c) = v;
        }
    } else {
        for (int c = 0; c < jb; ++c) {
            T v = xr[c];
            for (int t = 0; t < c; ++t) v = fma(-xr[t], sL[c * IBP + t], v);
            xr[c] = v / sL[c * IBP + c];
        }
    }
}

template <typename T>
static int launch_trsm_rows(void *X, int64_t ldx, int64_t rows, const void *Ljj, int64_t ldl, int jb,
                            hipStream_t st, const Batch *bt)
{
    if (rows <= 0 || jb <= 0) return GPX_OK;
    dim3 grid((unsigned)cdiv(rows, 256), (unsigned)(bt ? bt->count : 1)), block(256);
    const int64_t sX = bt ? bt->sA : 0, sL = bt ? bt->sB : 0;
    const bool vec_ok = (jb == IB) && (ldx % (16 / (int64_t)sizeof(T)) == 0) && (((uintptr_t)X) % 16 == 0);
    ProfScope prof(PC_TRSM_ROWS, (double)rows * jb * jb * grid.y, st);
    if (vec_ok)
        hipLaunchKernelGGL((trsm_rows_kernel<T, true>), grid, block, 0, st, (T *)X, ldx, rows,
                           (const T *)Ljj, ldl, jb, sX, sL);
    else
        hipLaunchKernelGGL((trsm_rows_kernel<T, false>), grid, block, 0, st, (T *)X, ldx, rows,
                           (const T *)Ljj, ldl, jb, sX, sL);
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}

int trsm_rows(int dtype, void *X, int64_t ldx, int64_t rows, const void *Ljj, int64_t ldl, int jb,
              hipStream_t st, const Batch *bt)
{
    if (dtype == GPX_F64) return launch_trsm_rows<double>(X, ldx, rows, Ljj, ldl, jb, st, bt);
    return launch_trsm_rows<float>(X, ldx, rows, Ljj, ldl, jb, st, bt);
}

template <typename T>
__global__ void tril_kernel(T *__restrict__ A, int64_t n, int64_t lda)
{
    for (int64_t i = blockIdx.y; i < n; i += gridDim.y)
        for (int64_t c = i + 1 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n;
             c += (int64_t)gridDim.x * blockDim.x)
            A[i * lda + c] = (T)0;
}

int tril(int dtype, void *A, int64_t n, int64_t lda, hipStream_t st)
{
    if (n <= 1) return GPX_OK;
    dim3 grid((unsigned)std::min<int64_t>(cdiv(n, 256), 64), (unsigned)std::min<int64_t>(n, 32768)), block(256);
    if (dtype == GPX_F64) hipLaunchKernelGGL((tril_kernel<double>), grid, block, 0, st, (double *)A, n, lda);
    else hipLaunchKernelGGL((tril_kernel<float>), grid, block, 0, st, (float *)A, n, lda);
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}

static int64_t outer_block(int64_t n, bool batched = false)
{
    const char *env = getenv("GPX_POTRF_NB");
    if (env) {
        int64_t v = atoll(env);
        if (v >= IB && v % IB == 0) return v;
    }
    // lock-step batches: the chain is shared by all matrices, so the deeper K = 512 tiles of the trailing update
    // win (64 x n = 8192: 0.238 -> 0.214 s; 8 x: 0.290 -> 0.272 s)
    // (with the one-launch panels: 1024 for n >= 8192 -- 64 x n = 8192: 0.217 -> 0.213 s, 16 x: 0.253 -> 0.240 s)
    if (batched && n >= 8192 && n <= 32768) return 1024;
    if (batched && n >= 4096 && n <= 32768) return 512;
    // measured per factorisation (v8): n = 8192: 128 / 256 / 512 -> 12.6 / 11.5 / 12.5 ms; n = 16384:
    // 256 / 512 -> 42.4 / 40.8; n = 24576: 256 / 512 / 1024 -> 106 / 98 / 102; n = 32768: 228 / 203 / 204;
    // n = 65536: 512 / 1024 -> 1445 / 1416 ms
    // with the resident panel kernel (gpx_panel.hip): n = 8192: 256 / 512 -> 8.77 / 8.89 ms; n = 12288: 18.4 / 17.5;
    // n = 16384: 36.3 / 32.8; n = 20480: 512 / 1024 -> 56.2 / 55.8; n = 24576: 89.9 / 87.8; n = 32768: 195.7 / 191.1
    // (fp32: 105.9 / 100.5).  The width is re-evaluated per panel with the rows that are left (potrf()), thresholds
    // (rows for 128 / 256 / 512) measured with that taper: 128 never pays with the one-launch panel (n = 2048:
    // 1.45 -> 1.35 ms), 512 -> 1024 above 12288 (n = 16384: 31.6 -> 31.3 ms, n = 32768 fp32: 101.1 -> 99.9)
    static int64_t thr[3] = {1, 8192, 12288};
    static const bool thr_env = [] {
        const char *e = getenv("GPX_POTRF_WIDTHS");            // "rows128,rows256,rows512"
        if (e) { long long a, b, c; if (sscanf(e, "%lld,%lld,%lld", &a, &b, &c) == 3) { thr[0] = a; thr[1] = b; thr[2] = c; } }
        return e != nullptr;
    }();
    (void)thr_env;
    if (n <= thr[0]) return 128;
    if (n <= thr[1]) return 256;
    if (n <= thr[2]) return 512;
    return 1024;
}

// one 64 x 64 scratch block per host thread and device for the leaf's inverse (leaves of a
// panel are stream-ordered, so one block is enough)
struct LeafScratch { void *p = nullptr; size_t bytes = 0; int device = -1; };
static thread_local LeafScratch g_leaf;
static int leaf_scratch(size_t bytes, void **out)
{
    int dev = 0;
    GPX_HIP(hipGetDevice(&dev));
    if (g_leaf.device != dev || g_leaf.bytes < bytes) {
        // growing on the same device: the old block may still be in use by queued leaves (a previous device's
        // block is left to that context)
        if (g_leaf.p && g_leaf.device == dev) { GPX_HIP(hipDeviceSynchronize()); (void)hipFree(g_leaf.p); }
        g_leaf.p = nullptr;
        GPX_HIP(hipMalloc(&g_leaf.p, std::max<size_t>(bytes, IB * IB * 8)));
        g_leaf.bytes = std::max<size_t>(bytes, IB * IB * 8);
        g_leaf.device = dev;
    }
    *out = g_leaf.p;
    return GPX_OK;
}

// ---- panel: rows [r0, n) x columns [c0, c0 + kb), diagonal block at (r0, c0) ----
// Recursive halving down to 64 columns: factor the left half, apply it to the right
// half with ONE MFMA GEMM (N = K = half the width: 57 % of a 512-wide panel's flops
// run as a 256 x 256-deep product, 29 % as 128 x 128, 14 % as 64 x 64), factor the
// right half.  A 64-wide leaf is the diagonal block (one workgroup) followed by the
// row substitution below it.
template <typename T>
static int potrf_panel_t(T *A, int64_t lda, int64_t n, int64_t r0, int64_t c0, int64_t kb, int *info_dev,
                         hipStream_t st, int dtype, const Batch *bt, T *inv_slots = nullptr, int64_t pc0 = 0, int64_t kpre = 0);

template <typename T>
__global__ void place_inv_kernel(const T *__restrict__ slots, T *__restrict__ W, T *__restrict__ Wt, int64_t ld)
{
    // block b: W[b, b] = X_b (the leaf's 64 x 64 inverse, row-major), Wt[b, b] = X_b^T
    const T *X = slots + (int64_t)blockIdx.x * (IB * IB);
    const int64_t o = (int64_t)blockIdx.x * IB * (ld + 1);
    for (int idx = threadIdx.x; idx < IB * IB; idx += blockDim.x) {
        const int r = idx / IB, c = idx - r * IB;
        const T v = X[idx];
        W[o + (int64_t)r * ld + c] = v;
        Wt[o + (int64_t)c * ld + r] = v;
    }
}

template <typename T>
__global__ void copy2d_kernel(const T *__restrict__ src, int64_t lds, T *__restrict__ dst, int64_t ldd, int64_t rows,
                              int cols)
{
    constexpr int V = 16 / sizeof(T);
    const int chunks = cols / V;                          // cols % 16 == 0 here
    const int64_t total = rows * chunks;
    struct alignas(16) Q { T e[V]; };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / chunks;
        const int c = (int)(i - r * chunks) * V;
        *reinterpret_cast<Q *>(dst + r * ldd + c) = *reinterpret_cast<const Q *>(src + r * lds + c);
    }
}

// grow-only scratch of the tall-panel route (one per host thread and device)
struct PanelScratch { void *p = nullptr; size_t bytes = 0; int device = -1; };
static thread_local PanelScratch g_pscr;
static int panel_scratch(size_t bytes, void **out)
{
    int dev = 0;
    GPX_HIP(hipGetDevice(&dev));
    if (g_pscr.device != dev || g_pscr.bytes < bytes) {
        if (g_pscr.p && g_pscr.device == dev) { GPX_HIP(hipDeviceSynchronize()); (void)hipFree(g_pscr.p); }
        g_pscr.p = nullptr; g_pscr.bytes = 0; g_pscr.device = dev;
        GPX_HIP(hipMalloc(&g_pscr.p, bytes));
        g_pscr.bytes = bytes;
    }
    *out = g_pscr.p;
    return GPX_OK;
}

// Tall panel (rows below the diagonal block >> its width): "diagonal block first".
//   1. factor ONLY the kb x kb diagonal block (the recursion below sees no further rows); every 64 x 64 leaf
//      leaves its inverse in a slot;
//   2. W = inv(L11) (kb x kb, lower) by recursive doubling from the leaf inverses: for block size s -> 2 s
//      W21 = -W22 L21 W11, three s x s x s products per pair, all pairs of a level in one batched launch
//      (W and its transpose are both kept: the NT product needs either operand row-major);
//   3. ALL rows below in ONE product  T = A21 W^T  (K = kb deep on the MFMA kernel, the k-loop of a tile
//      column stops at the triangle's edge), copied back over A21.
// The rows below are then touched by 1 efficient launch instead of 2 (kb / 64) thin ones (a one-lane-per-row
// substitution or a K = 64 product per leaf, plus the K = 64 ... kb / 2 updates in between).
template <typename T>
static int potrf_panel_tall(T *A, int64_t lda, int64_t n, int64_t r0, int64_t c0, int64_t kb, int *info_dev,
                            hipStream_t st, int dtype)
{
    const int64_t below = n - (r0 + kb);
    const int nleaf = (int)(kb / IB);
    const size_t es = sizeof(T);
    // scratch: leaf inverses | W | Wt | Pt (kb x kb each, ld = kb) | T (below x kb)
    const size_t o_w = (size_t)nleaf * IB * IB, o_wt = o_w + (size_t)kb * kb, o_pt = o_wt + (size_t)kb * kb,
                 o_t = o_pt + (size_t)kb * kb, total = o_t + (size_t)below * kb;
    void *scr = nullptr;
    GPX_TRY(panel_scratch(total * es, &scr));
    T *slots = (T *)scr, *W = slots + o_w, *Wt = slots + o_wt, *Pt = slots + o_pt, *Tm = slots + o_t;
    // 1. the diagonal block
    GPX_TRY(potrf_panel_t<T>(A, lda, r0 + kb, r0, c0, kb, info_dev, st, dtype, nullptr, slots, c0));
    // 2. W = inv(L11)
    GPX_HIP(hipMemsetAsync(W, 0, 2 * (size_t)kb * kb * es, st));           // W and Wt
    hipLaunchKernelGGL((place_inv_kernel<T>), dim3(nleaf), dim3(256), 0, st, slots, W, Wt, kb);
    GPX_LAUNCH_CHECK();
    const T *L11 = A + r0 * lda + c0;
    for (int64_t s = IB; s < kb; s *= 2) {
        Batch b;
        b.count = (int)(kb / (2 * s));
        const int64_t dW = 2 * s * kb + 2 * s, dL = 2 * s * lda + 2 * s;
        // Pt = W11^T L21^T  (s x s):  gemm_nt(A = Wt11, B = L21)
        b.sA = dW; b.sB = dL; b.sC = dW;
        GPX_TRY(gemm_nt(dtype, s, s, s, Wt, kb, L11 + s * lda, lda, Pt, kb, 1.0, GPX_FULL, 0, 0, st, 1, 0, &b));
        // W21 = -W22 (L21 W11) = -W22 Pt^T:  gemm_nt(A = W22, B = Pt)
        b.sA = dW; b.sB = dW; b.sC = dW;
        GPX_TRY(gemm_nt(dtype, s, s, s, W + s * kb + s, kb, Pt, kb, W + s * kb, kb, -1.0, GPX_FULL, 0, 0, st, 1, 0, &b));
        // (W^T)12 = W21^T = -Pt W22^T:  gemm_nt(A = Pt, B = W22)   -- not needed after the last level
        if (2 * s < kb)
            GPX_TRY(gemm_nt(dtype, s, s, s, Pt, kb, W + s * kb + s, kb, Wt + s, kb, -1.0, GPX_FULL, 0, 0, st, 1, 0, &b));
    }
    // 3. T = A21 W^T, then back over A21
    const T *A21 = A + (r0 + kb) * lda + c0;
    GPX_TRY(gemm_nt(dtype, below, kb, kb, A21, lda, W, kb, Tm, kb, 1.0, GPX_FULL, 0, 0, st, 1, 2));
    const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(below * (kb / (16 / (int64_t)es)), 256), 8192);
    hipLaunchKernelGGL((copy2d_kernel<T>), dim3(blocks), dim3(256), 0, st, Tm, kb, A + (r0 + kb) * lda + c0, lda, below,
                       (int)kb);
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}

static bool tall_route(int64_t below, int64_t kb, int64_t lda, size_t es, const void *base, const Batch *bt)
{
    // OPT-IN (GPX_POTRF_TALL=<min rows below>): measured on one MI355X it does not pay -- the doubling adds
    // 5 - 8 dependent small launches to the panel chain (n = 8192: fit 10.6 -> 13.8 ms; n = 16384: 37.9 -> 45.3;
    // n = 32768 fp32: 108 -> 116) and at n = 65536, where the chain is hidden anyway, the fit is unchanged
    // (1.383 vs 1.389 s).  Kept because it is the shape a fused inverse kernel would slot into.
    const int64_t env = getenv("GPX_POTRF_TALL") ? atoll(getenv("GPX_POTRF_TALL")) : 0;
    if (!env || bt) return false;
    if (kb < 2 * IB || (kb & (kb - 1)) != 0) return false;                 // 128, 256, 512, 1024
    if (below < 2 * kb || below < env) return false;
    return lda % (16 / (int64_t)es) == 0 && ((uintptr_t)base) % 16 == 0;
}

// lean panel route (kb <= 256, a multiple of 64): per 64 columns the leaf, then one row kernel
template <typename T>
static int potrf_panel_lean(T *A, int64_t lda, int64_t n, int64_t r0, int64_t c0, int64_t kb, int *info_dev,
                            hipStream_t st, const Batch *bt)
{
    const int nbatch = bt ? bt->count : 1;
    const int64_t sM = bt ? bt->sA : 0;
    void *p = nullptr;
    GPX_TRY(leaf_scratch((size_t)nbatch * (1 + LEAN_TMAX) * IB * IB * sizeof(T), &p));
    T *inv = (T *)p, *rd = inv + (size_t)nbatch * IB * IB;
    static const int pipe_env = getenv("GPX_LEAF_PIPE") ? atoi(getenv("GPX_LEAF_PIPE")) : -1;
    const bool pipe = pipe_env < 0 ? g_leaf_pipe : pipe_env != 0;
    for (int64_t s = 0; s < kb; s += IB) {
        const int tb = (int)((kb - s - IB) / IB);                   // 64-column blocks to the right, inside the panel
        T *D = A + (r0 + s) * lda + c0 + s;
        {
            ProfScope prof(PC_POTRF_DIAG, (double)IB * IB * IB / 3.0 * nbatch, st);
            if (pipe)
                hipLaunchKernelGGL((potrf_diag_pipe_kernel<T, true>), dim3(nbatch), dim3(320), 0, st, D, lda, r0 + s, IB,
                                   info_dev, inv, sM, IB / 4, (unsigned long long *)nullptr, rd, tb * IB);
            else
                hipLaunchKernelGGL((potrf_diag_kernel<T, true>), dim3(nbatch), dim3(256), 0, st, D, lda, r0 + s, IB, info_dev,
                                   inv, sM, rd, tb * IB);
        }
        const int64_t rb = r0 + s + IB, below = n - rb;
        if (below > 0) {
            ProfScope prof(PC_TRSM_ROWS, (double)below * IB * IB * (1.0 + 2.0 * tb) * nbatch, st);
            hipLaunchKernelGGL((panel_rows_kernel<T>), dim3((unsigned)cdiv(below, IB), (unsigned)nbatch), dim3(256), 0, st, A, lda, n,
                               rb, c0 + s, tb, (const T *)inv, (const T *)rd, sM);
        }
    }
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}

template <typename T>
static int potrf_panel_t(T *A, int64_t lda, int64_t n, int64_t r0, int64_t c0, int64_t kb, int *info_dev,
                         hipStream_t st, int dtype, const Batch *bt, T *inv_slots, int64_t pc0, int64_t kpre)
{
    const int nbatch = bt ? bt->count : 1;
    const int64_t sM = bt ? bt->sA : 0;              // stride between the matrices of a batch
    // (c0 is a LOCAL column for a rank of the multi-GPU schedule: r0 != c0 there)
    if (!inv_slots && kb % IB == 0 && kb <= panel_res_max())
        return potrf_panel_res(dtype, A, lda, n, r0, c0, kb, info_dev, st, bt, kpre);
    if (kpre != 0) { set_error("potrf_panel: a folded update needs the resident panel route"); return GPX_ERR_ARG; }
    if (!inv_slots && tall_route(n - (r0 + kb), kb, lda, sizeof(T), A + r0 * lda + c0, bt))
        return potrf_panel_tall<T>(A, lda, n, r0, c0, kb, info_dev, st, dtype);
    if (!inv_slots && kb % IB == 0 && kb <= fused_max() && lda % (16 / (int64_t)sizeof(T)) == 0 &&
        ((uintptr_t)(A + r0 * lda + c0)) % 16 == 0 && c0 % (16 / (int64_t)sizeof(T)) == 0)
        return potrf_panel_fused<T>(A, lda, n, r0, c0, kb, info_dev, st, bt);
    if (!inv_slots && kb % IB == 0 && kb <= lean_max() && n - r0 <= lean_rows_max() && lda % (16 / (int64_t)sizeof(T)) == 0 &&
        ((uintptr_t)(A + r0 * lda + c0)) % 16 == 0 && c0 % (16 / (int64_t)sizeof(T)) == 0)
        return potrf_panel_lean<T>(A, lda, n, r0, c0, kb, info_dev, st, bt);
    if (kb <= IB) {
        const int jb = (int)kb;
        T *D = A + r0 * lda + c0;
        const int64_t below = n - (r0 + jb);
        static const bool force_rows = getenv("GPX_POTRF_TRSM_ROWS") != nullptr;
        // measured: the inverse + MFMA route wins for short panels (N = 8192: -2.7 % per fit) and
        // loses for tall ones (N = 65536: +2 %, its 120 KiB tiles displace trailing-update workgroups)
        static const int64_t inv_max = getenv("GPX_POTRF_INV_MAX") ? atoll(getenv("GPX_POTRF_INV_MAX")) : 16384;
        const bool via_inverse = !force_rows && below > 0 && below <= inv_max && jb == IB &&
                                 lda % (16 / (int64_t)sizeof(T)) == 0 &&
                                 ((uintptr_t)(A + (r0 + jb) * lda + c0)) % 16 == 0;
        T *inv = nullptr;
        if (inv_slots) {
            inv = inv_slots + ((c0 - pc0) / IB) * (IB * IB);        // tall-panel route: every leaf keeps its inverse
        } else if (via_inverse) {
            void *p = nullptr;
            GPX_TRY(leaf_scratch((size_t)nbatch * IB * IB * sizeof(T), &p));
            inv = (T *)p;
        }
        {
            ProfScope prof(PC_POTRF_DIAG, (double)jb * jb * jb / 3.0 * nbatch, st);
            // the 320-thread pivot-wave leaf wins where CUs are set aside for the panel stream (n <= 12288: 30.4 -> 27 us)
            // and loses where it has to wait for a slot among the trailing update's workgroups (fp64 n = 16384:
            // potrf 37.2 -> 40.2 ms, n = 24576: 94.4 -> 98.1): GPX_LEAF_PIPE = 1 always, 0 never, default by reservation
            static const int pipe_env = getenv("GPX_LEAF_PIPE") ? atoi(getenv("GPX_LEAF_PIPE")) : -1;
            const bool pipe = pipe_env < 0 ? g_leaf_pipe : pipe_env != 0;
            static const int nsteps = getenv("GPX_LEAF_ABLATE") ? atoi(getenv("GPX_LEAF_ABLATE")) : IB / 4;   // timing only
            if (pipe) {
                if (inv)
                    hipLaunchKernelGGL((potrf_diag_pipe_kernel<T, true>), dim3(nbatch), dim3(320), 0, st, D, lda, r0, jb,
                                       info_dev, inv, sM, nsteps, g_leaf_stamps);
                else
                    hipLaunchKernelGGL((potrf_diag_pipe_kernel<T, false>), dim3(nbatch), dim3(320), 0, st, D, lda, r0, jb,
                                       info_dev, inv, sM, nsteps, g_leaf_stamps);
            } else if (inv)
                hipLaunchKernelGGL((potrf_diag_kernel<T, true>), dim3(nbatch), dim3(256), 0, st, D, lda, r0, jb,
                                   info_dev, inv, sM);
            else
                hipLaunchKernelGGL((potrf_diag_kernel<T, false>), dim3(nbatch), dim3(256), 0, st, D, lda, r0, jb,
                                   info_dev, inv, sM);
        }
        GPX_LAUNCH_CHECK();
        if (inv_slots ? below > 0 : via_inverse) {
            // rows below: X <- X * inv(L_jj)^T, in place (each tile reads all 64 columns of its own
            // rows before its epilogue stores them; no other tile touches those rows)
            T *Xb = A + (r0 + jb) * lda + c0;
            Batch bi; bi.count = nbatch; bi.sA = sM; bi.sB = IB * IB; bi.sC = sM;
            GPX_TRY(gemm_nt(dtype, below, jb, jb, Xb, lda, inv, IB, Xb, lda, 1.0, GPX_FULL, 0, 0, st, 1, 0,
                            bt ? &bi : nullptr));
        } else if (below > 0) {
            Batch br; br.count = nbatch; br.sA = sM; br.sB = sM; br.sC = 0;
            GPX_TRY(trsm_rows(dtype, A + (r0 + jb) * lda + c0, lda, below, D, lda, jb, st, bt ? &br : nullptr));
        }
        return GPX_OK;
    }
    const int64_t h = ((kb / IB + 1) / 2) * IB;          // left half, a multiple of 64
    GPX_TRY(potrf_panel_t<T>(A, lda, n, r0, c0, h, info_dev, st, dtype, bt, inv_slots, pc0));
    T *R = A + (r0 + h) * lda + c0;                       // rows below the left half's diagonal block
    // the right half takes the left half's update itself when it is one resident-kernel launch over few rows
    if (!inv_slots && panel_res_fold(n - (r0 + h), h, kb - h, sizeof(T), lda, A))
        return potrf_panel_t<T>(A, lda, n, r0 + h, c0 + h, kb - h, info_dev, st, dtype, bt, inv_slots, pc0, h);
    GPX_TRY(gemm_nt(dtype, n - (r0 + h), kb - h, h, R, lda, R, lda, R + h, lda, -1.0, GPX_LOWER, 0, 0, st, 0, 0, bt));
    return potrf_panel_t<T>(A, lda, n, r0 + h, c0 + h, kb - h, info_dev, st, dtype, bt, inv_slots, pc0);
}

int potrf_panel(int dtype, void *A, int64_t lda, int64_t n, int64_t r0, int64_t c0, int64_t kb,
                int *info_dev, hipStream_t st, const Batch *bt, int64_t kpre)
{
    if (dtype == GPX_F64)
        return potrf_panel_t<double>((double *)A, lda, n, r0, c0, kb, info_dev, st, dtype, bt, nullptr, 0, kpre);
    return potrf_panel_t<float>((float *)A, lda, n, r0, c0, kb, info_dev, st, dtype, bt, nullptr, 0, kpre);
}

// side stream + event pool for the look-ahead (one set per host thread and device)
struct LookAhead {
    int device = -1;
    hipStream_t q = nullptr;
    // trailing-update streams that leave `reserved` CUs to the panel stream, one per reservation ever asked for.
    // They live until the owning thread ends: pooled events keep referring to the stream they were last recorded
    // on, and a factorisation that alternates between reservations (single fits: 32, small lock-step batches: 16)
    // must not destroy a stream under them (seen as an intermittent hang of the next fit).  At thread end they are
    // given back -- a CU-masked stream that is still alive at process teardown crashes rocprofv3's finaliser.
    static constexpr int MAXT = 8;
    hipStream_t t[MAXT] = {};
    int reserved[MAXT] = {};
    int nt = 0;
    void drop_streams()
    {
        for (int i = 0; i < nt; ++i)
            if (t[i]) { (void)hipStreamSynchronize(t[i]); (void)hipStreamDestroy(t[i]); t[i] = nullptr; }
        nt = 0;
    }
    ~LookAhead() { drop_streams(); }
    std::vector<hipEvent_t> ev;
    size_t next = 0;
    int get(hipEvent_t *e)
    {
        if (next == ev.size()) {
            hipEvent_t x;
            GPX_HIP(hipEventCreateWithFlags(&x, hipEventDisableTiming));
            ev.push_back(x);
        }
        *e = ev[next++];
        return GPX_OK;
    }
};
static thread_local LookAhead g_la;

static int lookahead_setup()
{
    int dev = 0;
    GPX_HIP(hipGetDevice(&dev));
    if (g_la.device != dev) {
        g_la.drop_streams();
        g_la.q = nullptr; g_la.ev.clear();
        g_la.device = dev;
        // the panel is on the critical path of the NEXT step: give its stream the highest
        // priority so that its workgroups get the CUs that trailing-update workgroups free up
        int least = 0, greatest = 0;
        GPX_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        GPX_HIP(hipStreamCreateWithPriority(&g_la.q, hipStreamNonBlocking, greatest));
    }
    g_la.next = 0;
    return GPX_OK;
}

// Small matrices: the panel chain (a dozen dependent, mostly tiny kernels) is as long as the
// trailing update it should hide under, and every one of its kernels would wait for a
// trailing-update workgroup to retire before it finds room on a CU (measured at n = 8192:
// the 64 x 64 leaf 33 us alone, 105 us beside the update).  The update therefore runs on
// a stream whose CU mask leaves `reserve` CUs (mask bits interleave over the 8 XCDs) free;
// the panel stream keeps the whole chip.
static int trailing_stream(int reserve, hipStream_t *out)
{
    for (int i = 0; i < g_la.nt; ++i)
        if (g_la.reserved[i] == reserve) { *out = g_la.t[i]; return GPX_OK; }
    *out = nullptr;
    if (g_la.nt == LookAhead::MAXT) return GPX_OK;              // (more distinct reservations than anyone asks for: share the chip)
    hipDeviceProp_t prop;
    GPX_HIP(hipGetDeviceProperties(&prop, g_la.device));
    const int ncu = prop.multiProcessorCount;
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int i = reserve; i < ncu; ++i) mask[i / 32] |= 1u << (i % 32);
    hipStream_t t = nullptr;
    if (hipExtStreamCreateWithCUMask(&t, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
        (void)hipGetLastError();                                // masks unsupported here: share the chip as for large n
        return GPX_OK;
    }
    g_la.t[g_la.nt] = t; g_la.reserved[g_la.nt] = reserve; ++g_la.nt;
    *out = t;
    return GPX_OK;
}

static int reserve_cus(int64_t n)
{
    const char *env = getenv("GPX_POTRF_RESERVE_CUS");
    if (env) return std::max(0, std::min(128, atoi(env)));
    return n <= 12288 ? 32 : 0;      // measured: n = 8192 13.3 -> 12.3 ms per factorisation, n = 16384 41.9 -> 42.5
}

// Lock-step batches: without reserved CUs the panel stream's first kernel of every step (a handful of leaf
// workgroups) is not dispatched until the trailing update of all matrices has drained -- measured with 8
// matrices of n = 8192: that leaf "takes" 950 - 1030 us, the length of the update it should run under -- so
// the look-ahead overlaps nothing and every step is update + chain.
// With many matrices the trailing update is so long that the exposed chain no longer matters and the masked
// CUs cost more (64 matrices: 0.214 s without, 0.225 s with 16 reserved; 8 matrices: 0.290 / 0.272 s).
static int reserve_cus_batch(int64_t n, int count)
{
    const char *env = getenv("GPX_POTRF_RESERVE_CUS_BATCH");
    if (env) return std::max(0, std::min(128, atoi(env)));
    return (n <= 12288 && count <= 16) ? 16 : 0;
}

// Right-looking blocked Cholesky with one-panel look-ahead: while the main stream
// applies panel k to the block columns beyond k + 1, the side stream already
// factors panel k + 1 (whose block column was updated first).
// bt != null: bt->count matrices sA elements apart are factored in lock-step (every launch covers all of
// them; info_dev then holds one word per matrix).
int potrf(int dtype, void *A, int64_t n, int64_t lda, int *info_dev, hipStream_t st, const Batch *bt, int64_t xrows)
{
    // xrows extra rows below the n x n matrix take part in every panel and update as rows, never as columns: on
    // return row n + i holds L^-1 applied to what was stored there (a right-hand side rides along: gpx_gp_fit)
    const int64_t N = n + xrows;
    GPX_HIP(hipMemsetAsync(info_dev, 0, sizeof(int) * (bt ? bt->count : 1), st));
    const int64_t nb = outer_block(n, bt != nullptr);
    const int64_t nblk = cdiv(n, nb);
    const size_t es = esize(dtype);
    static const bool no_la = getenv("GPX_POTRF_NO_LOOKAHEAD") != nullptr;
    g_leaf_pipe = false;
    if (nblk <= 1) return potrf_panel(dtype, A, lda, N, 0, 0, n, info_dev, st, bt);
    auto at = [&](int64_t r, int64_t c) { return (char *)A + (r * lda + c) * es; };
    if (no_la) {
        for (int64_t k0 = 0; k0 < n; k0 += nb) {
            const int64_t kb = std::min(nb, n - k0), r = k0 + kb;
            GPX_TRY(potrf_panel(dtype, A, lda, N, k0, k0, kb, info_dev, st, bt));
            if (r < n) GPX_TRY(syrk_bc(dtype, N, r, A, lda, r, n, at(k0, k0), lda, k0, kb, nb, 1, 0, st, info_dev, bt));
        }
        return GPX_OK;
    }
    GPX_TRY(lookahead_setup());
    hipStream_t q = g_la.q;
    hipEvent_t e, ep;
    GPX_TRY(g_la.get(&e));
    GPX_HIP(hipEventRecord(e, st));
    GPX_HIP(hipStreamWaitEvent(q, e, 0));
    hipStream_t user = st;
    const int reserve = bt ? reserve_cus_batch(n, bt->count) : reserve_cus(n);
    hipStream_t masked = nullptr;
    if (reserve > 0) {
        GPX_TRY(trailing_stream(reserve, &masked));             // the updates go to the masked stream ...
        if (masked) g_leaf_pipe = true;
    }
    // ... once the step is bound by the panel chain: while the trailing update is the longer of the two (many
    // rows left) it keeps the whole chip
    static const int64_t reserve_below = getenv("GPX_POTRF_RESERVE_BELOW") ? atoll(getenv("GPX_POTRF_RESERVE_BELOW")) : 8192;
    auto switch_to = [&](hipStream_t want) -> int {
        if (want == st) return GPX_OK;
        hipEvent_t es;
        GPX_TRY(g_la.get(&es));
        GPX_HIP(hipEventRecord(es, st));
        GPX_HIP(hipStreamWaitEvent(want, es, 0));
        st = want;
        return GPX_OK;
    };
    if (masked && n <= reserve_below) GPX_TRY(switch_to(masked));
    // Tapered outer block: the width that suits a factorisation of the rows that are LEFT (wide while the trailing
    // update is the longer of the two, narrow once the step is bound by the panel chain); widths only ever shrink
    // and each divides the one before, so every panel stays aligned to its own width.  Lock-step batches and a
    // forced GPX_POTRF_NB keep one width.
    static const bool taper = !(getenv("GPX_POTRF_TAPER") && atoi(getenv("GPX_POTRF_TAPER")) == 0) && !getenv("GPX_POTRF_NB");
    auto nominal = [&](int64_t k0) -> int64_t { return (taper && !bt) ? std::min(nb, outer_block(n - k0)) : nb; };
    int64_t k0 = 0, kb = std::min(nominal(0), n);
    GPX_TRY(potrf_panel(dtype, A, lda, N, 0, 0, kb, info_dev, q, bt));
    GPX_TRY(g_la.get(&ep));
    GPX_HIP(hipEventRecord(ep, q));
    hipEvent_t e_rest = nullptr;                                // fires when the trailing update of the step before is done
    while (true) {
        const int64_t r = k0 + kb;
        if (masked && n - r <= reserve_below) GPX_TRY(switch_to(masked));
        GPX_HIP(hipStreamWaitEvent(st, ep, 0));                 // panel k is factored
        if (r >= n) break;
        const int64_t w1 = nominal(r), kb1 = std::min(w1, n - r);
        // block column k + 1 first, so that its panel can start ... (small n: the panel kernel applies panel k to its
        // own columns itself -- it then only waits for the trailing update of step k - 1, not for this stream's turn)
        const bool fold = panel_res_fold(N - r, kb, kb1, es, lda, A);
        if (!fold) {
            GPX_TRY(syrk_bc(dtype, N, r, A, lda, r, r + kb1, at(k0, k0), lda, k0, kb, w1, 1, 0, st, info_dev, bt));
            GPX_TRY(g_la.get(&e));
            GPX_HIP(hipEventRecord(e, st));
            GPX_HIP(hipStreamWaitEvent(q, e, 0));
        } else if (e_rest) {
            GPX_HIP(hipStreamWaitEvent(q, e_rest, 0));
        }
        GPX_TRY(potrf_panel(dtype, A, lda, N, r, r, kb1, info_dev, q, bt, fold ? kb : 0));
        GPX_TRY(g_la.get(&ep));
        GPX_HIP(hipEventRecord(ep, q));
        // ... while the rest of the trailing matrix is updated underneath it
        if (r + kb1 < n)
            GPX_TRY(syrk_bc(dtype, N, r, A, lda, r + kb1, n, at(k0, k0), lda, k0, kb, w1, 1, 0, st, info_dev, bt));
        GPX_TRY(g_la.get(&e_rest));
        GPX_HIP(hipEventRecord(e_rest, st));
        k0 = r; kb = kb1;
    }
    if (st != user) {
        GPX_TRY(g_la.get(&e));
        GPX_HIP(hipEventRecord(e, st));
        GPX_HIP(hipStreamWaitEvent(user, e, 0));
    }
    g_leaf_pipe = false;
    return GPX_OK;
}

}  // namespace gpx

using namespace gpx;

extern "C" int gpx_debug_leaf_stamps(void *dev_buffer)
{
    gpx::g_leaf_stamps = (unsigned long long *)dev_buffer;
    return GPX_OK;
}

extern "C" {

int gpx_d_potrf(int dtype, void *A, int64_t n, int64_t lda, int *info_dev, void *stream)
{
    GPX_TRY(ensure_device());
    GPX_ARG(dtype == GPX_F64 || dtype == GPX_F32, "dtype must be GPX_F64 or GPX_F32");
    GPX_ARG(n >= 0, "n < 0");
    GPX_ARG(info_dev, "info_dev is NULL");
    if (n == 0) { GPX_HIP(hipMemsetAsync(info_dev, 0, sizeof(int), S(stream))); return GPX_OK; }
    GPX_ARG(A, "A is NULL");
    GPX_ARG(lda >= n, "lda < n");
    GPX_ARG(lda % 16 == 0, "lda must be a multiple of 16 elements");
    GPX_ARG(((uintptr_t)A) % 16 == 0, "A must be 16-byte aligned");
    return potrf(dtype, A, n, lda, info_dev, S(stream));
}

int gpx_d_tril(int dtype, void *A, int64_t n, int64_t lda, void *stream)
{
    GPX_TRY(ensure_device());
    GPX_ARG(dtype == GPX_F64 || dtype == GPX_F32, "dtype must be GPX_F64 or GPX_F32");
    GPX_ARG(n >= 0 && lda >= n, "bad n / lda");
    if (n == 0) return GPX_OK;
    GPX_ARG(A, "A is NULL");
    return tril(dtype, A, n, lda, S(stream));
}

}  // extern "C"

extern "C" int gpx_d_potrf_panel(int dtype, void *A, int64_t lda, int64_t n, int64_t r0, int64_t c0,
                                 int64_t kb, int *info_dev, void *stream)
{
    GPX_TRY(ensure_device());
    GPX_ARG(dtype == GPX_F64 || dtype == GPX_F32, "dtype must be GPX_F64 or GPX_F32");
    GPX_ARG(n >= 0 && r0 >= 0 && c0 >= 0 && kb >= 0 && r0 + kb <= n, "bad dimensions");
    GPX_ARG(info_dev, "info_dev is NULL");
    if (kb == 0) return GPX_OK;
    GPX_ARG(A, "A is NULL");
    GPX_ARG(lda % 16 == 0 && ((uintptr_t)A) % 16 == 0 && c0 % 16 == 0, "A / lda / c0 must be 16-element aligned");
    return potrf_panel(dtype, A, lda, n, r0, c0, kb, info_dev, S(stream));
}
