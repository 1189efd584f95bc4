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

template <typename T, bool INV>
__global__ __launch_bounds__(256) void potrf_diag_kernel(T *__restrict__ blk, int64_t lda, int64_t j0,
                                                         int jb, int *__restrict__ info, T *__restrict__ inv,
                                                         int64_t sblk)
{
    // batched launches: workgroup b factors the block of matrix b (stride sblk), own info word and inverse
    blk += (int64_t)blockIdx.x * sblk;
    info += blockIdx.x;
    if (INV) inv += (int64_t)blockIdx.x * (IB * IB);
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
    const int64_t forced = tune().potrf_nb;
    if (forced >= IB && forced % IB == 0) return forced;
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
    const long long *thr = tune().potrf_widths;                 // GPX_POTRF_WIDTHS = "rows128,rows256,rows512"; default 1, 8192, 12288
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
// Panels of up to 256 columns (a multiple of 64) are ONE launch of the resident panel kernel (gpx_panel.hip).
// Wider panels halve recursively: factor the left half, apply it to the right half with ONE MFMA GEMM (or inside
// the right half's own launch when few rows are left), factor the right half.  Ragged widths (the last block
// column of a matrix whose order is not a multiple of 64) and GPX_POTRF_RES=0 take the chain of launches: a
// 64-wide leaf is the diagonal block (one workgroup) followed by the row substitution below it.
// Three further panel routes were built, measured slower and removed again in round 3 (DESIGN section 3.2 keeps the
// measurements): "tall" (diagonal block first, inv(L11) by recursive doubling, one product for all rows below),
// "fused" (one launch per 64 columns, left-looking) and "lean" (leaf + one row kernel per 64 columns).
template <typename T>
static int potrf_panel_t(T *A, int64_t lda, int64_t n, int64_t r0, int64_t c0, int64_t kb, int *info_dev,
                         hipStream_t st, int dtype, const Batch *bt, int64_t kpre = 0, hipEvent_t done = nullptr)
{
    const int nbatch = bt ? bt->count : 1;
    const int64_t sM = bt ? bt->sA : 0;              // stride between the matrices of a batch
    // (c0 is a LOCAL column for a rank of the multi-GPU schedule: r0 != c0 there)
    if (kb % IB == 0 && kb <= panel_res_max()) {
        route_hit(RT_PANEL_RES);
        return potrf_panel_res(dtype, A, lda, n, r0, c0, kb, info_dev, st, bt, kpre, done);
    }
    if (kpre != 0) { set_error("potrf_panel: a folded update needs the resident panel route"); return GPX_ERR_ARG; }
    if (kb <= IB) {
        route_hit(RT_PANEL_CHAIN);
        const int jb = (int)kb;
        T *D = A + r0 * lda + c0;
        const int64_t below = n - (r0 + jb);
        // measured: the inverse + MFMA route wins for short panels (N = 8192: -2.7 % per fit) and
        // loses for tall ones (N = 65536: +2 %, its 120 KiB tiles displace trailing-update workgroups)
        const bool via_inverse = !tune().potrf_trsm_rows && below > 0 && below <= tune().potrf_inv_max &&
                                 jb == IB && lda % (16 / (int64_t)sizeof(T)) == 0 &&
                                 ((uintptr_t)(A + (r0 + jb) * lda + c0)) % 16 == 0;
        T *inv = nullptr;
        if (via_inverse) {
            void *p = nullptr;
            GPX_TRY(leaf_scratch((size_t)nbatch * IB * IB * sizeof(T), &p));
            inv = (T *)p;
        }
        {
            ProfScope prof(PC_POTRF_DIAG, (double)jb * jb * jb / 3.0 * nbatch, st);
            // (the 320-thread pivot-wave form of this leaf only paid with CUs set aside for the panel stream; both went in round 6)
            if (inv)
                hipLaunchKernelGGL((potrf_diag_kernel<T, true>), dim3(nbatch), dim3(256), 0, st, D, lda, r0, jb,
                                   info_dev, inv, sM);
            else
                hipLaunchKernelGGL((potrf_diag_kernel<T, false>), dim3(nbatch), dim3(256), 0, st, D, lda, r0, jb,
                                   info_dev, inv, sM);
        }
        GPX_LAUNCH_CHECK();
        if (via_inverse) {
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
        if (done) GPX_HIP(hipEventRecord(done, st));
        return GPX_OK;
    }
    // wide panels of short matrices: a blocked factorisation of their own, with its own look-ahead (round 4)
    const int64_t h = ((kb / IB + 1) / 2) * IB;          // left half, a multiple of 64
    GPX_TRY(potrf_panel_t<T>(A, lda, n, r0, c0, h, info_dev, st, dtype, bt));
    T *R = A + (r0 + h) * lda + c0;                       // rows below the left half's diagonal block
    // the right half takes the left half's update itself when it is one resident-kernel launch over few rows
    if (panel_res_fold(n - (r0 + h), h, kb - h, sizeof(T), lda, A))
        return potrf_panel_t<T>(A, lda, n, r0 + h, c0 + h, kb - h, info_dev, st, dtype, bt, h, done);
    GPX_TRY(gemm_nt(dtype, n - (r0 + h), kb - h, h, R, lda, R, lda, R + h, lda, -1.0, GPX_LOWER, 0, 0, st, 0, 0, bt));
    return potrf_panel_t<T>(A, lda, n, r0 + h, c0 + h, kb - h, info_dev, st, dtype, bt, 0, done);
}

int potrf_panel(int dtype, void *A, int64_t lda, int64_t n, int64_t r0, int64_t c0, int64_t kb,
                int *info_dev, hipStream_t st, const Batch *bt, int64_t kpre, hipEvent_t done)
{
    if (dtype == GPX_F64)
        return potrf_panel_t<double>((double *)A, lda, n, r0, c0, kb, info_dev, st, dtype, bt, kpre, done);
    return potrf_panel_t<float>((float *)A, lda, n, r0, c0, kb, info_dev, st, dtype, bt, kpre, done);
}

// side stream + event pool for the look-ahead (one set per host thread and device)
struct LookAhead {
    int device = -1;
    hipStream_t q = nullptr;
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

// progress hook of this host thread (gpx_gp_fit builds the solves' block operators while the factorisation runs)
static thread_local const PotrfHook *g_hook = nullptr;
void potrf_set_hook(const PotrfHook *hook) { g_hook = hook; }

hipStream_t potrf_side_stream()
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return g_la.device == dev ? g_la.q : nullptr;
}

static int lookahead_setup()
{
    int dev = 0;
    GPX_HIP(hipGetDevice(&dev));
    if (g_la.device != dev) {
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

// potrf()'s loop sets this right before a panel whose FIRST launch will find the chip idle (it is ordered behind the update
// before it and ahead of the one that runs beside it); the resident panel launch that takes the hint may claim whole CUs
// (gpx_panel.hip, GPX_PANEL_EXCL_ROWS).  Later launches of the same panel (the right half of a wide one) start on a chip that
// the update has filled meanwhile: they must not wait for empty CUs.
static thread_local bool g_idle_chip = false;
bool potrf_take_idle_chip_hint() { const bool v = g_idle_chip; g_idle_chip = false; return v; }
// Right-looking blocked Cholesky with one-panel look-ahead: while the main stream
// applies panel k to the block columns beyond k + 1, the side stream already
// factors panel k + 1 (whose block column was updated first).
// bt != null: bt->count matrices sA elements apart are factored in lock-step (every launch covers all of
// them; info_dev then holds one word per matrix).
int potrf(int dtype, void *A, int64_t n, int64_t lda, int *info_dev, hipStream_t st, const Batch *bt, int64_t xrows, bool may_block)
{
    // xrows extra rows below the n x n matrix take part in every panel and update as rows, never as columns: on
    // return row n + i holds L^-1 applied to what was stored there (a right-hand side rides along: gpx_gp_fit)
    const int64_t N = n + xrows;
    GPX_HIP(hipMemsetAsync(info_dev, 0, sizeof(int) * (bt ? bt->count : 1), st));
    const int64_t nb = outer_block(n, bt != nullptr);
    const int64_t nblk = cdiv(n, nb);
    const size_t es = esize(dtype);
    const bool no_la = tune().no_lookahead;
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
    // Tapered outer block: the width that suits a factorisation of the rows that are LEFT (wide while the trailing
    // update is the longer of the two, narrow once the step is bound by the panel chain); widths only ever shrink
    // and each divides the one before, so every panel stays aligned to its own width.  Lock-step batches and a
    // forced GPX_POTRF_NB keep one width.
    const bool taper = tune().taper != 0 && !tune().potrf_nb_set;
    auto nominal = [&](int64_t k0) -> int64_t { return (taper && !bt) ? std::min(nb, outer_block(n - k0)) : nb; };
    // (Round 4 also built a "pair phase" here -- while many rows are left, the far trailing matrix updated once per TWO 256-wide
    //  panels with one K = 512 product, the panels in between applying their predecessors themselves, 512 / 256 columns deep --
    //  correct, and no faster: n = 8192 5.67 - 5.97 ms against 5.58 (profiles/r04_ab_pair_phase_dropped.log).  The K = 512 update
    //  does run at 60 TF/s, but the second panel of every pair starts on a chip the update has filled and takes 280 - 470 us
    //  instead of 115 (profiles/r04_timeline_n8192_pair_phase_dropped.txt): one panel per update, dispatched ahead of it, is
    //  the schedule this machine rewards.  Removed again.)
    int64_t k0 = 0, kb = std::min(nominal(0), n);
    GPX_TRY(g_la.get(&ep));
    g_idle_chip = !bt;
    GPX_TRY(potrf_panel(dtype, A, lda, N, 0, 0, kb, info_dev, q, bt, 0, ep));
    g_idle_chip = false;
    hipEvent_t e_rest = nullptr;                                // fires when the trailing update of the step before is done
    // host pacing blocks the calling thread inside the loop: only where the caller said it may (the handle's gpx_gp_fit,
    // documented in include/gpx.h; gpx_d_potrf stays a pure enqueue) and never while the stream is being captured
    bool host_paced = may_block && !bt && n <= tune().host_paced;
    if (host_paced) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
        else if (cs != hipStreamCaptureStatusNone) host_paced = false;
    }
    // ---- PAIR phase (round 6; OPT-IN: GPX_POTRF_PAIR_ROWS = rows that must lie beyond a pair, e.g. 20480; default 0 = never).
    // While many rows are left, the far trailing matrix is updated once per TWO 1024-wide panels with ONE product of depth
    // K = 2048 -- the operand is simply both block columns of L side by side -- instead of two of depth 1024, and both panels
    // of the NEXT pair are factored on the side stream while that product runs:
    //   invariant: panels A = [k0, k0 + 1024) and B = [k0 + 1024, k0 + 2048) are factored / in flight (epA, epB); everything
    //   before A has been applied everywhere; A has been applied to B's block column only.
    //     st: U_a  block column C <- (A | B), K = 2048         q: panel C   (after U_a)
    //     st: U_b  block column D <- (A | B), K = 2048         q: V = D <- C, K = 1024 (after U_b and panel C); panel D
    //     st: U_c  columns beyond D <- (A | B), K = 2048       (the long one: both panels and V hide under it)
    // MEASURED at N = 65536 fp64 (profiles/r06_ab_pair_phase.log, r06_timeline_n65536_{pair,nopair}.txt): the step gets 0.2 ... 0.6 %
    // SHORTER on the final build (1.334 -> 1.326 - 1.331 s; 0.9 ... 1.4 % on the round's earlier build; log_lh identical to the
    // last digit) because a fit has 60 long trailing launches instead of 81 -- fewer ramps, tails and cross-stream hand-offs --
    // and the trailing updates' C traffic halves.  But the update kernel's own figure gets worse (class fraction 0.88 against
    // 0.893; per tile 0.916 of peak against 0.921): the panel stream works two panels and V back to back beside ONE launch.
    // A slightly shorter step for a lower roofline fraction of the kernel this project is measured by: not the default; the
    // route is kept, tested, for larger N.  fp32 at N = 32768 (three pairs at most): 96.5 vs 96.4 ms.
    const int64_t pair_rows = (bt || nb != 1024 || xrows > 1) ? 0 : tune().pair_rows[dtype == GPX_F64 ? 0 : 1];
    // (every panel of the phase is 1024 wide: with the taper on, the widths shrink once <= 12288 rows are left)
    auto pair_ok = [&](int64_t k) { return pair_rows > 0 && n - (k + 4 * 1024) >= pair_rows && nominal(k + 1024) == 1024 &&
                                           nominal(k + 2048) == 1024 && nominal(k + 3072) == 1024; };
    if (kb == 1024 && pair_ok(k0)) {
        // entry: B's block column <- A, then panel B (the only panel of the phase that no update hides)
        const int64_t rB = k0 + 1024;
        hipEvent_t epA = ep, epB;
        GPX_HIP(hipStreamWaitEvent(st, epA, 0));
        GPX_TRY(syrk_bc(dtype, N, rB, A, lda, rB, rB + 1024, at(k0, k0), lda, k0, 1024, 1024, 1, 0, st, info_dev, nullptr));
        GPX_TRY(g_la.get(&e));
        GPX_HIP(hipEventRecord(e, st));
        GPX_HIP(hipStreamWaitEvent(q, e, 0));
        GPX_TRY(g_la.get(&epB));
        GPX_TRY(potrf_panel(dtype, A, lda, N, rB, rB, 1024, info_dev, q, nullptr, 0, epB));
        if (g_hook) GPX_TRY(g_hook->fn(g_hook->user, rB, epA));
        route_hit(RT_POTRF_PAIR);
        while (pair_ok(k0)) {
            const int64_t rC = k0 + 2048, rD = rC + 1024, rE = rD + 1024;
            GPX_HIP(hipStreamWaitEvent(st, epB, 0));            // A and B are factored (B follows A on q)
            GPX_TRY(syrk_bc(dtype, N, rC, A, lda, rC, rD, at(k0, k0), lda, k0, 2048, 1024, 1, 0, st, info_dev, nullptr));   // U_a
            GPX_TRY(g_la.get(&e));
            GPX_HIP(hipEventRecord(e, st));
            GPX_HIP(hipStreamWaitEvent(q, e, 0));
            hipEvent_t epC, epD;
            GPX_TRY(g_la.get(&epC));
            GPX_TRY(potrf_panel(dtype, A, lda, N, rC, rC, 1024, info_dev, q, nullptr, 0, epC));
            GPX_TRY(syrk_bc(dtype, N, rD, A, lda, rD, rE, at(k0, k0), lda, k0, 2048, 1024, 1, 0, st, info_dev, nullptr));   // U_b
            GPX_TRY(g_la.get(&e));
            GPX_HIP(hipEventRecord(e, st));
            GPX_HIP(hipStreamWaitEvent(q, e, 0));
            // V: a panel-class product on the panel stream (it is part of the chain to panel D and runs beside U_c)
            GPX_TRY(gemm_nt(dtype, N - rD, 1024, 1024, at(rD, rC), lda, at(rD, rC), lda, at(rD, rD), lda, -1.0, GPX_LOWER, rD, rD, q));
            GPX_TRY(g_la.get(&epD));
            GPX_TRY(potrf_panel(dtype, A, lda, N, rD, rD, 1024, info_dev, q, nullptr, 0, epD));
            if (rE < n)
                GPX_TRY(syrk_bc(dtype, N, rE, A, lda, rE, n, at(k0, k0), lda, k0, 2048, 1024, 1, 0, st, info_dev, nullptr)); // U_c
            if (g_hook) {
                GPX_TRY(g_hook->fn(g_hook->user, rC, epB));
                GPX_TRY(g_hook->fn(g_hook->user, rD, epC));
            }
            k0 = rC; epA = epC; epB = epD;
        }
        // exit: A <- everything beyond B (K = 1024); then the loop below carries on with B as "the panel in flight"
        const int64_t rB2 = k0 + 1024, rC2 = k0 + 2048;
        GPX_HIP(hipStreamWaitEvent(st, epA, 0));
        if (rC2 < n)
            GPX_TRY(syrk_bc(dtype, N, rC2, A, lda, rC2, n, at(k0, k0), lda, k0, 1024, 1024, 1, 0, st, info_dev, nullptr));
        GPX_TRY(g_la.get(&e_rest));
        GPX_HIP(hipEventRecord(e_rest, st));
        k0 = rB2; kb = 1024; ep = epB;
    }
    while (true) {
        const int64_t r = k0 + kb;
        GPX_HIP(hipStreamWaitEvent(st, ep, 0));                 // panel k is factored
        if (r >= n) {
            if (g_hook && !bt) GPX_TRY(g_hook->fn(g_hook->user, n, ep));
            break;
        }
        const hipEvent_t ep_k = ep;
        const int64_t w1 = nominal(r), kb1 = std::min(w1, n - r);
        // block column k + 1 first, so that its panel can start ... (small n: the panel kernel applies panel k to its
        // own columns itself -- it then only waits for the trailing update of step k - 1, not for this stream's turn)
        const bool fold = panel_res_fold(N - r, kb, kb1, es, lda, A);
        hipEvent_t e_gate = nullptr;
        if (!fold) {
            GPX_TRY(syrk_bc(dtype, N, r, A, lda, r, r + kb1, at(k0, k0), lda, k0, kb, w1, 1, 0, st, info_dev, bt));
            GPX_TRY(g_la.get(&e));
            GPX_HIP(hipEventRecord(e, st));
            GPX_HIP(hipStreamWaitEvent(q, e, 0));
            // The panel has to get onto the chip BEFORE the rest of the update: that one follows the block-column update
            // directly in its stream, while the panel sits behind a cross-stream wait -- it used to start ~6 us late, found
            // every CU slot taken and ran for as long as the update did (n = 8192 with 512-wide blocks: the left half 504 us
            // beside a 441 us update, the right half 107 us after it; profiles/r04_timeline_n8192_nb512_before_gate.txt).
            // An event recorded on the panel's stream right behind its wait gates the rest of the update.
            if (!bt && N - r <= tune().gate_rows) {
                GPX_TRY(g_la.get(&e_gate));
                GPX_HIP(hipEventRecord(e_gate, q));
            }
        } else if (e_rest) {
            // Host-paced panels (single matrices, n <= 16384: the sizes whose panels take the update to their left themselves).  A wait on an
            // event of another stream that is still pending at ENQUEUE time becomes a barrier packet in front of the panel;
            // the command processor takes ~6 us over it even when the event has long fired by then (tools/sync_probe.hip:
            // 1.5 us between two kernels of a stream, 6.7 with a record, 15 - 18 with a cross-stream wait).  The update of
            // step k - 1 is done well before panel k ends, so the HOST waits for it here -- the stream wait below then costs
            // nothing and panel k + 1 queues directly behind panel k.  n = 8192: potrf 6.33 -> 6.09 ms, n = 4096: 2.19 ->
            // 2.09, n = 2048: 0.99 -> 0.96 (tools/r3_ab.sh).  The call is then no longer a pure enqueue (include/gpx.h).
            // (Round 4 tried leaving these waits to the command processor while many rows are left -- the panel's stream waits
            //  for the update's event, an event behind that wait gates the next update -- so that the host runs ahead: the
            //  update-to-update gap went from ~30 to ~22 us and the factorisation did not get faster: n = 8192 5.63 vs
            //  5.62 - 5.72 ms over four thresholds, 5.85 with every step device-paced; profiles/r04_ab_dev_paced.log.)
            if (host_paced) GPX_HIP(hipEventSynchronize(e_rest));   // (polling hipEventQuery instead: no faster, r04_ab_*_host_spin.log)
            GPX_HIP(hipStreamWaitEvent(q, e_rest, 0));
        }
        GPX_TRY(g_la.get(&ep));
        g_idle_chip = !bt && (e_gate != nullptr || (fold && (host_paced || !e_rest)));
        GPX_TRY(potrf_panel(dtype, A, lda, N, r, r, kb1, info_dev, q, bt, fold ? kb : 0, ep));
        g_idle_chip = false;
        // ... while the rest of the trailing matrix is updated underneath it
        if (r + kb1 < n) {
            if (e_gate) GPX_HIP(hipStreamWaitEvent(st, e_gate, 0));
            GPX_TRY(syrk_bc(dtype, N, r, A, lda, r + kb1, n, at(k0, k0), lda, k0, kb, w1, 1, 0, st, info_dev, bt));
        }
        GPX_TRY(g_la.get(&e_rest));
        GPX_HIP(hipEventRecord(e_rest, st));
        // (after the next panel and this step's update are on their way: the hook's launches never delay the chain)
        if (g_hook && !bt) GPX_TRY(g_hook->fn(g_hook->user, r, ep_k));
        k0 = r; kb = kb1;
    }
    return GPX_OK;
}

}  // namespace gpx

using namespace gpx;

extern "C" {

int gpx_d_potrf(int dtype, void *A, int64_t n, int64_t lda, int *info_dev, void *stream)
{
    gpx::StreamTurn turn__((hipStream_t)stream);     // (this thread's scratch buffers: one stream at a time, gpx_common.h)
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
    gpx::StreamTurn turn__((hipStream_t)stream);     // (this thread's scratch buffers: one stream at a time, gpx_common.h)
    GPX_TRY(ensure_device());
    GPX_ARG(dtype == GPX_F64 || dtype == GPX_F32, "dtype must be GPX_F64 or GPX_F32");
    GPX_ARG(n >= 0 && r0 >= 0 && c0 >= 0 && kb >= 0 && r0 + kb <= n, "bad dimensions");
    GPX_ARG(info_dev, "info_dev is NULL");
    if (kb == 0) return GPX_OK;
    GPX_ARG(A, "A is NULL");
    GPX_ARG(lda % 16 == 0 && ((uintptr_t)A) % 16 == 0 && c0 % 16 == 0, "A / lda / c0 must be 16-element aligned");
    return potrf_panel(dtype, A, lda, n, r0, c0, kb, info_dev, S(stream));
}
