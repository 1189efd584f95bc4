// gpx_solve.hip -- triangular solves and O(n) reductions on the factor (gfx950).
//
// Replaces scipy.linalg.cho_solve((L, True), y) of gp/gp.py:332-334 (LAPACK
// dpotrs), np.linalg.slogdet(K) of gp/ext/gp_c.pyx:21 (a second, redundant LU in
// the reference; here 2*sum(log diag L)), np.dot(y, Kiy) of gp_c.pyx:26, and the
// explicit-inverse route to the posterior covariance (gp/gp.py:311-312,622-625).
//
// Roofline: HBM read bandwidth -- a single-right-hand-side solve reads the
// lower triangle of L once per direction (n^2/2 * sizeof(T) bytes, 2 n^2 flop for
// both directions together).  The 64 x 64 diagonal blocks are inverted up front in
// ONE batched launch (trinv64_kernel); each 64-wide block step is then one launch
// whose workgroups all form z = inv(L_jj) b_j by a 64 x 64 mat-vec (no serial
// substitution on the critical path) and stream their own slice of the panel
// below (forward) / to the left (backward) in coalesced rows.
#include "gpx_common.h"

namespace gpx {

constexpr int SB = 64;
constexpr int SBP = SB + 1;

// ---- batched inverse of the 64 x 64 diagonal blocks --------------------------
// One workgroup per block (all blocks in one launch); lane c builds column c of
// inv(L_jj) by forward substitution.  Blocks shorter than 64 are padded with the
// identity.  Output: Linv[blk][64][64], row-major, strictly-upper part zero.
template <typename T>
__global__ __launch_bounds__(64) void trinv64_kernel(const T *__restrict__ L, int64_t ldl, int64_t ncols,
                                                     T *__restrict__ Linv)
{
    __shared__ T sL[SB * SBP];
    __shared__ T sX[SB * SBP];
    const int c = threadIdx.x;
    const int64_t k0 = (int64_t)blockIdx.x * SB;
    const int jb = (int)min((int64_t)SB, ncols - k0);
    const T *blk = L + k0 * ldl + k0;
    for (int i = 0; i < SB; ++i) {
        T v = (i == c) ? (T)1 : (T)0;
        if (i < jb && c <= i) v = blk[(int64_t)i * ldl + c];
        sL[i * SBP + c] = v;
    }
    __syncthreads();
    // column c of X = L^-1: X[c][c] = 1 / L[c][c]; X[i][c] = -(sum_{t=c}^{i-1} L[i][t] X[t][c]) / L[i][i]
    for (int i = 0; i < c; ++i) sX[i * SBP + c] = (T)0;
    sX[c * SBP + c] = (T)1 / sL[c * SBP + c];
    for (int i = c + 1; i < SB; ++i) {
        T acc = (T)0;
        for (int t = c; t < i; ++t) acc = fma(sL[i * SBP + t], sX[t * SBP + c], acc);
        sX[i * SBP + c] = -acc / sL[i * SBP + i];
    }
    __syncthreads();
    T *out = Linv + (int64_t)blockIdx.x * SB * SB;
    for (int i = 0; i < SB; ++i) out[i * SB + c] = sX[i * SBP + c];
}

// grow-only device scratch for the block inverses (one per host thread)
struct SolveScratch { void *p = nullptr; size_t bytes = 0; int device = -1; };
static thread_local SolveScratch g_scr;
static int scratch(size_t bytes, void **out)
{
    int dev = 0;
    GPX_HIP(hipGetDevice(&dev));
    if (g_scr.device != dev || g_scr.bytes < bytes) {
        if (g_scr.p && g_scr.device == dev) { GPX_HIP(hipDeviceSynchronize()); (void)hipFree(g_scr.p); }
        g_scr.p = nullptr; g_scr.bytes = 0; g_scr.device = dev;
        GPX_HIP(hipMalloc(&g_scr.p, bytes));
        g_scr.bytes = bytes;
    }
    *out = g_scr.p;
    return GPX_OK;
}

// z = Linv_blk * v (forward) or Linv_blk^T * v (backward) for one 64-block, by all
// 256 threads of the workgroup; v in LDS (sv), result to LDS (sz).
template <typename T, bool TRANS>
__device__ __forceinline__ void block_matvec(const T *__restrict__ Li, const T *sv, T *sz, T *red, int tid)
{
    const int r = tid & 63, part = tid >> 6;            // 4 partial sums per output row
    T acc = (T)0;
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        const int c = part * 16 + cc;
        const T m = TRANS ? Li[c * SB + r] : Li[r * SB + c];
        acc = fma(m, sv[c], acc);
    }
    red[part * SB + r] = acc;
    __syncthreads();
    if (tid < SB) sz[tid] = ((red[tid] + red[SB + tid]) + red[2 * SB + tid]) + red[3 * SB + tid];
    __syncthreads();
}

// forward: z = inv(L_jj) b[k0:k0+jb], x[k0:k0+jb] = z, b[r] -= L[r, k0:k0+jb] . z for r >= k0 + jb
template <typename T>
__global__ __launch_bounds__(256) void trsv_fwd_step(const T *__restrict__ L, int64_t ldl,
                                                     const T *__restrict__ Linv, T *__restrict__ b,
                                                     T *__restrict__ x, int64_t k0, int jb, int64_t n)
{
    __shared__ T sTile[SB * SBP];
    __shared__ T sv[SB], sz[SB], red[4 * SB];
    const int tid = threadIdx.x;
    if (tid < SB) sv[tid] = (tid < jb) ? b[k0 + tid] : (T)0;
    // this workgroup's slice of the panel below: 64 rows x jb columns, coalesced along the row
    const int64_t r0 = k0 + jb + (int64_t)blockIdx.x * SB;
    for (int idx = tid; idx < SB * SB; idx += 256) {
        const int i = idx >> 6, c = idx & 63;
        sTile[i * SBP + c] = (r0 + i < n && c < jb) ? L[(r0 + i) * ldl + k0 + c] : (T)0;
    }
    __syncthreads();
    block_matvec<T, false>(Linv + (k0 / SB) * SB * SB, sv, sz, red, tid);
    if (blockIdx.x == 0 && tid < jb) x[k0 + tid] = sz[tid];
    // 4 lanes per row, 16 columns each
    const int row = tid >> 2, part = tid & 3;
    T acc = (T)0;
#pragma unroll
    for (int c = 0; c < 16; ++c) acc = fma(sTile[row * SBP + part * 16 + c], sz[part * 16 + c], acc);
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0 && r0 + row < n) b[r0 + row] -= acc;
}

// backward: a = inv(L_jj)^T b[k0:k0+jb], x[k0:k0+jb] = a, b[c] -= sum_i L[k0+i, c] a_i for c < k0
template <typename T>
__global__ __launch_bounds__(256) void trsv_bwd_step(const T *__restrict__ L, int64_t ldl,
                                                     const T *__restrict__ Linv, T *__restrict__ b,
                                                     T *__restrict__ x, int64_t k0, int jb)
{
    __shared__ T sv[SB], sz[SB], red[4 * SB];
    const int tid = threadIdx.x;
    if (tid < SB) sv[tid] = (tid < jb) ? b[k0 + tid] : (T)0;
    __syncthreads();
    block_matvec<T, true>(Linv + (k0 / SB) * SB * SB, sv, sz, red, tid);
    if (blockIdx.x == 0 && tid < jb) x[k0 + tid] = sz[tid];
    const int64_t c = (int64_t)blockIdx.x * 256 + tid;
    if (c < k0) {
        const T *col = L + k0 * ldl + c;
        T acc = (T)0;
#pragma unroll 8
        for (int i = 0; i < jb; ++i) acc = fma(col[(int64_t)i * ldl], sz[i], acc);
        b[c] -= acc;
    }
}

template <typename T>
static int trsv_t(const T *L, int64_t n, int64_t ldl, T *b, T *x, int transpose, hipStream_t st,
                  int64_t ncols = -1)
{
    // ncols < n (forward only): the matrix is a trapezoid -- an ncols x ncols lower
    // triangle on top of (n - ncols) further rows; x[0:ncols] is solved and the
    // remaining right-hand side b[ncols:n] is reduced by L[ncols:n, 0:ncols] x.
    if (ncols < 0 || ncols > n) ncols = n;
    ProfScope prof(PC_TRSV, ((double)n * ncols - 0.5 * (double)ncols * (ncols - 1)) * sizeof(T), st);
    const int64_t nblk = cdiv(ncols, SB);
    void *scr = nullptr;
    GPX_TRY(scratch((size_t)nblk * SB * SB * sizeof(T), &scr));
    T *Linv = (T *)scr;
    hipLaunchKernelGGL((trinv64_kernel<T>), dim3((unsigned)nblk), dim3(64), 0, st, L, ldl, ncols, Linv);
    if (!transpose) {
        for (int64_t k0 = 0; k0 < ncols; k0 += SB) {
            const int jb = (int)std::min<int64_t>(SB, ncols - k0);
            const int64_t below = n - k0 - jb;
            dim3 grid((unsigned)std::max<int64_t>(1, cdiv(below, SB))), block(256);
            hipLaunchKernelGGL((trsv_fwd_step<T>), grid, block, 0, st, L, ldl, Linv, b, x, k0, jb, n);
        }
    } else {
        for (int64_t kb = nblk - 1; kb >= 0; --kb) {
            const int64_t k0 = kb * SB;
            const int jb = (int)std::min<int64_t>(SB, n - k0);
            dim3 grid((unsigned)std::max<int64_t>(1, cdiv(k0, 256))), block(256);
            hipLaunchKernelGGL((trsv_bwd_step<T>), grid, block, 0, st, L, ldl, Linv, b, x, k0, jb);
        }
    }
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}

int trsv_lower(int dtype, const void *L, int64_t n, int64_t ldl, void *b, void *x, int transpose,
               hipStream_t st)
{
    if (n <= 0) return GPX_OK;
    if (dtype == GPX_F64) return trsv_t<double>((const double *)L, n, ldl, (double *)b, (double *)x, transpose, st);
    return trsv_t<float>((const float *)L, n, ldl, (float *)b, (float *)x, transpose, st);
}

int trsv_lower_cols(int dtype, const void *L, int64_t n, int64_t ldl, int64_t ncols, void *b, void *x,
                    hipStream_t st)
{
    if (n <= 0 || ncols <= 0) return GPX_OK;
    if (dtype == GPX_F64)
        return trsv_t<double>((const double *)L, n, ldl, (double *)b, (double *)x, 0, st, ncols);
    return trsv_t<float>((const float *)L, n, ldl, (float *)b, (float *)x, 0, st, ncols);
}

// y[c] -= sum_r Lp[r, c] * x[r]   (c < ncols <= 1024, r < rows): the transposed
// panel product of the distributed back substitution.  Two stages, fixed
// summation order: 256-row chunks -> partial sums in `work`, then one reduction.
template <typename T>
__global__ __launch_bounds__(256) void panel_gemv_t_partial(const T *__restrict__ Lp, int64_t ldl,
                                                            int64_t rows, int ncols,
                                                            const T *__restrict__ x, double *__restrict__ work)
{
    __shared__ T sx[256];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 256;
    sx[tid] = (r0 + tid < rows) ? x[r0 + tid] : (T)0;
    __syncthreads();
    const int nr = (int)std::min<int64_t>(256, rows - r0);
    for (int c = tid; c < ncols; c += 256) {
        const T *col = Lp + r0 * ldl + c;
        double acc = 0.0;
        for (int i = 0; i < nr; ++i) acc = fma((double)col[(int64_t)i * ldl], (double)sx[i], acc);
        work[(int64_t)blockIdx.x * ncols + c] = acc;
    }
}

template <typename T>
__global__ void panel_gemv_t_reduce(const double *__restrict__ work, int64_t nchunks, int ncols,
                                    T *__restrict__ y)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    double acc = 0.0;
    for (int64_t k = 0; k < nchunks; ++k) acc += work[k * ncols + c];
    y[c] = (T)((double)y[c] - acc);
}

int panel_gemv_t(int dtype, const void *Lp, int64_t ldl, int64_t rows, int64_t ncols, const void *x,
                 void *y, double *work, hipStream_t st)
{
    if (rows <= 0 || ncols <= 0) return GPX_OK;
    const int64_t nchunks = cdiv(rows, 256);
    ProfScope prof(PC_TRSV, (double)rows * ncols * esize(dtype), st);
    if (dtype == GPX_F64) {
        hipLaunchKernelGGL((panel_gemv_t_partial<double>), dim3((unsigned)nchunks), dim3(256), 0, st,
                           (const double *)Lp, ldl, rows, (int)ncols, (const double *)x, work);
        hipLaunchKernelGGL((panel_gemv_t_reduce<double>), dim3((unsigned)cdiv(ncols, 256)), dim3(256), 0, st,
                           work, nchunks, (int)ncols, (double *)y);
    } else {
        hipLaunchKernelGGL((panel_gemv_t_partial<float>), dim3((unsigned)nchunks), dim3(256), 0, st,
                           (const float *)Lp, ldl, rows, (int)ncols, (const float *)x, work);
        hipLaunchKernelGGL((panel_gemv_t_reduce<float>), dim3((unsigned)cdiv(ncols, 256)), dim3(256), 0, st,
                           work, nchunks, (int)ncols, (float *)y);
    }
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}

// X (m x n) <- X * L^-T, blocked and RIGHT-looking: after the columns of a block are
// solved they are applied at once to every remaining column,
//   X[:, r:] -= X[:, blk] * L[r:, blk]^T      (m x (n - r) output, K = block width),
// which keeps all CUs busy even for few right-hand sides (m ~ 1000 test points); a
// left-looking sweep would launch m/256 x 2 tiles per step.  Inside a block: 64-wide
// substitutions by trsm_rows (shared with the Cholesky panel) with small MFMA updates.
int trsm_right_lt(int dtype, const void *L, int64_t n, int64_t ldl, void *X, int64_t m, int64_t ldx,
                  hipStream_t st)
{
    if (n <= 0 || m <= 0) return GPX_OK;
    const size_t es = esize(dtype);
    const int64_t NB = n >= 8192 ? 512 : 256;
    auto Lp = [&](int64_t r, int64_t c) { return (const char *)L + (r * ldl + c) * es; };
    auto Xp = [&](int64_t c) { return (char *)X + c * es; };
    for (int64_t k0 = 0; k0 < n; k0 += NB) {
        const int64_t kb = std::min(NB, n - k0);
        for (int64_t j0 = k0; j0 < k0 + kb; j0 += SB) {
            const int jb = (int)std::min<int64_t>(SB, k0 + kb - j0);
            if (j0 > k0)
                GPX_TRY(gemm_nt(dtype, m, jb, j0 - k0, Xp(k0), ldx, Lp(j0, k0), ldl, Xp(j0), ldx, -1.0,
                                GPX_FULL, 0, 0, st));
            GPX_TRY(trsm_rows(dtype, Xp(j0), ldx, m, Lp(j0, j0), ldl, jb, st));
        }
        const int64_t r = k0 + kb;
        if (r < n)
            GPX_TRY(gemm_nt(dtype, m, n - r, kb, Xp(k0), ldx, Lp(r, k0), ldl, Xp(r), ldx, -1.0, GPX_FULL, 0, 0,
                            st));
    }
    return GPX_OK;
}

// ---- reductions (single workgroup, fixed order => deterministic) ----------
template <typename T, int MODE>   // MODE 0: sum a[i]*b[i]   1: 2*sum log a[i*stride]
__global__ __launch_bounds__(1024) void reduce_kernel(const T *__restrict__ a, const T *__restrict__ b,
                                                      int64_t n, int64_t stride, double *__restrict__ out)
{
    __shared__ double red[16];
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (int64_t i = tid; i < n; i += 1024) {
        if (MODE == 0) acc = fma((double)a[i], (double)b[i], acc);
        else acc += log((double)a[i * stride]);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += red[w];
        out[0] = (MODE == 0) ? s : 2.0 * s;
    }
}

int logdet_chol(int dtype, const void *L, int64_t n, int64_t ldl, double *out_dev, hipStream_t st)
{
    if (dtype == GPX_F64)
        hipLaunchKernelGGL((reduce_kernel<double, 1>), dim3(1), dim3(1024), 0, st, (const double *)L,
                           (const double *)nullptr, n, ldl + 1, out_dev);
    else
        hipLaunchKernelGGL((reduce_kernel<float, 1>), dim3(1), dim3(1024), 0, st, (const float *)L,
                           (const float *)nullptr, n, ldl + 1, out_dev);
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}

int dot(int dtype, const void *a, const void *b, int64_t n, double *out_dev, hipStream_t st)
{
    if (dtype == GPX_F64)
        hipLaunchKernelGGL((reduce_kernel<double, 0>), dim3(1), dim3(1024), 0, st, (const double *)a,
                           (const double *)b, n, 1, out_dev);
    else
        hipLaunchKernelGGL((reduce_kernel<float, 0>), dim3(1), dim3(1024), 0, st, (const float *)a,
                           (const float *)b, n, 1, out_dev);
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}

}  // namespace gpx

using namespace gpx;

extern "C" {

int gpx_d_trsv_lower(int dtype, const void *L, int64_t n, int64_t ldl, void *b, void *x,
                     int transpose, void *stream)
{
    GPX_TRY(ensure_device());
    GPX_ARG(dtype == GPX_F64 || dtype == GPX_F32, "dtype must be GPX_F64 or GPX_F32");
    GPX_ARG(n >= 0, "n < 0");
    if (n == 0) return GPX_OK;
    GPX_ARG(L && b && x && b != x, "NULL pointer or b == x");
    GPX_ARG(ldl >= n, "ldl < n");
    return trsv_lower(dtype, L, n, ldl, b, x, transpose, S(stream));
}

int gpx_d_trsm_right_lt(int dtype, const void *L, int64_t n, int64_t ldl, void *X, int64_t m,
                        int64_t ldx, void *stream)
{
    GPX_TRY(ensure_device());
    GPX_ARG(dtype == GPX_F64 || dtype == GPX_F32, "dtype must be GPX_F64 or GPX_F32");
    GPX_ARG(n >= 0 && m >= 0, "negative dimension");
    if (n == 0 || m == 0) return GPX_OK;
    GPX_ARG(L && X, "NULL pointer");
    GPX_ARG(ldl >= n && ldx >= n, "leading dimension too small");
    GPX_ARG(ldl % 16 == 0 && ldx % 16 == 0, "ldl/ldx must be multiples of 16 elements");
    GPX_ARG(((uintptr_t)L) % 16 == 0 && ((uintptr_t)X) % 16 == 0, "L/X must be 16-byte aligned");
    return trsm_right_lt(dtype, L, n, ldl, X, m, ldx, S(stream));
}

int gpx_d_logdet_chol(int dtype, const void *L, int64_t n, int64_t ldl, double *out_dev, void *stream)
{
    GPX_TRY(ensure_device());
    GPX_ARG(dtype == GPX_F64 || dtype == GPX_F32, "dtype must be GPX_F64 or GPX_F32");
    GPX_ARG(n >= 0 && out_dev && (n == 0 || L), "bad arguments");
    return logdet_chol(dtype, L, n, ldl, out_dev, S(stream));
}

int gpx_d_dot(int dtype, const void *a, const void *b, int64_t n, double *out_dev, void *stream)
{
    GPX_TRY(ensure_device());
    GPX_ARG(dtype == GPX_F64 || dtype == GPX_F32, "dtype must be GPX_F64 or GPX_F32");
    GPX_ARG(n >= 0 && out_dev && (n == 0 || (a && b)), "bad arguments");
    return dot(dtype, a, b, n, out_dev, S(stream));
}

}  // extern "C"

extern "C" {

int gpx_d_trsv_lower_cols(int dtype, const void *L, int64_t n, int64_t ldl, int64_t ncols, void *b,
                          void *x, void *stream)
{
    GPX_TRY(ensure_device());
    GPX_ARG(dtype == GPX_F64 || dtype == GPX_F32, "dtype must be GPX_F64 or GPX_F32");
    GPX_ARG(n >= 0 && ncols >= 0 && ncols <= n, "need 0 <= ncols <= n");
    if (n == 0 || ncols == 0) return GPX_OK;
    GPX_ARG(L && b && x && b != x, "NULL pointer or b == x");
    GPX_ARG(ldl >= ncols, "ldl < ncols");
    return trsv_lower_cols(dtype, L, n, ldl, ncols, b, x, S(stream));
}

int gpx_d_panel_gemv_t(int dtype, const void *Lp, int64_t ldl, int64_t rows, int64_t ncols,
                       const void *x, void *y, void *work, void *stream)
{
    GPX_TRY(ensure_device());
    GPX_ARG(dtype == GPX_F64 || dtype == GPX_F32, "dtype must be GPX_F64 or GPX_F32");
    GPX_ARG(rows >= 0 && ncols >= 0, "negative dimension");
    if (rows == 0 || ncols == 0) return GPX_OK;
    GPX_ARG(Lp && x && y && work, "NULL pointer");
    GPX_ARG(ldl >= ncols, "ldl < ncols");
    return panel_gemv_t(dtype, Lp, ldl, rows, ncols, x, y, (double *)work, S(stream));
}

}  // extern "C"
