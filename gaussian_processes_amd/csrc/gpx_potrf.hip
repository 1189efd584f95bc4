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
//     for each 64-wide sub-column j0 inside it                     -- left-looking
//       (a) A[j0:, j0:j0+64] -= A[j0:, k0:j0] * A[j0:j0+64, k0:j0]^T    MFMA gemm_nt
//       (b) factor the 64 x 64 diagonal block in LDS (one workgroup)
//       (c) rows below: A[r, j0:j0+64] <- A[r, j0:j0+64] * Ljj^-T       (one lane per row)
//     trailing update A[k0+kb:, k0+kb:] -= P * P^T, P = A[k0+kb:, k0:k0+kb]  MFMA gemm_nt, lower
// >= 97 % of the flops at N = 65536 are in the trailing update (K = nb deep).
#include "gpx_common.h"

namespace gpx {

constexpr int IB = 64;
constexpr int IBP = IB + 1;   // LDS pitch (elements): conflict-free column walks

// ---- (b) diagonal block: unblocked right-looking Cholesky in LDS ----------
template <typename T>
__global__ __launch_bounds__(256) void potrf_diag_kernel(T *__restrict__ A, int64_t lda, int64_t j0,
                                                         int jb, int *__restrict__ info)
{
    __shared__ T s[IB * IBP];
    __shared__ T sdiag[IB];
    const int tid = threadIdx.x;
    T *blk = A + j0 * lda + j0;
    for (int idx = tid; idx < jb * jb; idx += 256) {
        const int i = idx / jb, c = idx - i * jb;
        s[i * IBP + c] = (c <= i) ? blk[(int64_t)i * lda + c] : (T)0;
    }
    __syncthreads();
    for (int j = 0; j < jb; ++j) {
        const T piv = s[j * IBP + j];
        if (!(piv > (T)0)) {                       // also catches NaN
            if (tid == 0 && *info == 0) *info = (int)(j0 + j + 1);
        }
        const T ljj = sqrt(piv);
        if (tid == 0) sdiag[j] = ljj;
        if (tid > j && tid < jb) s[tid * IBP + j] = s[tid * IBP + j] / ljj;
        __syncthreads();
        const int rem = jb - j - 1;
        for (int idx = tid; idx < rem * rem; idx += 256) {
            const int ii = idx / rem, cc = idx - ii * rem;
            if (cc <= ii) {
                const int i = j + 1 + ii, c = j + 1 + cc;
                s[i * IBP + c] = fma(-s[i * IBP + j], s[c * IBP + j], s[i * IBP + c]);
            }
        }
        __syncthreads();
    }
    for (int idx = tid; idx < jb * jb; idx += 256) {
        const int i = idx / jb, c = idx - i * jb;
        if (c < i) blk[(int64_t)i * lda + c] = s[i * IBP + c];
        else if (c == i) blk[(int64_t)i * lda + c] = sdiag[i];
    }
}

// ---- (c) X[r, 0:jb] <- X[r, 0:jb] * Ljj^-T : one lane per row --------------
// Forward substitution along the row: x_c = (a_c - sum_{t<c} x_t L[c,t]) / L[c,c].
template <typename T, bool FULL64>
__global__ __launch_bounds__(256) void trsm_rows_kernel(T *__restrict__ X, int64_t ldx, int64_t rows,
                                                        const T *__restrict__ Ljj, int64_t ldl, int jb)
{
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
                            hipStream_t st)
{
    if (rows <= 0 || jb <= 0) return GPX_OK;
    dim3 grid((unsigned)cdiv(rows, 256)), block(256);
    const bool vec_ok = (jb == IB) && (ldx % (16 / (int64_t)sizeof(T)) == 0) && (((uintptr_t)X) % 16 == 0);
    ProfScope prof(PC_TRSM_ROWS, (double)rows * jb * jb, st);
    if (vec_ok)
        hipLaunchKernelGGL((trsm_rows_kernel<T, true>), grid, block, 0, st, (T *)X, ldx, rows,
                           (const T *)Ljj, ldl, jb);
    else
        hipLaunchKernelGGL((trsm_rows_kernel<T, false>), grid, block, 0, st, (T *)X, ldx, rows,
                           (const T *)Ljj, ldl, jb);
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}

int trsm_rows(int dtype, void *X, int64_t ldx, int64_t rows, const void *Ljj, int64_t ldl, int jb,
              hipStream_t st)
{
    if (dtype == GPX_F64) return launch_trsm_rows<double>(X, ldx, rows, Ljj, ldl, jb, st);
    return launch_trsm_rows<float>(X, ldx, rows, Ljj, ldl, jb, st);
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

static int64_t outer_block(int64_t n)
{
    const char *env = getenv("GPX_POTRF_NB");
    if (env) {
        int64_t v = atoll(env);
        if (v >= IB && v % IB == 0) return v;
    }
    if (n <= 2048) return 128;
    if (n <= 16384) return 256;
    return 512;
}

template <typename T>
static int potrf_t(T *A, int64_t n, int64_t lda, int *info_dev, hipStream_t st, int dtype)
{
    GPX_HIP(hipMemsetAsync(info_dev, 0, sizeof(int), st));
    const int64_t nb = outer_block(n);
    for (int64_t k0 = 0; k0 < n; k0 += nb) {
        const int64_t kb = std::min(nb, n - k0);
        for (int64_t j0 = k0; j0 < k0 + kb; j0 += IB) {
            const int jb = (int)std::min<int64_t>(IB, k0 + kb - j0);
            T *Aj = A + j0 * lda;                       // row j0
            if (j0 > k0)
                GPX_TRY(gemm_nt(dtype, n - j0, jb, j0 - k0, Aj + k0, lda, Aj + k0, lda, Aj + j0, lda,
                                -1.0, GPX_LOWER, 0, 0, st));
            {
                ProfScope prof(PC_POTRF_DIAG, (double)jb * jb * jb / 3.0, st);
                hipLaunchKernelGGL((potrf_diag_kernel<T>), dim3(1), dim3(256), 0, st, A, lda, j0, jb,
                                   info_dev);
            }
            GPX_LAUNCH_CHECK();
            const int64_t below = n - (j0 + jb);
            if (below > 0)
                GPX_TRY(trsm_rows(dtype, A + (j0 + jb) * lda + j0, lda, below, Aj + j0, lda, jb, st));
        }
        const int64_t r = k0 + kb;
        if (r < n) {
            T *P = A + r * lda + k0;
            GPX_TRY(gemm_nt(dtype, n - r, n - r, kb, P, lda, P, lda, A + r * lda + r, lda, -1.0,
                            GPX_LOWER, 0, 0, st));
        }
    }
    return GPX_OK;
}

int potrf(int dtype, void *A, int64_t n, int64_t lda, int *info_dev, hipStream_t st)
{
    if (dtype == GPX_F64) return potrf_t<double>((double *)A, n, lda, info_dev, st, dtype);
    return potrf_t<float>((float *)A, n, lda, info_dev, st, dtype);
}

}  // namespace gpx

using namespace gpx;

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
