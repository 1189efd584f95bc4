// gpx_gemm.hip -- C += alpha * A * B^T on the fp64 / fp32 matrix cores of gfx950.
//
// The one O(N^3) kernel of the GP fit path: the trailing SYRK/GEMM update of the
// blocked Cholesky (replaces the level-3 BLAS inside LAPACK dpotrf reached via
// scipy.linalg.cholesky, gp/gp.py:294), the left-looking panel updates, and the
// TRSM/SYRK updates of the posterior covariance (gp/gp.py:622-625).
//
// Shape: both operands are row-major with the reduction index contiguous
// ("NT"): C[i, j] += alpha * sum_k A[i, k] * B[j, k]  (alpha = -1 in the factorisation).  A lower Cholesky on row-major
// storage only ever needs this form.
//
// Roofline: fp64 MFMA (v_mfma_f64_16x16x4_f64: 2048 flop per wave-instruction).
// Algorithmic flops per launch = 2*M*N*K (M*N*K... halved for the lower-only
// SYRK form, where tiles strictly above the diagonal are skipped).
//
// Tiling: workgroup = 256 threads = 4 waves (2 x 2); block tile 128 x 128; each
// wave owns 64 x 64 = 4 x 4 MFMA tiles (128 accumulator VGPRs in fp64).  A
// k-step is 128 bytes of every operand row (16 doubles / 32 floats): one full
// cache line per row from HBM/L2, staged global -> registers -> LDS with the
// next k-step's loads in flight under the current step's 64 (128 for fp32)
// MFMAs, double-buffered LDS, one barrier per k-step.  The MFMA reduction index
// is permuted so that every lane reads 32 contiguous bytes of its row from LDS
// (2 x ds_read_b128) per k-step: lane (i, q) holds k = q*S + s for sub-step s.
#include "gpx_common.h"

namespace gpx {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float  f4_t __attribute__((ext_vector_type(4)));

template <typename T> struct MF;
template <> struct MF<double> {
    typedef d4_t acc_t;
    static constexpr int EPK = 16;   // elements per k-step (128 B per row)
    static constexpr int CH = 2;     // elements per 16-byte chunk
    __device__ static __forceinline__ acc_t mfma(double a, double b, acc_t c)
    { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    __device__ static __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct MF<float> {
    typedef f4_t acc_t;
    static constexpr int EPK = 32;
    static constexpr int CH = 4;
    __device__ static __forceinline__ acc_t mfma(float a, float b, acc_t c)
    { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
    __device__ static __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

constexpr int GB_M = 128, GB_N = 128;
constexpr int G_ROWB = 128;            // bytes of each operand row per k-step
constexpr int G_PITCH = G_ROWB + 16;   // LDS row pitch in bytes (16-B aligned, de-phased banks)
constexpr int G_TILE_BYTES = GB_M * G_PITCH;           // one operand tile in LDS
constexpr int G_SMEM = 2 * 2 * G_TILE_BYTES;           // double-buffered A and B: 73,728 B

struct alignas(16) Chunk16 { unsigned int w[4]; };

// Load 16 bytes (CH elements) of row `r`, elements [k, k + CH) with zero fill
// outside [0, rows) x [0, K).
template <typename T>
__device__ __forceinline__ Chunk16 load_chunk(const T *__restrict__ base, int64_t ld, int64_t r,
                                              int64_t rows, int64_t k, int64_t K)
{
    Chunk16 c;
    c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0u;
    if (r < rows && k < K) {
        const T *p = base + r * ld + k;
        if (k + MF<T>::CH <= K) {
            c = *reinterpret_cast<const Chunk16 *>(p);
        } else {
            T tmp[MF<T>::CH];
#pragma unroll
            for (int e = 0; e < MF<T>::CH; ++e) tmp[e] = (k + e < K) ? p[e] : (T)0;
            memcpy(&c, tmp, 16);
        }
    }
    return c;
}

// C (M x N) += alpha * A (M x K) * B (N x K)^T ; tri: skip/mask the strict upper part,
// where element (i, j) is upper iff row0 + i < col0 + j.
template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(int64_t M, int64_t N, int64_t K,
                                                         const T *__restrict__ A, int64_t lda,
                                                         const T *__restrict__ B, int64_t ldb,
                                                         T *__restrict__ C, int64_t ldc, T alpha,
                                                         int tri, int64_t row0, int64_t col0)
{
    typedef typename MF<T>::acc_t acc_t;
    constexpr int EPK = MF<T>::EPK;
    constexpr int CH = MF<T>::CH;
    constexpr int SUB = EPK / 4;        // MFMA sub-steps per k-step; also elements per lane per row

    const int64_t bm0 = (int64_t)blockIdx.y * GB_M;
    const int64_t bn0 = (int64_t)blockIdx.x * GB_N;
    if (tri == GPX_LOWER && col0 + bn0 > row0 + bm0 + GB_M - 1) return;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // global -> LDS staging map: 8 threads cover one 128-B row, 32 rows per pass
    const int s_row = tid >> 3;          // 0..31
    const int s_chk = tid & 7;           // 16-B chunk within the row
    Chunk16 ra[4], rb[4];

    auto load_tiles = [&](int64_t kbase) {
        const int64_t k = kbase + (int64_t)s_chk * CH;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            ra[p] = load_chunk<T>(A, lda, bm0 + s_row + 32 * p, M, k, K);
            rb[p] = load_chunk<T>(B, ldb, bn0 + s_row + 32 * p, N, k, K);
        }
    };
    auto store_tiles = [&](int buf) {
        unsigned char *ta = smem + (size_t)buf * 2 * G_TILE_BYTES;
        unsigned char *tb = ta + G_TILE_BYTES;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int off = (s_row + 32 * p) * G_PITCH + s_chk * 16;
            *reinterpret_cast<Chunk16 *>(ta + off) = ra[p];
            *reinterpret_cast<Chunk16 *>(tb + off) = rb[p];
        }
    };

    acc_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = (T)0;

    const int nk = (int)((K + EPK - 1) / EPK);
    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    // per-lane fragment address: row (l & 15) of the MFMA tile, bytes [32*q, 32*q + 32)
    const int f_off = (lane & 15) * G_PITCH + (lane >> 4) * 32;

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles((int64_t)(kt + 1) * EPK);

        const unsigned char *ta = smem + (size_t)cur * 2 * G_TILE_BYTES;
        const unsigned char *tb = ta + G_TILE_BYTES;
        T fa[4][SUB], fb[4][SUB];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned char *pa = ta + (wr * 64 + i * 16) * G_PITCH + f_off;
            const unsigned char *pb = tb + (wc * 64 + i * 16) * G_PITCH + f_off;
            Chunk16 a0 = *reinterpret_cast<const Chunk16 *>(pa);
            Chunk16 a1 = *reinterpret_cast<const Chunk16 *>(pa + 16);
            Chunk16 b0 = *reinterpret_cast<const Chunk16 *>(pb);
            Chunk16 b1 = *reinterpret_cast<const Chunk16 *>(pb + 16);
            memcpy(&fa[i][0], &a0, 16);
            memcpy(&fa[i][SUB / 2], &a1, 16);
            memcpy(&fb[i][0], &b0, 16);
            memcpy(&fb[i][SUB / 2], &b1, 16);
        }
#pragma unroll
        for (int s = 0; s < SUB; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = MF<T>::mfma(fa[i][s], fb[j][s], acc[i][j]);

        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }

    // epilogue: C += alpha * acc  (16 lanes = 16 consecutive columns = one 128-B / 64-B segment)
    const int ccol = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t gc = bn0 + wc * 64 + j * 16 + ccol;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gr = bm0 + wr * 64 + i * 16 + MF<T>::row(lane, r);
                if (gr < M && gc < N && !(tri == GPX_LOWER && row0 + gr < col0 + gc)) {
                    T *p = C + gr * ldc + gc;
                    *p = fma(alpha, acc[i][j][r], *p);
                }
            }
        }
    }
}

// number of C elements a launch updates (all of M x N, or those with row0+i >= col0+j)
static double updated_elements(int64_t M, int64_t N, int tri, int64_t row0, int64_t col0)
{
    if (tri != GPX_LOWER) return (double)M * (double)N;
    double cnt = 0;
    // column j is updated for rows i >= col0 + j - row0
    const int64_t off = col0 - row0;
    // closed form over j in [0, N): rows max(0, j + off) .. M-1
    for (int64_t j = 0; j < N; ++j) {
        const int64_t first = std::max<int64_t>(0, j + off);
        if (first < M) cnt += (double)(M - first);
    }
    return cnt;
}

template <typename T>
int launch_gemm_nt(int64_t M, int64_t N, int64_t K, const void *A, int64_t lda, const void *B,
                   int64_t ldb, void *C, int64_t ldc, double alpha, int tri, int64_t row0,
                   int64_t col0, hipStream_t st)
{
    if (M <= 0 || N <= 0 || K <= 0) return GPX_OK;
    static bool attr_done = false;
    if (!attr_done) {
        GPX_HIP(hipFuncSetAttribute((const void *)gemm_nt_kernel<T>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, G_SMEM));
        attr_done = true;
    }
    dim3 grid((unsigned)cdiv(N, GB_N), (unsigned)cdiv(M, GB_M)), block(256);
    ProfScope prof(PC_GEMM, 2.0 * (double)K * updated_elements(M, N, tri, row0, col0), st);
    hipLaunchKernelGGL((gemm_nt_kernel<T>), grid, block, G_SMEM, st, M, N, K, (const T *)A, lda,
                       (const T *)B, ldb, (T *)C, ldc, (T)alpha, tri, row0, col0);
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}

int gemm_nt(int dtype, int64_t M, int64_t N, int64_t K, const void *A, int64_t lda, const void *B,
            int64_t ldb, void *C, int64_t ldc, double alpha, int tri, int64_t row0, int64_t col0,
            hipStream_t st)
{
    if (dtype == GPX_F64)
        return launch_gemm_nt<double>(M, N, K, A, lda, B, ldb, C, ldc, alpha, tri, row0, col0, st);
    return launch_gemm_nt<float>(M, N, K, A, lda, B, ldb, C, ldc, alpha, tri, row0, col0, st);
}

}  // namespace gpx

using namespace gpx;

extern "C" int gpx_d_gemm_nt(int dtype, int64_t M, int64_t N, int64_t K, double alpha,
                             const void *A, int64_t lda, const void *B, int64_t ldb, void *C,
                             int64_t ldc, int tri, int64_t row0, int64_t col0, void *stream)
{
    GPX_TRY(ensure_device());
    GPX_ARG(dtype == GPX_F64 || dtype == GPX_F32, "dtype must be GPX_F64 or GPX_F32");
    GPX_ARG(M >= 0 && N >= 0 && K >= 0, "negative dimension");
    if (M == 0 || N == 0 || K == 0) return GPX_OK;
    GPX_ARG(A && B && C, "NULL pointer");
    const int64_t ch = 16 / (int64_t)esize(dtype);
    GPX_ARG(lda >= K && ldb >= K && ldc >= N, "leading dimension too small");
    GPX_ARG(lda % ch == 0 && ldb % ch == 0, "lda/ldb must be multiples of 16 bytes");
    GPX_ARG(((uintptr_t)A) % 16 == 0 && ((uintptr_t)B) % 16 == 0, "A/B must be 16-byte aligned");
    GPX_ARG(tri == GPX_FULL || tri == GPX_LOWER, "tri must be GPX_FULL or GPX_LOWER");
    return gemm_nt(dtype, M, N, K, A, lda, B, ldb, C, ldc, alpha, tri, row0, col0, S(stream));
}
