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

// Epilogue shared by both kernels: C += alpha * acc for one wave's 64 x 64 tile.
// All 16 loads of an MFMA tile row are issued before the first use (clamped
// addresses keep them unconditional), so a lane pays 4 memory round trips per
// tile instead of 64.
template <typename T>
__device__ __forceinline__ void store_wave_tile(typename MF<T>::acc_t (&acc)[4][4], T *__restrict__ C,
                                                int64_t ldc, int64_t M, int64_t N, int64_t r_base,
                                                int64_t c_base, int lane, T alpha, int tri,
                                                int64_t row0, int64_t col0)
{
    const int ccol = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        T cv[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t gc = min(c_base + j * 16 + ccol, N - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gr = min(r_base + i * 16 + MF<T>::row(lane, r), M - 1);
                cv[j][r] = C[gr * ldc + gc];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t gc = c_base + j * 16 + ccol;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gr = r_base + i * 16 + MF<T>::row(lane, r);
                if (gr < M && gc < N && !(tri == GPX_LOWER && row0 + gr < col0 + gc))
                    C[gr * ldc + gc] = fma(alpha, acc[i][j][r], cv[j][r]);
            }
        }
    }
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
    store_wave_tile<T>(acc, C, ldc, M, N, bm0 + wr * 64, bn0 + wc * 64, lane, alpha, tri, row0, col0);
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


// ===========================================================================
// Fast path: K % (128 B) == 0, 16-B aligned operands.
//   block tile 256 x 128, 512 threads = 8 waves (4 x 2), wave tile 64 x 64;
//   operands go HBM/L2 -> LDS directly (global_load_lds_dwordx4, no VGPR
//   staging), 3 LDS stages of 48 KiB (384 rows x 128 B), two k-steps in flight,
//   one raw s_barrier per k-step with a counted vmcnt (never 0 in the loop);
//   LDS rows are unpadded (the DMA writes 1 KiB = 8 rows contiguously), bank
//   conflicts of the ds_read_b128 fragment reads are removed by an XOR swizzle
//   applied to the per-lane GLOBAL source chunk and again on the read;
//   tiles are walked in 4 x 8 patches (1024 x 1024 elements) and patch p is
//   given to the blocks with blockIdx % 8 == p % 8, so that the 32 blocks that
//   run together on one XCD share 4 A-panels and 8 B-panels in that XCD's L2.
// ===========================================================================
constexpr int F_BM = 256, F_BN = 128;
constexpr int F_ROWS = F_BM + F_BN;            // 384 tile rows per stage
constexpr int F_STAGE = F_ROWS * G_ROWB;       // 49,152 B
constexpr int F_NST = 3;
constexpr int F_SMEM = F_NST * F_STAGE;        // 147,456 B

// chunk swizzle of tile row r: a permutation of (r >> 1) & 7 chosen so that every
// 16-lane group of a ds_read_b128 fragment read hits 16 distinct 16-B slots
__device__ __forceinline__ int f_swz(int r) { return (0x64753120u >> (4 * ((r >> 1) & 7))) & 7; }

struct FastMap { int np; int pbc; int tri_enum; };

typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
#define GPX_DSR(dst, addr, off) \
    asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))

template <typename T>
__global__ __launch_bounds__(512, 2) void gemm_nt_fast_kernel(int64_t M, int64_t N, int64_t K,
                                                              const T *__restrict__ A, int64_t lda,
                                                              const T *__restrict__ B, int64_t ldb,
                                                              T *__restrict__ C, int64_t ldc, T alpha,
                                                              int tri, int64_t row0, int64_t col0,
                                                              FastMap fm)
{
    typedef typename MF<T>::acc_t acc_t;
    constexpr int EPK = MF<T>::EPK;
    constexpr int SUB = EPK / 4;

    // ---- block -> tile (XCD-aware patch order) ----
    const int bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int patch = (loc >> 5) * 8 + xcd, within = loc & 31;
    if (patch >= fm.np) return;
    int pb_r, pb_c;
    if (fm.tri_enum) {
        int b = (int)((sqrt(8.0 * (double)patch + 1.0) - 1.0) * 0.5);
        while ((b + 1) * (b + 2) / 2 <= patch) ++b;
        while (b * (b + 1) / 2 > patch) --b;
        pb_r = b; pb_c = patch - b * (b + 1) / 2;
    } else {
        pb_r = patch / fm.pbc; pb_c = patch - pb_r * fm.pbc;
    }
    const int64_t bm0 = ((int64_t)pb_r * 4 + (within >> 3)) * F_BM;
    const int64_t bn0 = ((int64_t)pb_c * 8 + (within & 7)) * F_BN;
    if (bm0 >= M || bn0 >= N) return;
    if (tri == GPX_LOWER && col0 + bn0 > row0 + bm0 + F_BM - 1) return;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // ---- DMA source pointers: wave w owns pieces 6w .. 6w+5 (1 KiB = 8 rows each) of every stage ----
    const unsigned char *gsrc[6];
    {
        const int r_in = lane >> 3, c = lane & 7;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int tr = 8 * (wave * 6 + j) + r_in;          // tile row 0..383
            const int g = c ^ f_swz(tr);                       // global 16-B chunk held by LDS chunk c
            if (tr < F_BM) {
                const int64_t r = min(bm0 + tr, M - 1);
                gsrc[j] = reinterpret_cast<const unsigned char *>(A + r * lda) + g * 16;
            } else {
                const int64_t r = min(bn0 + (tr - F_BM), N - 1);
                gsrc[j] = reinterpret_cast<const unsigned char *>(B + r * ldb) + g * 16;
            }
        }
    }
    auto issue = [&](int kt, int stage) {
        unsigned char *dst = smem + stage * F_STAGE + (wave * 6) * 1024;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(gsrc[j] + (size_t)kt * G_ROWB),
                (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, 0, 0);
        }
    };

    acc_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = (T)0;

    const int nk = (int)(K / EPK);
    issue(0, 0);
    if (nk > 1) issue(1, 1);

    // Fragment reads are inline asm: hipcc cannot prove that the in-flight LDS-DMA
    // writes (other stages) do not alias them and would otherwise drain vmcnt(0)
    // before every k-step's first ds_read.  Ordering is by hand: the counted vmcnt
    // + barrier makes the stage visible, lgkmcnt(0) + sched_barrier fences the
    // MFMAs behind the reads.
    // per-lane read address: row (lane & 15) of a 16-row MFMA tile, chunks 2q, 2q+1 swizzled
    const int q = lane >> 4;
    const int fl = f_swz(lane & 15);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const unsigned rd0 = (unsigned)((lane & 15) * G_ROWB + ((2 * q) ^ fl) * 16);
    const unsigned rd1 = (unsigned)((lane & 15) * G_ROWB + ((2 * q + 1) ^ fl) * 16);
    const unsigned a_base = lds0 + (unsigned)((wr * 64) * G_ROWB);
    const unsigned b_base = lds0 + (unsigned)((F_BM + wc * 64) * G_ROWB);

    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else             asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) {
            int ns = stage + 2; if (ns >= F_NST) ns -= F_NST;
            issue(kt + 2, ns);
        }
        const unsigned so = (unsigned)(stage * F_STAGE);
        const unsigned aa0 = a_base + so + rd0, aa1 = a_base + so + rd1;
        const unsigned ab0 = b_base + so + rd0, ab1 = b_base + so + rd1;
        u4_t ra[4][2], rb[4][2];
        GPX_DSR(ra[0][0], aa0, 0);    GPX_DSR(ra[0][1], aa1, 0);
        GPX_DSR(rb[0][0], ab0, 0);    GPX_DSR(rb[0][1], ab1, 0);
        GPX_DSR(ra[1][0], aa0, 2048); GPX_DSR(ra[1][1], aa1, 2048);
        GPX_DSR(rb[1][0], ab0, 2048); GPX_DSR(rb[1][1], ab1, 2048);
        GPX_DSR(ra[2][0], aa0, 4096); GPX_DSR(ra[2][1], aa1, 4096);
        GPX_DSR(rb[2][0], ab0, 4096); GPX_DSR(rb[2][1], ab1, 4096);
        GPX_DSR(ra[3][0], aa0, 6144); GPX_DSR(ra[3][1], aa1, 6144);
        GPX_DSR(rb[3][0], ab0, 6144); GPX_DSR(rb[3][1], ab1, 6144);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        T fa[4][SUB], fb[4][SUB];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            memcpy(&fa[i][0], &ra[i][0], 16);
            memcpy(&fa[i][SUB / 2], &ra[i][1], 16);
            memcpy(&fb[i][0], &rb[i][0], 16);
            memcpy(&fb[i][SUB / 2], &rb[i][1], 16);
        }
#pragma unroll
        for (int s = 0; s < SUB; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = MF<T>::mfma(fa[i][s], fb[j][s], acc[i][j]);
        ++stage; if (stage >= F_NST) stage = 0;
    }

    store_wave_tile<T>(acc, C, ldc, M, N, bm0 + wr * 64, bn0 + wc * 64, lane, alpha, tri, row0, col0);
}
#undef GPX_DSR

template <typename T>
static int launch_gemm_nt_fast(int64_t M, int64_t N, int64_t K, const void *A, int64_t lda,
                               const void *B, int64_t ldb, void *C, int64_t ldc, double alpha, int tri,
                               int64_t row0, int64_t col0, hipStream_t st)
{
    static bool attr_done = false;
    if (!attr_done) {
        GPX_HIP(hipFuncSetAttribute((const void *)gemm_nt_fast_kernel<T>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, F_SMEM));
        attr_done = true;
    }
    const int64_t tm = cdiv(M, F_BM), tn = cdiv(N, F_BN);
    FastMap fm;
    fm.tri_enum = (tri == GPX_LOWER && M == N && row0 == col0) ? 1 : 0;
    const int64_t pbr = cdiv(tm, 4), pbc = cdiv(tn, 8);
    fm.pbc = (int)pbc;
    const int64_t np = fm.tri_enum ? pbr * (pbr + 1) / 2 : pbr * pbc;
    fm.np = (int)np;
    const int64_t blocks = cdiv(np, 8) * 8 * 32;
    ProfScope prof(PC_GEMM, 2.0 * (double)K * updated_elements(M, N, tri, row0, col0), st);
    hipLaunchKernelGGL((gemm_nt_fast_kernel<T>), dim3((unsigned)blocks), dim3(512), F_SMEM, st, M, N, K,
                       (const T *)A, lda, (const T *)B, ldb, (T *)C, ldc, (T)alpha, tri, row0, col0, fm);
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}

int gemm_nt(int dtype, int64_t M, int64_t N, int64_t K, const void *A, int64_t lda, const void *B,
            int64_t ldb, void *C, int64_t ldc, double alpha, int tri, int64_t row0, int64_t col0,
            hipStream_t st)
{
    if (M <= 0 || N <= 0 || K <= 0) return GPX_OK;
    static const bool no_fast = getenv("GPX_GEMM_NO_FAST") != nullptr;
    const int64_t epk = 128 / (int64_t)esize(dtype), ch = 16 / (int64_t)esize(dtype);
    const bool fast = !no_fast && K % epk == 0 && lda % ch == 0 && ldb % ch == 0 &&
                      ((uintptr_t)A) % 16 == 0 && ((uintptr_t)B) % 16 == 0;
    if (fast) {
        if (dtype == GPX_F64)
            return launch_gemm_nt_fast<double>(M, N, K, A, lda, B, ldb, C, ldc, alpha, tri, row0, col0, st);
        return launch_gemm_nt_fast<float>(M, N, K, A, lda, B, ldb, C, ldc, alpha, tri, row0, col0, st);
    }
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
