// gpx_gemm.hip -- C += alpha * A * B^T on the fp64 / fp32 matrix cores of gfx950.
//
// The one O(N^3) kernel of the GP fit path: the trailing SYRK/GEMM update of the
// blocked Cholesky (replaces the level-3 BLAS inside LAPACK dpotrf reached via
// scipy.linalg.cholesky, gp/gp.py:294), the left-looking panel updates, and the
// TRSM/SYRK updates of the posterior covariance (gp/gp.py:622-625).
//
// Shape: both operands are row-major with the reduction index contiguous ("NT"):
// C[i, j] += alpha * sum_k A[i, k] * B[j, k]  (alpha = -1 in the factorisation).
// A lower Cholesky on row-major storage only ever needs this form.
//
// Roofline: fp64 MFMA (v_mfma_f64_16x16x4_f64: 2048 flop per wave-instruction).
// Algorithmic flops per launch = 2*K per updated element of C (all M*N of them, or
// only those at or below the diagonal in the lower-only SYRK form).
//
// Two kernels.  gemm_nt_kernel (generic shapes: K tail, unaligned operands): workgroup
// = 256 threads = 4 waves (2 x 2), block tile 128 x 128, wave tile 64 x 64 = 4 x 4 MFMA
// tiles (128 accumulator VGPRs in fp64); a k-step is 128 bytes of every operand row
// (16 doubles / 32 floats), staged global -> registers -> LDS with the next k-step's
// loads in flight under the current step's MFMAs, double-buffered LDS, one barrier
// per k-step.  gemm_nt_fast_kernel (everything the factorisation launches): the same
// wave tile fed by LDS-DMA, see the block comment at "Fast path".  In both the MFMA
// reduction index is permuted so that every lane reads 32 contiguous bytes of its row
// from LDS (2 x ds_read_b128) per k-step: lane (i, q) holds k = q*S + s for sub-step s.
#include "gpx_common.h"

namespace gpx {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float  f4_t __attribute__((ext_vector_type(4)));

template <typename T> struct MF;
template <> struct MF<double> {
    typedef d4_t acc_t;
    static constexpr int EPK = 16;   // elements per k-step (128 B per row)
    static constexpr int CH = 2;     // elements per 16-byte chunk
    static constexpr int NR = 4;     // accumulator registers per 16 x 16 tile
    __device__ static __forceinline__ acc_t mfma(double a, double b, acc_t c)
    { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    __device__ static __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct MF<float> {
    typedef f4_t acc_t;
    static constexpr int EPK = 32;
    static constexpr int CH = 4;
    static constexpr int NR = 4;
    __device__ static __forceinline__ acc_t mfma(float a, float b, acc_t c)
    { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
    __device__ static __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

typedef float f16_t __attribute__((ext_vector_type(16)));

// MFMA shape of the FAST kernel.  fp64: v_mfma_f64_16x16x4 (64 cycles).  fp32: v_mfma_f32_32x32x2 (64
// cycles, 4096 flop) instead of 16x16x4 (32 cycles, 2048 flop): the same matrix-pipe rate and the same
// LDS bytes per flop for a 64 x 64 wave tile, but HALF as many MFMA instructions per k-step -- every
// ds_read / LDS-DMA / waitcnt issued between two MFMAs then has 64 cycles to hide in instead of 32.
// TM: tile edge; NR: accumulator registers per tile; lane -> (row within tile, k group): lane & (TM-1), lane / TM;
// KG = 64 / TM k groups; a lane holds EPK / KG consecutive elements of its row per k-step.
template <typename T> struct MFF;
template <> struct MFF<double> : MF<double> {
    static constexpr int TM = 16, NR = 4;
};
template <> struct MFF<float> {
    typedef f16_t acc_t;
    static constexpr int EPK = 32, CH = 4, TM = 32, NR = 16;
    __device__ static __forceinline__ acc_t mfma(float a, float b, acc_t c)
    { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    // C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = 8 * (reg / 4) + 4 * (lane >> 5) + reg % 4
    __device__ static __forceinline__ int row(int lane, int reg) { return 8 * (reg >> 2) + 4 * (lane >> 5) + (reg & 3); }
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

// Epilogue shared by both kernels: C = beta * C + alpha * acc (beta = 1, or 0 when
// `beta0`) for one wave's 64 x (16 * NTW) tile.  The 32 loads of two MFMA tile rows
// are issued before the first use (clamped addresses keep them unconditional), so a
// lane pays 2 memory round trips per tile instead of 64.
template <typename T, int NTW, typename M = MF<T>, int NTJ = NTW, int IBS = 0>
__device__ __forceinline__ void store_wave_tile(typename M::acc_t (&acc)[64 / (M::NR == 4 ? 16 : 32)][NTJ], T *__restrict__ C,
                                                int64_t ldc, int64_t Mr, int64_t N, int64_t r_base,
                                                int64_t c_base, int lane, T alpha, int tri,
                                                int64_t row0, int64_t col0, int beta0)
{
    constexpr int TM = M::NR == 4 ? 16 : 32, NR = M::NR, TI = 64 / TM, IB = IBS ? IBS : (TM == 16 ? 2 : 1);
    const int ccol = lane & (TM - 1);
#pragma unroll
    for (int ib = 0; ib < TI; ib += IB) {
        T cv[IB][NTJ][NR];
#pragma unroll
        for (int ii = 0; ii < IB; ++ii)
#pragma unroll
            for (int j = 0; j < NTJ; ++j) {
                const int64_t gc = min(c_base + j * TM + ccol, N - 1);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int64_t gr = min(r_base + (ib + ii) * TM + M::row(lane, r), Mr - 1);
                    cv[ii][j][r] = beta0 ? (T)0 : C[gr * ldc + gc];
                }
            }
#pragma unroll
        for (int ii = 0; ii < IB; ++ii)
#pragma unroll
            for (int j = 0; j < NTJ; ++j) {
                const int64_t gc = c_base + j * TM + ccol;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int64_t gr = r_base + (ib + ii) * TM + M::row(lane, r);
                    if (gr < Mr && gc < N && !(tri == GPX_LOWER && row0 + gr < col0 + gc))
                        C[gr * ldc + gc] = fma(alpha, acc[ib + ii][j][r], cv[ii][j][r]);
                }
            }
    }
}

// Atomic epilogue: C += alpha * acc as no-return global_atomic_add (executed at the memory
// side, no load round trip).  Every element of C is touched by exactly ONE tile of a launch,
// so the result is the single correctly rounded sum C + alpha*acc, identical to the
// load/fma/store form and independent of scheduling.
__device__ __forceinline__ void atomic_add_nr(double *p, double v)
{
    (void)__builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double *)p, v);
}
__device__ __forceinline__ void atomic_add_nr(float *p, float v)
{
    (void)__builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)p, v);
}
template <typename T, int NTW, typename M = MF<T>, int NTJ = NTW>
__device__ __forceinline__ void store_wave_tile_atomic(typename M::acc_t (&acc)[64 / (M::NR == 4 ? 16 : 32)][NTJ],
                                                       T *__restrict__ C, int64_t ldc, int64_t Mr, int64_t N,
                                                       int64_t r_base, int64_t c_base, int lane, T alpha, int tri,
                                                       int64_t row0, int64_t col0)
{
    constexpr int TM = M::NR == 4 ? 16 : 32, NR = M::NR, TI = 64 / TM;
    const int ccol = lane & (TM - 1);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < NTJ; ++j) {
            const int64_t gc = c_base + j * TM + ccol;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int64_t gr = r_base + i * TM + M::row(lane, r);
                if (gr < Mr && gc < N && !(tri == GPX_LOWER && row0 + gr < col0 + gc))
                    atomic_add_nr(C + gr * ldc + gc, alpha * acc[i][j][r]);
            }
        }
}

// The same for a wave tile that lies wholly inside the matrix (every tile of an aligned update but those on its ragged
// edges): no bounds tests, one lane pointer, the row part of the offset wave-uniform and the column part an immediate, the
// triangle as ONE 32-bit compare of a per-lane constant against a wave-uniform threshold per element -- a handful of
// instructions per atomic instead of seventeen (the general form's 64-bit index arithmetic and exec masking ran on the
// SIMD whose other wave is in its k-loop: in situ the difference is 1.8 % of the N = 65536 step).
// The instruction stream is THE SAME for tiles below the diagonal and tiles on it (those only execute fewer lanes): a first
// version that skipped the compare for interior tiles made their epilogue shorter than the diagonal tiles' general one,
// the workgroups of an XCD fell out of the lock-step in which they share operand slices in L2, and the kernel fetched
// 25 % more alone and 80 % more in situ (FETCH_SIZE 6.56 -> 8.1 GB raw per launch at M = 32768;
// profiles/r06_fetch_probe.log).  Equal work per tile keeps the convoy together.
template <typename T, typename M, int NTJ>
__device__ __forceinline__ void store_wave_tile_atomic_inner(typename M::acc_t (&acc)[64 / (M::NR == 4 ? 16 : 32)][NTJ],
                                                             T *__restrict__ C, int64_t ldc, int64_t r_base,
                                                             int64_t c_base, int lane, T alpha, int tri,
                                                             int64_t row0, int64_t col0)
{
    constexpr int TM = M::NR == 4 ? 16 : 32, NR = M::NR, TI = 64 / TM;
    T *p = C + (r_base + M::row(lane, 0)) * ldc + c_base + (lane & (TM - 1));
    // element (i, r, j) of this lane is at or below the diagonal iff  delta + lane_d + rowoff - coloff >= 0
    const int64_t delta64 = (row0 + r_base) - (col0 + c_base);
    const int delta = tri == GPX_LOWER ? (int)max((int64_t)-4096, min((int64_t)4096, delta64)) : 4096;   // wave-uniform; clamped: |offsets| < 128
    const int lane_d = M::row(lane, 0) - (lane & (TM - 1));
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int rowoff = i * TM + M::row(0, r);
            T *pr = p + (int64_t)rowoff * ldc;
#pragma unroll
            for (int j = 0; j < NTJ; ++j)
                if (lane_d >= j * TM - rowoff - delta) atomic_add_nr(pr + j * TM, alpha * acc[i][j][r]);
        }
}

// exchange a value with the neighbouring lane (lane ^ 1) through DPP quad_perm [1,0,3,2]
__device__ __forceinline__ double swap_pair(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float swap_pair(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}

// Vector epilogue (ldc even, C 2-element aligned, N even): neighbouring lanes hold
// neighbouring columns of the same rows, so after one DPP exchange per register
// pair the even lane owns two columns of rows {q, q+8} and the odd lane two columns
// of rows {q+4, q+12}: every lane moves 2 elements per access and the tile needs half
// the memory instructions.  Elements outside the triangle are written back unchanged.
template <typename T, int NTW, typename M = MF<T>, int NTJ = NTW>
__device__ __forceinline__ void store_wave_tile_v2(typename M::acc_t (&acc)[64 / (M::NR == 4 ? 16 : 32)][NTJ], T *__restrict__ C,
                                                   int64_t ldc, int64_t Mr, int64_t N, int64_t r_base,
                                                   int64_t c_base, int lane, T alpha, int tri,
                                                   int64_t row0, int64_t col0, int beta0)
{
    constexpr int TM = M::NR == 4 ? 16 : 32, NH = M::NR / 2, TI = 64 / TM, IB = TM == 16 ? 2 : 1;
    struct alignas(2 * sizeof(T)) P2 { T x, y; };
    const int odd = lane & 1;
    const int ccol = (lane & (TM - 1)) & ~1;           // first column of the lane pair
#pragma unroll
    for (int ib = 0; ib < TI; ib += IB) {              // IB MFMA tile rows per round trip
        P2 cv[IB][NTJ][NH];
        int64_t grr[IB][NH];
#pragma unroll
        for (int ii = 0; ii < IB; ++ii)
#pragma unroll
            for (int h = 0; h < NH; ++h)               // register pair (2h, 2h + 1): two neighbouring rows
                grr[ii][h] = r_base + (ib + ii) * TM + M::row(lane, 2 * h + odd);
#pragma unroll
        for (int ii = 0; ii < IB; ++ii)
#pragma unroll
            for (int j = 0; j < NTJ; ++j) {
                const int64_t gc = min(c_base + j * TM + ccol, N - 2);
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const int64_t gr = min(grr[ii][h], Mr - 1);
                    if (beta0) { cv[ii][j][h].x = (T)0; cv[ii][j][h].y = (T)0; }
                    else cv[ii][j][h] = *reinterpret_cast<const P2 *>(C + gr * ldc + gc);
                }
            }
#pragma unroll
        for (int ii = 0; ii < IB; ++ii)
#pragma unroll
            for (int j = 0; j < NTJ; ++j) {
                const int64_t gc = c_base + j * TM + ccol;
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    // even lane keeps register 2h (sends 2h+1), odd lane keeps 2h+1 (sends 2h)
                    const T keep = odd ? acc[ib + ii][j][2 * h + 1] : acc[ib + ii][j][2 * h];
                    const T send = odd ? acc[ib + ii][j][2 * h] : acc[ib + ii][j][2 * h + 1];
                    const T recv = swap_pair(send);
                    const T vx = odd ? recv : keep;    // column gc
                    const T vy = odd ? keep : recv;    // column gc + 1
                    const int64_t gr = grr[ii][h];
                    if (gr < Mr && gc + 1 < N) {
                        P2 o = cv[ii][j][h];
                        if (!(tri == GPX_LOWER && row0 + gr < col0 + gc)) o.x = fma(alpha, vx, o.x);
                        if (!(tri == GPX_LOWER && row0 + gr < col0 + gc + 1)) o.y = fma(alpha, vy, o.y);
                        if (!(tri == GPX_LOWER && row0 + gr < col0 + gc))     // at least column gc is inside
                            *reinterpret_cast<P2 *>(C + gr * ldc + gc) = o;
                    }
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
                                                         int tri, int64_t row0, int64_t col0, Batch bt)
{
    typedef typename MF<T>::acc_t acc_t;
    constexpr int EPK = MF<T>::EPK;
    constexpr int CH = MF<T>::CH;
    constexpr int SUB = EPK / 4;        // MFMA sub-steps per k-step; also elements per lane per row

    A += (int64_t)blockIdx.z * bt.sA; B += (int64_t)blockIdx.z * bt.sB; C += (int64_t)blockIdx.z * bt.sC;
    const int64_t bm0 = (int64_t)blockIdx.y * GB_M;
    const int64_t bn0 = (int64_t)blockIdx.x * GB_N;
    if (tri == GPX_LOWER && col0 + bn0 > row0 + bm0 + GB_M - 1) return;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar registers
    const int wr = wave >> 1, wc = wave & 1;

    // global -> LDS staging map: 8 threads cover one 128-B row, 32 rows per pass
    const int s_row = tid >> 3;          // 0..31
    const int s_chk = tid & 7;           // 16-B chunk within the row
    // ONE set of four staging registers per thread: the next k-step's A rows travel under the first half of this
    // step's MFMAs, its B rows under the second half (round 6: both at once were 32 registers beside the accumulators)
    Chunk16 rs[4];

    auto load_op = [&](int op, int64_t kbase) {
        const int64_t k = kbase + (int64_t)s_chk * CH;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            rs[p] = op == 0 ? load_chunk<T>(A, lda, bm0 + s_row + 32 * p, M, k, K)
                            : load_chunk<T>(B, ldb, bn0 + s_row + 32 * p, N, k, K);
    };
    auto store_op = [&](int op, int buf) {
        unsigned char *t = smem + (size_t)buf * 2 * G_TILE_BYTES + (size_t)op * G_TILE_BYTES;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            *reinterpret_cast<Chunk16 *>(t + (s_row + 32 * p) * G_PITCH + s_chk * 16) = rs[p];
    };

    acc_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = (T)0;

    const int nk = (int)((K + EPK - 1) / EPK);
    load_op(0, 0); store_op(0, 0);
    load_op(1, 0); store_op(1, 0);
    __syncthreads();

    // per-lane fragment address: row (l & 15) of the MFMA tile, bytes [32*q, 32*q + 32)
    const int f_off = (lane & 15) * G_PITCH + (lane >> 4) * 32;

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;

        const unsigned char *ta = smem + (size_t)cur * 2 * G_TILE_BYTES;
        const unsigned char *tb = ta + G_TILE_BYTES;
        // fragments in two halves of 16 bytes per row (round 6: all 32 bytes of the 8 rows at once were 64 live registers
        // beside the 128 accumulators and the 32 staging registers -- 34 VGPRs spilled in fp64)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (more) load_op(hf, (int64_t)(kt + 1) * EPK);
            T fa[4][SUB / 2], fb[4][SUB / 2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned char *pa = ta + (wr * 64 + i * 16) * G_PITCH + f_off + hf * 16;
                const unsigned char *pb = tb + (wc * 64 + i * 16) * G_PITCH + f_off + hf * 16;
                Chunk16 a0 = *reinterpret_cast<const Chunk16 *>(pa);
                Chunk16 b0 = *reinterpret_cast<const Chunk16 *>(pb);
                memcpy(&fa[i][0], &a0, 16);
                memcpy(&fb[i][0], &b0, 16);
            }
#pragma unroll
            for (int s = 0; s < SUB / 2; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = MF<T>::mfma(fa[i][s], fb[j][s], acc[i][j]);
            if (more) store_op(hf, cur ^ 1);         // the other buffer: nobody reads it during this k-step
            __builtin_amdgcn_sched_barrier(0);       // keep the second half's reads behind the first half's MFMAs
        }
        __syncthreads();
    }

    // epilogue: C += alpha * acc  (16 lanes = 16 consecutive columns = one 128-B / 64-B segment)
    // (one MFMA tile row per round trip: 16 loads in flight beside the 128 accumulators.  The lane index is taken again
    // behind an opaque asm: the epilogue's per-lane indices were computed before the k-loop and spilled across it)
    int lane_e = (int)(threadIdx.x & 63);
    asm volatile("" : "+v"(lane_e));
    store_wave_tile<T, 4, MF<T>, 4, 1>(acc, C, ldc, M, N, bm0 + wr * 64, bn0 + wc * 64, lane_e, alpha, tri, row0, col0, 0);
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
                   int64_t col0, hipStream_t st, const Batch &bt)
{
    if (M <= 0 || N <= 0 || K <= 0) return GPX_OK;
    GPX_TRY(set_max_lds((const void *)gemm_nt_kernel<T>, G_SMEM));
    dim3 grid((unsigned)cdiv(N, GB_N), (unsigned)cdiv(M, GB_M), (unsigned)bt.count), block(256);
    ProfScope prof(PC_GEMM_GENERIC, 2.0 * (double)K * updated_elements(M, N, tri, row0, col0) * bt.count, st);
    hipLaunchKernelGGL((gemm_nt_kernel<T>), grid, block, G_SMEM, st, M, N, K, (const T *)A, lda,
                       (const T *)B, ldb, (T *)C, ldc, (T)alpha, tri, row0, col0, bt);
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}


// ===========================================================================
// Fast path: K % (128 B) == 0, 16-B aligned operands.
//   wave tile 64 x 64 (4 x 4 MFMA tiles); block tile 128 x 128 (128 x 64 for the skinny panel products and
//   short updates), 4 waves (2 x 2), 2 LDS stages of 32 KiB, TWO workgroups per CU -- one's prologue,
//   epilogue and barrier stalls fall under the other's k-loop;
//   operands go HBM/L2 -> LDS directly (global_load_lds_dwordx4, no VGPR staging), one
//   raw s_barrier per k-step;
//   LDS rows are unpadded (the DMA writes 1 KiB = 8 rows contiguously), bank conflicts
//   of the ds_read_b128 fragment reads are removed by an XOR swizzle applied to the
//   per-lane GLOBAL source chunk and again on the read;
//   tiles are walked in 1024 x 1024 patches (8 x 8 tiles of 128 x 128) and patch p is
//   given to the blocks with blockIdx % 8 == p % 8, so that the 64 workgroups that run
//   together on one XCD (2 per CU) are one patch and share 8 A- and 8 B-slices in that
//   XCD's L2.
// ===========================================================================
constexpr int F_BM = 128;                           // tile rows of the fast kernel (the 256 x 128 / 8-wave / 3-stage /
                                                    // one-workgroup-per-CU form of rounds 1 - 5 was measured slower with
                                                    // every schedule, last in profiles/r06_gemm_bm256_dropped.log, and removed)
template <int BN> struct FGeo {
    static constexpr int NW = F_BM / 32;            // waves: 2 x 2
    static constexpr int NST = 2;                   // LDS stages
    static constexpr int ROWS = F_BM + BN;          // tile rows per stage: 256 / 192
    static constexpr int STAGE = ROWS * G_ROWB;     // 32,768 / 24,576 B
    static constexpr int SMEM = NST * STAGE;        // 65,536 / 49,152 B
    static constexpr int PW = ROWS / (8 * NW);      // DMA pieces (1 KiB) per wave and stage: 8 / 6
    static constexpr int NTW = BN / 32;             // 16-column MFMA tiles per wave: 4 / 2
};
template <int N> struct WaitVm;
#define GPX_WAITVM(N) template <> struct WaitVm<N> { static __device__ __forceinline__ void go() { asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); } }
GPX_WAITVM(0); GPX_WAITVM(5); GPX_WAITVM(6); GPX_WAITVM(8); GPX_WAITVM(10); GPX_WAITVM(12);
#undef GPX_WAITVM

// chunk swizzle of tile row r: a permutation of (r >> 1) & 7 chosen so that every
// 16-lane group of a ds_read_b128 fragment read hits 16 distinct 16-B slots
__device__ __forceinline__ int f_swz(int r) { return (0x64753120u >> (4 * ((r >> 1) & 7))) & 7; }

// Tile map of one launch.
//  * columns: local column c of C takes its B-operand row from
//        c + boff + ((cbase + c) / nb) * pm1nb
//    and sits at "global" column c + col0 + ((cbase + c) / nb) * pm1nb for the
//    triangle test.  Plain GEMM: boff = pm1nb = 0.  1-D block-cyclic trailing
//    update on rank r of P: pm1nb = (P - 1) * nb, so that consecutive local block
//    columns map to global block columns P apart.
//  * enumeration: 1024 x 1024 patches, column-major staircase -- patch column pc
//    holds patch rows [b + a * pc, PBR); R = PBR - b; np patches in total.
constexpr int G_MAX_BANDS = 192;
struct GemmMap {
    int64_t boff, cbase, nb, pm1nb, brows;
    int np, a, b, R;
    // diagonal split (exactly aligned lower-triangular maps): blocks >= dbegin enumerate
    // ONLY the 36 tiles at or below the diagonal of the ndiag diagonal patches (patch column pc,
    // patch row bdiag + pc); the staircase above then starts one patch below the diagonal.  Without
    // it 28 of the 64 workgroups of every diagonal patch return at once -- and still pass, in order,
    // through the dispatcher in front of real tiles.
    int dbegin, ndiag, bdiag;
    // ktri: both operands are UPPER triangular in their own (row, k) index space (X = L^-T): the
    // product of tile rows bm0.. needs only k >= bm0, the k-loop starts there (K = M = N products)
    int ktri;
    int csh;      // log2 of the tile columns per patch (3: 8 x BN = 1024 columns; fewer for narrow products,
                  // so that no workgroup is launched only to find its tile outside the matrix)
    // diagnostic: when non-null, wave 0 of every workgroup stores 4 s_memtime stamps
    // (start, first barrier passed, k-loop done, epilogue done) 2 s_memrealtime stamps and the hardware ids at stamps[8 * blockIdx]
    unsigned long long *stamps;
    int vec_c;    // C allows 2-element vector accesses (ldc even, aligned base, N even)
    int atomic_c; // epilogue by no-return atomic adds (GPX_GEMM_ATOMIC_C)
    // the factorisation's `info`: once a pivot has failed the factor is garbage whatever the remaining
    // updates do, so every workgroup of a trailing update returns at once (a non-PD theta in an ML-II
    // sweep then costs the launches, not the flops)
    const int *abort_flag;
    // batched launches (blockIdx.y = matrix index): element strides between consecutive matrices
    int64_t sA, sB, sC;
    int64_t tA, tB, tC;   // second batch dimension (blockIdx.z)
    int sflag;    // stride of abort_flag (one info word per matrix)
    // EXACT enumeration (lower-triangular trailing updates whose triangle is tile aligned): the
    // grid holds only tiles that exist -- tile (i, j) with j <= min(eTC - 1, i + eD), i < eTR.  Order: bands
    // of 8 tile rows, column-major inside a band, so that 64 consecutive tiles are an 8 x 8 patch (8 A- and
    // 8 B-slices); chunks of 2^ecl consecutive tiles go to one XCD (chunk c -> blocks with blockIdx % 8 ==
    // c % 8).  epre[b] = tiles before band b.  With the 1024 x 1024 patch grid a trailing update of a
    // small matrix launched up to 6 x more workgroups than it has tiles (n = 8192, nb = 256: 800 for the
    // 124 tiles of a block-column update), each still passing through the dispatcher.
    int exact, ecl, eT, eD, eTR, eTC, ebands;
    // block-cyclic ranks (P > 1): local tile column j is GLOBAL tile column j + (j / etpb) * epm1t (etpb = nb / 128
    // tile columns per block column, epm1t = (P - 1) * etpb); the triangle is tested in global columns
    int etpb, epm1t;
    int ecs;           // 1: tile columns are 64 wide (BN = 64; etpb, epm1t, eTC and j count 64-column units, eD / rows stay in 128s)
    // up to G_MAX_BANDS bands of 1024 rows: trailing updates of up to 196,608 rows, beyond what one GPU's HBM holds in fp64
    // (round 6: 64 bands until then, so that above N = 65536 the first updates fell back to the patch map)
    int epre[G_MAX_BANDS + 2];
};
static unsigned long long *g_gemm_stamps = nullptr;

typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
#define GPX_DSR(dst, addr, off) \
    asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
// The wait that belongs to those reads.  The compiler takes an asm's output for written when the statement has executed; a
// ds_read's destination is written when the data RETURNS.  Every fragment register therefore passes through the wait as an
// in/out operand: nothing can read, copy or -- once the fragments are dead, after the k-loop -- REUSE such a register
// before the wait.  (Round 6, found by the soak test beside a neighbour: the plain wait after the loop came behind the
// compiler's copies of the accumulators for the epilogue, one of which went to a register of the last, unused fragment
// read; about one fp32 fit in a thousand at n = 8192 got the late LDS data in the last accumulator of a wave instead --
// rows 59 / 63 of a 64-row wave tile.  tools/r6_soak_probe.py, profiles/r06_soak_probe_*.log.)
#define GPX_FRAG_WAIT(CNT, RA, RB, NB4)                                                                        \
    do {                                                                                                       \
        if (NB4)                                                                                               \
            asm volatile("s_waitcnt " CNT : "+v"(RA[0]), "+v"(RA[1]), "+v"(RA[2]), "+v"(RA[3]), "+v"(RB[0]), "+v"(RB[1]), \
                                            "+v"(RB[2]), "+v"(RB[3]) : : "memory");                             \
        else                                                                                                   \
            asm volatile("s_waitcnt " CNT : "+v"(RA[0]), "+v"(RA[1]), "+v"(RA[2]), "+v"(RA[3]), "+v"(RB[0]), "+v"(RB[1]) \
                         : : "memory");                                                                        \
    } while (0)

// TAG only changes the symbol name: 1 = the block-cyclic trailing update of the
// factorisation (gpx_d_syrk_bc), so that profilers list the dominant kernel separately
// from the panel / covariance products that share its code.
template <typename T, int BN, int TAG, int ABL = 0>
__global__ __launch_bounds__(F_BM * 2, 2) void gemm_nt_fast_kernel(int64_t M, int64_t N, int64_t K,
                                                              const T *__restrict__ A, int64_t lda,
                                                              const T *__restrict__ B, int64_t ldb,
                                                              T *__restrict__ C, int64_t ldc, T alpha,
                                                              int tri, int64_t row0, int64_t col0,
                                                              GemmMap fm, int beta0)
{
    typedef MFF<T> MM;                                   // MFMA shape of this kernel (fp64 16x16x4, fp32 32x32x2)
    typedef typename MM::acc_t acc_t;
    constexpr int EPK = MM::EPK;
    constexpr int TM = MM::TM;                           // MFMA tile edge
    constexpr int TI = 64 / TM, TJ = (BN / 2) / TM;      // MFMA tiles of the 64 x (BN / 2) wave tile
    constexpr int KG = 64 / TM;                          // k groups over the lanes
    constexpr int SUB = EPK / KG;                        // MFMA sub-steps per k-step = elements per lane and row
    constexpr int CPL = SUB * (int)sizeof(T) / 16;       // 16-byte chunks per lane and row per k-step: 2 / 4
    constexpr int UH = CPL / 2;                          // chunks per lane, row and HALF k-step: 1 / 2
    typedef FGeo<BN> Geo;
    constexpr int F_STAGE = Geo::STAGE, PW = Geo::PW, NTW = Geo::NTW, F_NST = Geo::NST;
    (void)NTW;
    // requested now, looked at after the prologue's DMA is under way (its latency hides there)
    const int aborted = fm.abort_flag ? fm.abort_flag[(int64_t)blockIdx.y * fm.sflag] : 0;
    A += (int64_t)blockIdx.y * fm.sA + (int64_t)blockIdx.z * fm.tA;
    B += (int64_t)blockIdx.y * fm.sB + (int64_t)blockIdx.z * fm.tB;
    C += (int64_t)blockIdx.y * fm.sC + (int64_t)blockIdx.z * fm.tC;
    const int bid = blockIdx.x;
    // hardware XCD = blockIdx.x % 8 (the grid's x extent is a multiple of 8).  In a batched launch every
    // matrix would send its first patch / chunk to the same XCD -- a batch of small products (one patch each)
    // then runs on an eighth of the chip -- so the logical XCD is rotated by the matrix index.
    const int xcd = (bid - (int)blockIdx.y - 3 * (int)blockIdx.z) & 7, loc = bid >> 3;
    constexpr int RSH = 3;                                          // log2 of the tile rows per patch
    const int tsh = RSH + fm.csh;
    int patch = (loc >> tsh) * 8 + xcd, within = loc & ((1 << tsh) - 1);
    int pb_r, pb_c;
    if (fm.exact) {
        const int chunk = ((loc >> fm.ecl) << 3) + xcd;
        const int t = (chunk << fm.ecl) + (loc & ((1 << fm.ecl) - 1));
        if (t >= fm.eT) return;
        int lo = 0, hi = fm.ebands;                                   // band: epre[lo] <= t < epre[lo + 1]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (fm.epre[mid] <= t) lo = mid; else hi = mid; }
        int tt = t - fm.epre[lo];
        const int h = min(8, fm.eTR - 8 * lo);                        // tile rows in this band
        const int dj = 8 * lo + fm.eD;                                // column j holds rows r >= g(j) - dj
        // columns that hold all h rows: local columns whose global column is <= dj
        int jfull = 0;
        if (dj >= 0) {
            const int djc = (dj + 1) << fm.ecs;                       // tile columns (of BN) up to the band's first row tile
            const int period = fm.etpb + fm.epm1t, qd = djc / period, rem = djc - qd * period;
            jfull = min(fm.eTC, qd * fm.etpb + min(rem, fm.etpb));
        }
        int j, r;
        if (tt < jfull * h) { j = tt / h; r = tt - j * h; }
        else {
            tt -= jfull * h;
            j = jfull;
            for (;;) {                                                // at most 8 (16 with 64-wide columns) partial columns
                const int gj = j + (j / fm.etpb) * fm.epm1t;
                const int rmin = max(0, (gj >> fm.ecs) - dj), cnt = max(0, h - rmin);
                if (tt < cnt) { r = rmin + tt; break; }
                tt -= cnt; ++j;
            }
        }
        pb_r = lo; pb_c = j >> 3; within = (r << 3) | (j & 7);
    } else if (bid >= fm.dbegin) {
        const int b2 = bid - fm.dbegin;
        const int loc2 = b2 >> 3, dp = (loc2 / 36) * 8 + (b2 & 7), t = loc2 % 36;
        if (dp >= fm.ndiag) return;
        int r = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);     // t = r (r + 1) / 2 + c, c <= r
        if ((r + 1) * (r + 2) / 2 <= t) ++r;
        if (r * (r + 1) / 2 > t) --r;
        pb_c = dp; pb_r = fm.bdiag + dp; within = r * 8 + (t - r * (r + 1) / 2);
    } else if (patch >= fm.np) {
        return;
    } else if (fm.a == 0) {
        pb_c = patch / fm.R; pb_r = fm.b + (patch - pb_c * fm.R);
    } else {
        // patches before column pc: cum(pc) = pc * R - a * pc * (pc - 1) / 2
        auto cum = [&](int pc) { return pc * fm.R - fm.a * (pc * (pc - 1) / 2); };
        const double rh = (double)fm.R + 0.5 * fm.a;
        const double disc = rh * rh - 2.0 * fm.a * (double)patch;
        int pc = (int)((rh - sqrt(disc > 0.0 ? disc : 0.0)) / fm.a);
        if (pc < 0) pc = 0;
        while (cum(pc + 1) <= patch) ++pc;
        while (pc > 0 && cum(pc) > patch) --pc;
        pb_c = pc; pb_r = fm.b + fm.a * pc + (patch - cum(pc));
    }
    const int64_t bm0 = ((int64_t)pb_r * (1024 / F_BM) + (within >> fm.csh)) * F_BM;
    const int64_t bn0 = (((int64_t)pb_c << fm.csh) + (within & ((1 << fm.csh) - 1))) * BN;
    if (bm0 >= M || bn0 >= N) return;
    const int64_t cshift = ((fm.cbase + bn0) / fm.nb) * fm.pm1nb;   // same for the tile's 128 columns
    col0 += cshift;
    if (tri == GPX_LOWER && col0 + bn0 > row0 + bm0 + F_BM - 1) return;
    const int64_t brow0 = bn0 + fm.boff + cshift;                   // B-operand row of tile column 0

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // panel products (TAG 0) run beside the trailing update (TAG 1) of the other stream and are on
    // the critical path of the next step: let their waves win the instruction arbiter
    if (TAG == 0) __builtin_amdgcn_s_setprio(2);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: keep it scalar
    const int wr = wave >> 1, wc = wave & 1;

    unsigned long long st0 = 0, st1 = 0, st2 = 0, rt0 = 0;
    if (fm.stamps) { st0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }

    // ktri: rows >= bm0 of an upper-triangular operand are zero for k < bm0 (bm0 is a multiple of EPK)
    const int kskip = fm.ktri == 1 ? (int)(min(bm0, K - EPK) / EPK) : 0;
    // ---- DMA source pointers: wave w owns pieces PW*w .. PW*w + PW-1 (1 KiB = 8 rows each) of every stage ----
    const unsigned char *gsrc[PW];      // next 128-B k-slice to fetch, per DMA piece
    {
        const int r_in = lane >> 3, c = lane & 7;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int tr = 8 * (wave * PW + j) + r_in;         // tile row 0 .. ROWS-1
            const int g = c ^ f_swz(tr);                       // global 16-B chunk held by LDS chunk c
            if (tr < F_BM) {
                const int64_t r = min(bm0 + tr, M - 1);
                gsrc[j] = reinterpret_cast<const unsigned char *>(A + r * lda) + g * 16 + kskip * (int64_t)G_ROWB;
            } else {
                const int64_t r = min(brow0 + (tr - F_BM), fm.brows - 1);
                gsrc[j] = reinterpret_cast<const unsigned char *>(B + r * ldb) + g * 16 + kskip * (int64_t)G_ROWB;
            }
        }
    }
    acc_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < MM::NR; ++r) acc[i][j][r] = (T)0;

    // ktri == 2: the B operand is LOWER triangular in (row, k) (W = inv(L)): tile columns [bn0, bn0 + BN)
    // only have k < bn0 + BN, the k-loop ends there
    const int nk = (int)((fm.ktri == 2 ? min(K, bn0 + (int64_t)BN) : K) / EPK) - kskip;
    // three stages in flight before the first wait; when K has fewer than three slices the
    // extra stages re-fetch the last slice (never read)
    {
        unsigned char *dst0 = smem + (wave * PW) * 1024;
#pragma unroll
        for (int st3 = 0; st3 < F_NST; ++st3) {
            const int inc = (st3 + 1 < nk) ? G_ROWB : 0;
#pragma unroll
            for (int j = 0; j < PW; ++j) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc[j],
                                                 (__attribute__((address_space(3))) void *)(dst0 + st3 * F_STAGE + j * 1024),
                                                 16, 0, 0);
                gsrc[j] += inc;
            }
        }
    }

    // Fragment reads are inline asm: hipcc cannot prove that the in-flight LDS-DMA
    // writes (other stages) do not alias them and would otherwise drain vmcnt(0)
    // before every k-step's first ds_read.  Ordering is by hand (counted vmcnt /
    // lgkmcnt + raw barrier + sched_barrier).
    // per-lane read address: row (lane % TM) of an MFMA tile, chunks q * CPL .. q * CPL + CPL - 1 swizzled
    // (the swizzle of a tile row only depends on (row >> 1) & 7, and tiles start at multiples of 16 rows)
    const int q = lane / TM;
    const int fl = f_swz(lane & (TM - 1));
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    unsigned rdc[CPL];                                   // byte offset of chunk c of this lane's row
#pragma unroll
    for (int c = 0; c < CPL; ++c) rdc[c] = (unsigned)((lane & (TM - 1)) * G_ROWB + ((q * CPL + c) ^ fl) * 16);
    const unsigned a_base = lds0 + (unsigned)((wr * 64) * G_ROWB);
    const unsigned b_base = lds0 + (unsigned)((F_BM + wc * (BN / 2)) * G_ROWB);

    // Schedule of one k-step.  R0 = first-half fragments (MFMA sub-steps 0..H-1), R1 = second half; each
    // half is 4 A and up to 4 B fragment registers of 16 bytes (read slot v of operand A / B = tile v / UH,
    // chunk v % UH of the half).  Every memory instruction is issued BETWEEN groups of MFMAs worth >= 128
    // matrix-pipe cycles, so that its issue cost (an LDS-DMA costs the wave ~60 cycles) falls into the
    // cycles the matrix pipe needs for the group before it:
    //   wait R0 | { MFMA group(R0) ; ds_read R1[g] } | wait R1 | wait stage kt+1 | barrier |
    //           { MFMA group(R1) ; DMA piece g of stage kt+NST -> buffer kt % NST ; ds_read R0(kt+1)[g] }
    constexpr int H = SUB / 2;
    u4_t r0a[4], r0b[4], r1a[4], r1b[4];
    constexpr bool NB4 = TM == 16 ? TJ > 2 : TJ > 1;          // the B fragments a half reads: four (else two, GPX_SLOT_READ)
    // one piece of the next un-fetched k-slice.  Issued unconditionally: past the last slice
    // the increment is 0, so the tail re-fetches the final slice into a buffer nobody reads
    // any more (no branch in the k-loop, always the same vmcnt bookkeeping).
    auto issue1 = [&](int stage, int j, int inc) {
        unsigned char *dst = smem + stage * F_STAGE + (wave * PW + j) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc[j],
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        gsrc[j] += inc;
    };
    // read slot g of a half: g = 2 v + operand (A even, B odd), v = 0..3; AA / AB: per-u address registers
    // (base + stage + rdc[half * UH + u]); the tile offset is an immediate
#define GPX_SLOT_READ(g, RA, RB, AA, AB)                                                        \
    do {                                                                                        \
        if (TM == 16) {                                                                         \
            switch (g) {                                                                        \
            case 0: GPX_DSR(RA[0], AA[0], 0); break;    case 1: GPX_DSR(RB[0], AB[0], 0); break;    \
            case 2: GPX_DSR(RA[1], AA[0], 2048); break; case 3: GPX_DSR(RB[1], AB[0], 2048); break; \
            case 4: GPX_DSR(RA[2], AA[0], 4096); break;                                         \
            case 5: if (TJ > 2) GPX_DSR(RB[2], AB[0], 4096); break;                             \
            case 6: GPX_DSR(RA[3], AA[0], 6144); break;                                         \
            case 7: if (TJ > 2) GPX_DSR(RB[3], AB[0], 6144); break;                             \
            default: break;                                                                     \
            }                                                                                   \
        } else {                                                                                \
            switch (g) {                                                                        \
            case 0: GPX_DSR(RA[0], AA[0], 0); break;    case 1: GPX_DSR(RB[0], AB[0], 0); break;    \
            case 2: GPX_DSR(RA[1], AA[UH - 1], 0); break; case 3: GPX_DSR(RB[1], AB[UH - 1], 0); break; \
            case 4: GPX_DSR(RA[2], AA[0], 4096); break;                                         \
            case 5: if (TJ > 1) GPX_DSR(RB[2], AB[0], 4096); break;                             \
            case 6: GPX_DSR(RA[3], AA[UH - 1], 4096); break;                                    \
            case 7: if (TJ > 1) GPX_DSR(RB[3], AB[UH - 1], 4096); break;                        \
            default: break;                                                                     \
            }                                                                                   \
        }                                                                                       \
    } while (0)

    // stage 0 landed (this wave's pieces), then everybody's
    WaitVm<(F_NST - 1) * PW>::go();
    __builtin_amdgcn_s_barrier();
    if (aborted) {                                   // uniform over the grid: see GemmMap::abort_flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    if (fm.stamps) st1 = __builtin_amdgcn_s_memtime();
    {
        unsigned aa0[UH], ab0[UH];
#pragma unroll
        for (int u = 0; u < UH; ++u) { aa0[u] = a_base + rdc[u]; ab0[u] = b_base + rdc[u]; }
#pragma unroll
        for (int g = 0; g < 8; ++g) GPX_SLOT_READ(g, r0a, r0b, aa0, ab0);
    }
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned so = (unsigned)(stage * F_STAGE);
        int nstage = stage + 1; if (nstage >= F_NST) nstage = 0;
        const unsigned sn = (unsigned)(nstage * F_STAGE);
        unsigned aa1[UH], ab1[UH], aa0[UH], ab0[UH];
#pragma unroll
        for (int u = 0; u < UH; ++u) {
            aa1[u] = a_base + so + rdc[UH + u]; ab1[u] = b_base + so + rdc[UH + u];     // second half of this stage
            aa0[u] = a_base + sn + rdc[u];      ab0[u] = b_base + sn + rdc[u];          // first half of the next
        }
        const int inc = (kt + F_NST + 1 < nk) ? G_ROWB : 0;   // slice kt+NST is fetched now; is there one more?

        GPX_FRAG_WAIT("lgkmcnt(0)", r0a, r0b, NB4);             // R0 of this stage is in
        __builtin_amdgcn_sched_barrier(0);
        {
            T ha[TI][H], hb[TJ][H];
#pragma unroll
            for (int i = 0; i < TI; ++i) memcpy(&ha[i][0], &r0a[i * UH], 16 * UH);
#pragma unroll
            for (int j = 0; j < TJ; ++j) memcpy(&hb[j][0], &r0b[j * UH], 16 * UH);
#pragma unroll
            for (int ss = 0; ss < H; ++ss)
#pragma unroll
                for (int i = 0; i < TI; ++i) {
#pragma unroll
                    for (int j = 0; j < TJ; ++j) acc[i][j] = MM::mfma(ha[i][ss], hb[j][ss], acc[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                    const int g = ss * TI + i;               // fp64: 8 slots of 4 MFMAs; fp32: 16 slots of 2 MFMAs
                    // the reads sit in the FIRST half of the slots: the last fragment is requested >= 1024 matrix-pipe
                    // cycles before the wait that needs it (round 6; one read per slot put the last one right in
                    // front of that wait, and a workgroup alone on its CU -- its neighbour in its prologue or
                    // epilogue -- paid the LDS latency twice per k-step: 0.904 -> 0.925 of peak at M = 32768, K = 1024)
                    if (!(ABL & 4)) {
                        if (TM == 16) { if (g < 4) { GPX_SLOT_READ(2 * g, r1a, r1b, aa1, ab1); GPX_SLOT_READ(2 * g + 1, r1a, r1b, aa1, ab1); } }
                        else if (g < 8) GPX_SLOT_READ(g, r1a, r1b, aa1, ab1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        GPX_FRAG_WAIT("lgkmcnt(0)", r1a, r1b, NB4);             // R1 in: this wave is done reading the stage
        if (!(ABL & 1)) {
            WaitVm<(F_NST - 2) * PW>::go();          // slice kt+1 landed (this wave's pieces), then everybody's
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            T ha[TI][H], hb[TJ][H];
#pragma unroll
            for (int i = 0; i < TI; ++i) memcpy(&ha[i][0], &r1a[i * UH], 16 * UH);
#pragma unroll
            for (int j = 0; j < TJ; ++j) memcpy(&hb[j][0], &r1b[j * UH], 16 * UH);
#pragma unroll
            for (int ss = 0; ss < H; ++ss)
#pragma unroll
                for (int i = 0; i < TI; ++i) {
#pragma unroll
                    for (int j = 0; j < TJ; ++j) acc[i][j] = MM::mfma(ha[i][ss], hb[j][ss], acc[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                    const int g = ss * TI + i;
                    if (TM == 16) {
                        if (g < PW && !(ABL & 2)) issue1(stage, g, inc);            // into the buffer just consumed
                        if (g < 4 && !(ABL & 4)) { GPX_SLOT_READ(2 * g, r0a, r0b, aa0, ab0); GPX_SLOT_READ(2 * g + 1, r0a, r0b, aa0, ab0); }
                    } else {                                                         // 16 slots: DMA pieces on the odd ones
                        if (g < 8 && !(ABL & 4)) GPX_SLOT_READ(g, r0a, r0b, aa0, ab0);
                        if ((g & 1) && g / 2 < PW && !(ABL & 2)) issue1(stage, g / 2, inc);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        stage = nstage;
    }
#undef GPX_SLOT_READ
    // tail DMA / reads must not outlive the tile -- nor land in a register the epilogue has been given meanwhile
    GPX_FRAG_WAIT("vmcnt(0) lgkmcnt(0)", r0a, r0b, NB4);

    if (fm.stamps) { __builtin_amdgcn_s_barrier(); st2 = __builtin_amdgcn_s_memtime(); }   // all waves done
    if (ABL & 8) {                             // timing only: no epilogue (one element keeps the accumulators alive)
        T sum = (T)0;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < MM::NR; ++r) sum += acc[i][j][r];
        if (sum == (T)123.456) C[0] = sum;
    } else
    if (!(ABL & 16) && fm.atomic_c && !beta0 && bm0 + F_BM <= M && bn0 + BN <= N)
        store_wave_tile_atomic_inner<T, MM, TJ>(acc, C, ldc, bm0 + wr * 64, bn0 + wc * (BN / 2), lane, alpha, tri, row0, col0);
    else if (fm.atomic_c && !beta0)
        store_wave_tile_atomic<T, NTW, MM, TJ>(acc, C, ldc, M, N, bm0 + wr * 64, bn0 + wc * (BN / 2), lane, alpha, tri, row0,
                                       col0);
    else if (fm.vec_c)
        store_wave_tile_v2<T, NTW, MM, TJ>(acc, C, ldc, M, N, bm0 + wr * 64, bn0 + wc * (BN / 2), lane, alpha, tri, row0,
                                   col0, beta0);
    else
        store_wave_tile<T, NTW, MM, TJ>(acc, C, ldc, M, N, bm0 + wr * 64, bn0 + wc * (BN / 2), lane, alpha, tri, row0,
                                col0, beta0);
    if (fm.stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st3 = __builtin_amdgcn_s_memtime();
        unsigned long long *o = fm.stamps + 8 * (size_t)bid;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
        o[4] = rt0; o[5] = __builtin_amdgcn_s_memrealtime();      // 100 MHz reference: shader clock = d(memtime) / d(realtime)
        // where it ran: HW_REG_HW_ID (cu / sh / se ...) and HW_REG_XCC_ID
        o[6] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
               (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}
#undef GPX_DSR

// ABL: timing-only ablations of the trailing-update kernel (GPX_GEMM_ABLATE, wrong results on purpose; compile-time so
// that the product kernel carries no branch for them): 1 no barrier / vmcnt, 2 no DMA in the loop, 4 no LDS reads in
// the loop, 8 no epilogue
template <typename T, int BN = 128, int TAG = 0, int ABL = 0>
static int launch_gemm_nt_fast(int64_t M, int64_t N, int64_t K, const void *A, int64_t lda,
                               const void *B, int64_t ldb, void *C, int64_t ldc, double alpha, int tri,
                               int64_t row0, int64_t col0, hipStream_t st, const GemmMap *map = nullptr,
                               double work = -1.0, int beta0 = 0, int ktri = 0, const Batch *bt = nullptr)
{
    constexpr int F_SMEM = FGeo<BN>::SMEM;
    const int pad_lds = TAG == 1 ? (int)tune().gemm_pad_lds : 0;
    GPX_TRY(set_max_lds((const void *)gemm_nt_fast_kernel<T, BN, TAG, ABL>, TAG == 1 ? 160 * 1024 : F_SMEM));
    GemmMap fm{};
    if (map) {
        fm = *map;
    } else {
        fm.boff = 0; fm.cbase = 0; fm.nb = (int64_t)1 << 40; fm.pm1nb = 0; fm.brows = N;
        fm.abort_flag = nullptr; fm.sflag = 0; fm.exact = 0;
        fm.a = 0; fm.b = 0; fm.csh = 3;
        // lower-triangular result: skip the patch rows above the diagonal; with more than one patch
        // column the staircase (a = 1) drops one more patch row per column
        if (BN == 128 && tri == GPX_LOWER && col0 >= row0) { fm.a = N > 1024 ? 1 : 0; fm.b = (int)((col0 - row0) / 1024); }
        if (fm.a == 0) while (fm.csh > 0 && ((int64_t)BN << (fm.csh - 1)) >= N) --fm.csh;   // narrow product: narrow patches
        const int64_t pbr = cdiv(M, 1024), pbc = cdiv(N, (int64_t)BN << fm.csh);
        fm.R = (int)pbr - fm.b;
        if (fm.R <= 0) return GPX_OK;                       // nothing at or below the diagonal
        int64_t np = 0;
        for (int64_t pc = 0; pc < pbc; ++pc) {
            const int64_t cnt = fm.R - (int64_t)fm.a * pc;
            if (cnt <= 0) break;
            np += cnt;
        }
        fm.np = (int)np;
    }
    const int nbatch = bt ? bt->count : 1;
    fm.sA = bt ? bt->sA : 0; fm.sB = bt ? bt->sB : 0; fm.sC = bt ? bt->sC : 0;
    fm.tA = bt ? bt->tA : 0; fm.tB = bt ? bt->tB : 0; fm.tC = bt ? bt->tC : 0;
    const int nbatch2 = bt ? bt->count2 : 1;
    fm.dbegin = 0x7fffffff; fm.ndiag = 0; fm.bdiag = 0;
    fm.ktri = (ktri == 1 && M == N && N == K) ? 1 : ((ktri == 2 && N == K) ? 2 : 0);
    int64_t dblocks = 0;
    {
        if (!fm.exact && BN == 128 && tri == GPX_LOWER && fm.a == 1 && fm.csh == 3 && fm.pm1nb == 0 &&
            col0 - row0 == (int64_t)fm.b * 1024 && fm.np > 0) {
            const int64_t pbc = cdiv(N, 1024);
            fm.bdiag = fm.b;
            fm.ndiag = (int)std::min<int64_t>(pbc, fm.R);
            fm.b += 1; fm.R -= 1;
            int64_t npa = 0;
            for (int64_t pc = 0; pc < pbc; ++pc) {
                const int64_t cnt = fm.R - pc;
                if (cnt <= 0) break;
                npa += cnt;
            }
            fm.np = (int)npa;
            dblocks = cdiv(fm.ndiag, 8) * 8 * 36;
        }
    }
    const int64_t np = fm.np;
    if (fm.exact) { if (fm.eT <= 0) return GPX_OK; }
    else if (np <= 0 && dblocks == 0) return GPX_OK;
    {
        fm.vec_c = (ldc % 2 == 0 && N % 2 == 0 && N >= 2 && ((uintptr_t)C) % (2 * sizeof(T)) == 0) ? 1 : 0;
        // default on: measured epilogue 20 k -> 11.7 k cycles per tile, whole fit 1.59 -> 1.54 s
        fm.atomic_c = 1;
    }
    {
        fm.stamps = g_gemm_stamps;
    }
    const int64_t ablocks = (cdiv(np, 8) * 8 * (1024 / F_BM)) << fm.csh;
    if (dblocks) fm.dbegin = (int)ablocks;
    int64_t blocks = ablocks + dblocks;
    if (fm.exact) blocks = cdiv(fm.eT, (int64_t)8 << fm.ecl) * ((int64_t)8 << fm.ecl);
    ProfScope prof(TAG == 1 ? (BN == 64 ? PC_GEMM_N64 : PC_GEMM) : (BN == 128 ? PC_GEMM_PANEL : PC_GEMM_SKINNY),
                   (work >= 0 ? work : 2.0 * (double)K * updated_elements(M, N, tri, row0, col0)) * nbatch * nbatch2, st);
    hipLaunchKernelGGL((gemm_nt_fast_kernel<T, BN, TAG, ABL>), dim3((unsigned)blocks, (unsigned)nbatch, (unsigned)nbatch2), dim3(F_BM * 2), F_SMEM + pad_lds, st, M, N, K,
                       (const T *)A, lda, (const T *)B, ldb, (T *)C, ldc, (T)alpha, tri, row0, col0, fm, beta0);
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}

int gemm_nt(int dtype, int64_t M, int64_t N, int64_t K, const void *A, int64_t lda, const void *B,
            int64_t ldb, void *C, int64_t ldc, double alpha, int tri, int64_t row0, int64_t col0,
            hipStream_t st, int beta0, int ktri, const Batch *bt, int wide_tiles)
{
    if (M <= 0 || N <= 0 || K <= 0) return GPX_OK;
    const bool no_fast = tune().gemm_no_fast;
    const int64_t epk = 128 / (int64_t)esize(dtype), ch = 16 / (int64_t)esize(dtype);
    const bool fast = !no_fast && K % epk == 0 && lda % ch == 0 && ldb % ch == 0 &&
                      ((uintptr_t)A) % 16 == 0 && ((uintptr_t)B) % 16 == 0;
    route_hit(fast ? RT_GEMM_FAST : RT_GEMM_GENERIC);
    if (fast) {
        // few 128 x 128 tiles (a product that cannot fill the chip anyway): 128 x 64 tiles halve the time of the one round
        // there is -- the posterior covariance's in-block products, 1024 x 512 x 512: 32 tiles
        const int64_t t128 = cdiv(M, 128) * cdiv(N, 128) * (bt ? (int64_t)bt->count * std::max(1, bt->count2) : 1);
        const bool few = !wide_tiles && !ktri && tri == GPX_FULL && t128 <= tune().gemm_bn64_tiles;
        if (N <= 64 || few) {
            if (dtype == GPX_F64)
                return launch_gemm_nt_fast<double, 64, 0>(M, N, K, A, lda, B, ldb, C, ldc, alpha, tri, row0, col0,
                                                               st, nullptr, -1.0, beta0, 0, bt);
            return launch_gemm_nt_fast<float, 64, 0>(M, N, K, A, lda, B, ldb, C, ldc, alpha, tri, row0, col0, st,
                                                          nullptr, -1.0, beta0, 0, bt);
        }
        if (dtype == GPX_F64)
            return launch_gemm_nt_fast<double, 128, 0>(M, N, K, A, lda, B, ldb, C, ldc, alpha, tri, row0,
                                                            col0, st, nullptr, -1.0, beta0, ktri, bt);
        return launch_gemm_nt_fast<float, 128, 0>(M, N, K, A, lda, B, ldb, C, ldc, alpha, tri, row0, col0,
                                                       st, nullptr, -1.0, beta0, ktri, bt);
    }
    if (beta0) { set_error("gemm_nt: beta = 0 needs the aligned fast path"); return GPX_ERR_UNSUPPORTED; }
    if (bt && bt->count2 > 1) { set_error("gemm_nt: a two-dimensional batch needs the aligned fast path"); return GPX_ERR_UNSUPPORTED; }
    Batch one; one.count = 1; one.sA = one.sB = one.sC = 0;
    if (dtype == GPX_F64)
        return launch_gemm_nt<double>(M, N, K, A, lda, B, ldb, C, ldc, alpha, tri, row0, col0, st, bt ? *bt : one);
    return launch_gemm_nt<float>(M, N, K, A, lda, B, ldb, C, ldc, alpha, tri, row0, col0, st, bt ? *bt : one);
}


// ---------------------------------------------------------------------------
// Trailing update of the blocked Cholesky on a 1-D block-cyclic column layout.
//   Cloc (n x local columns, ldc): this rank's block columns; local block jl is
//   global block jl * P + rank (width nb).  Updates rows [row_begin, n) of the
//   local columns [cl0, cl1) with the factored panel Pb (row i of Pb = global row
//   k0 + i, kb columns, ldp):   C[g, c] -= sum_k Pb[g - k0, k] * Pb[gcol(c) - k0, k]
//   for global row g >= global column gcol(c).  P = 1, rank = 0 is the ordinary
//   single-GPU SYRK.  One launch for all local block columns (staircase map).
// ---------------------------------------------------------------------------
int syrk_bc(int dtype, int64_t n, int64_t row_begin, void *Cloc, int64_t ldc, int64_t cl0, int64_t cl1,
            const void *Pb, int64_t ldp, int64_t k0, int64_t kb, int64_t nb, int P, int rank,
            hipStream_t st, const int *abort_flag, const Batch *bt)
{
    const int64_t M = n - row_begin, Ncols = cl1 - cl0;
    if (M <= 0 || Ncols <= 0 || kb <= 0) return GPX_OK;
    const size_t es = esize(dtype);
    const int64_t epk = 128 / (int64_t)es, ch = 16 / (int64_t)es;
    auto gcol = [&](int64_t c) { return ((c / nb) * P + rank) * nb + c % nb; };   // local -> global column
    // algorithmic flops: 2 * kb per updated element (global row >= global column)
    double elems = 0;
    if (P == 1) {
        // one rank: global column = local column.  Closed form (this runs on the host between a panel's launch and the
        // update's, in every step: the column loop below cost ~8 us at n = 8192)
        const int64_t a = std::min(cl1, std::max(cl0, row_begin)), b = std::min(cl1, n);     // [cl0, a): full height; [a, b): from the diagonal
        if (row_begin < n) elems += (double)(a - cl0) * (double)(n - row_begin);
        if (b > a) elems += (double)(b - a) * (double)n - 0.5 * (double)(a + b - 1) * (double)(b - a);
    } else
    for (int64_t c = cl0; c < cl1; c += nb) {
        const int64_t w = std::min(nb - c % nb, cl1 - c), g = gcol(c);
        for (int64_t j = 0; j < w; ++j) {
            const int64_t first = std::max(row_begin, g + j);
            if (first < n) elems += (double)(n - first);
        }
    }
    const double work = 2.0 * (double)kb * elems;
    const bool fast = kb % epk == 0 && ldp % ch == 0 && ((uintptr_t)Pb) % 16 == 0 && cl0 % 128 == 0 &&
                      nb % 128 == 0 && (1024 % nb == 0) && !tune().gemm_no_fast;
    char *C = (char *)Cloc + (row_begin * ldc + cl0) * es;
    const char *A = (const char *)Pb + (row_begin - k0) * ldp * es;
    if (fast) {
        GemmMap fm{};
        fm.cbase = cl0; fm.nb = nb; fm.pm1nb = (int64_t)(P - 1) * nb;
        fm.boff = cl0 + (int64_t)rank * nb - k0;
        fm.brows = n - k0;
        const int64_t G0 = gcol(cl0);
        const int64_t pbr = cdiv(M, 1024), pbc = cdiv(Ncols, 1024);
        fm.a = P; fm.csh = 3;
        fm.b = (int)std::max<int64_t>(0, (G0 - row_begin) / 1024);
        fm.R = (int)pbr - fm.b;
        if (fm.R <= 0) return GPX_OK;
        int64_t np = 0;
        for (int64_t pc = 0; pc < pbc; ++pc) {
            const int64_t cnt = fm.R - (int64_t)fm.a * pc;
            if (cnt <= 0) break;
            np += cnt;
        }
        fm.np = (int)np;
        fm.stamps = nullptr;
        fm.abort_flag = abort_flag; fm.sflag = bt ? 1 : 0;
        fm.exact = 0; fm.ecs = 0;
        bool bn64 = false;
        {
            // single rank, triangle aligned to the 128 x 128 tiles: enumerate exactly the tiles that exist
            const int64_t exact_env = tune().gemm_exact;
            const int64_t off = row_begin - G0;                       // row origin minus column origin (global)
            if (exact_env && off % 128 == 0 && cdiv(M, 1024) <= G_MAX_BANDS && cl0 % nb == 0) {
                // Short updates take 128 x 64 tiles: twice the tiles at little more than half the time each, so the
                // last, partly filled round of workgroups costs half as much and the 2 - 5 rounds of an n <= 8192 step
                // lose less to it (full products of these shapes, K = 256: 7680 x 3840 46 -> 55 TF/s, 5632 x 2816
                // 35 -> 47, 2048 x 1024 14 -> 24; tools/bn64_probe.py).  Decided on the 128 x 128 tile count of the launch
                // (all matrices of a lock-step batch together): potrf n = 8192 6.22 -> 5.73 ms, 4096 2.08 -> 1.97, 12288
                // 14.32 -> 13.83, 16384 28.37 -> 27.79; thresholds 1024 / 2100 / 2600 / 3400 / 5000 at n = 12288: 13.9x /
                // 13.86 / 13.83 / 13.96 / 14.08; fp32 n = 8192 4.60 -> 4.47, N = 32768 94.45 -> 93.89 with 4000.
                const int64_t t128 = std::min<int64_t>((int64_t)cdiv(M, 128) * (cdiv(M, 128) + 1) / 2, (int64_t)cdiv(M, 128) * cdiv(Ncols, 128)) *
                                     (bt ? bt->count : 1);
                bn64 = t128 <= tune().syrk_bn64_tiles[dtype == GPX_F64 ? 0 : 1] && nb % 64 == 0;
                const int cs = bn64 ? 1 : 0, cw = 128 >> cs;
                const int TR = (int)cdiv(M, 128), TC = (int)cdiv(Ncols, cw), D = (int)(off / 128);
                const int bands = (int)cdiv(TR, 8);
                const int tpb = (int)(nb / cw), pm1t = (P - 1) * tpb;
                fm.etpb = tpb; fm.epm1t = pm1t; fm.ecs = cs;
                int total = 0;
                for (int b = 0; b < bands; ++b) {
                    fm.epre[b] = total;
                    const int h = std::min(8, TR - 8 * b), dj = 8 * b + D;
                    int jfull = 0;
                    if (dj >= 0) {
                        const int djc = (dj + 1) << cs;
                        const int period = tpb + pm1t, qd = djc / period, rem = djc - qd * period;
                        jfull = std::min(TC, qd * tpb + std::min(rem, tpb));
                    }
                    total += jfull * h;
                    for (int j = jfull; j < TC; ++j) {
                        const int gj = j + (j / tpb) * pm1t;
                        const int cnt = std::max(0, h - std::max(0, (gj >> cs) - dj));
                        if (cnt == 0) break;
                        total += cnt;
                    }
                }
                fm.epre[bands] = total;
                fm.exact = 1; fm.eT = total; fm.eD = D; fm.eTR = TR; fm.eTC = TC; fm.ebands = bands;
                // Chunk of 2^cl consecutive tiles per XCD turn.  64 = one 8 x 8 patch (8 A- and 8 B-slices in that XCD's
                // L2 at a time).  An XCD runs 64 tiles at a time (56 with CUs reserved for the panel stream), so the
                // chunk is also the granule of the load balance BETWEEN the XCDs: with 64-tile chunks one XCD can hold a
                // whole round of tiles more than its neighbours, and a short update (n <= 12288: 1 - 5 rounds in all)
                // waits for it -- measured at n = 8192: steps with 1711 / 1275 / 820 tiles took 5 / 4 / 3 rounds of
                // ~75 us where 4 / 3 / 2 would do (gpurun_out/r3_timeline8192.txt).  Up to GPX_GEMM_FINE_TILES tiles the
                // chunk is one tile COLUMN of a band (8 tiles: the same 8 A-slices as its neighbours on that XCD, one
                // B-slice), which keeps the slices per resident tile the same and evens the XCDs out to within 8 tiles.
                int cl = total <= tune().gemm_fine_tiles ? 3 : 6;
                while (cl > 2 && ((int64_t)8 << cl) > total) --cl;   // few tiles: smaller chunks, every XCD still gets some
                fm.ecl = cl;
                if (total <= 0) return GPX_OK;
            }
        }
        route_hit(fm.exact ? RT_SYRK_EXACT : RT_SYRK_PATCH);
        if (fm.exact && bn64) {
            if (dtype == GPX_F64)
                return launch_gemm_nt_fast<double, 64, 1>(M, Ncols, kb, A, ldp, Pb, ldp, C, ldc, -1.0, GPX_LOWER,
                                                               row_begin, cl0 + (int64_t)rank * nb, st, &fm, work, 0, 0, bt);
            return launch_gemm_nt_fast<float, 64, 1>(M, Ncols, kb, A, ldp, Pb, ldp, C, ldc, -1.0, GPX_LOWER,
                                                          row_begin, cl0 + (int64_t)rank * nb, st, &fm, work, 0, 0, bt);
        }
        if (dtype == GPX_F64 && tune().gemm_ablate) {
#define GPX_ABL_CASE(V)                                                                                            \
    case V: return launch_gemm_nt_fast<double, 128, 1, V>(M, Ncols, kb, A, ldp, Pb, ldp, C, ldc, -1.0, GPX_LOWER, \
                                                               row_begin, cl0 + (int64_t)rank * nb, st, &fm, work, 0, 0, bt)
            switch ((int)tune().gemm_ablate) { GPX_ABL_CASE(1); GPX_ABL_CASE(2); GPX_ABL_CASE(4); GPX_ABL_CASE(7); GPX_ABL_CASE(8); GPX_ABL_CASE(15); GPX_ABL_CASE(16); default: break; }
#undef GPX_ABL_CASE
        }
        if (dtype == GPX_F64)
            return launch_gemm_nt_fast<double, 128, 1>(M, Ncols, kb, A, ldp, Pb, ldp, C, ldc, -1.0, GPX_LOWER,
                                                            row_begin, cl0 + (int64_t)rank * nb, st, &fm, work, 0, 0, bt);
        return launch_gemm_nt_fast<float, 128, 1>(M, Ncols, kb, A, ldp, Pb, ldp, C, ldc, -1.0, GPX_LOWER,
                                                       row_begin, cl0 + (int64_t)rank * nb, st, &fm, work, 0, 0, bt);
    }
    // generic route: one launch per local block column
    for (int64_t c = cl0; c < cl1;) {
        const int64_t w = std::min(nb - c % nb, cl1 - c), g = gcol(c);
        const int64_t rb = std::max(row_begin, g);
        if (rb < n)
            GPX_TRY(gemm_nt(dtype, n - rb, w, kb, (const char *)Pb + (rb - k0) * ldp * es, ldp,
                            (const char *)Pb + (g - k0) * ldp * es, ldp,
                            (char *)Cloc + (rb * ldc + c) * es, ldc, -1.0, GPX_LOWER, rb, g, st, 0, 0, bt));
        c += w;
    }
    return GPX_OK;
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

extern "C" int gpx_d_syrk_bc(int dtype, int64_t n, int64_t row_begin, void *Cloc, int64_t ldc,
                             int64_t cl0, int64_t cl1, const void *Pb, int64_t ldp, int64_t k0,
                             int64_t kb, int64_t nb, int P, int rank, void *stream)
{
    GPX_TRY(ensure_device());
    GPX_ARG(dtype == GPX_F64 || dtype == GPX_F32, "dtype must be GPX_F64 or GPX_F32");
    GPX_ARG(n >= 0 && row_begin >= 0 && cl0 >= 0 && cl1 >= cl0 && kb >= 0, "bad dimensions");
    GPX_ARG(P >= 1 && rank >= 0 && rank < P && nb >= 64 && nb % 64 == 0, "bad P / rank / nb");
    GPX_ARG(k0 + kb <= row_begin, "row_begin must lie below the panel's diagonal block");
    if (n == 0 || cl1 == cl0 || kb == 0) return GPX_OK;
    GPX_ARG(Cloc && Pb, "NULL pointer");
    const int64_t ch = 16 / (int64_t)esize(dtype);
    GPX_ARG(ldp >= kb && ldp % ch == 0 && ((uintptr_t)Pb) % 16 == 0, "panel must be 16-byte aligned");
    return syrk_bc(dtype, n, row_begin, Cloc, ldc, cl0, cl1, Pb, ldp, k0, kb, nb, P, rank, S(stream));
}

// diagnostic hook (not declared in gpx.h): per-workgroup s_memtime stamps of the fast GEMM
extern "C" int gpx_debug_gemm_stamps(void *dev_buffer)
{
    g_gemm_stamps = (unsigned long long *)dev_buffer;
    return GPX_OK;
}
