// gpx_kmat.hip -- pairwise kernel-matrix build and fused posterior mean (gfx950).
//
// Replaces the O(n*m) double loops of gp/ext/gaussian_c.pyx:18-164 and
// gp/ext/periodic_c.pyx:18-235 (one exp per entry, single CPU thread) and the
// `K += eye(n) * s**2` temporaries of gp/gp.py:265.
//
// Roofline: HBM write bandwidth for small d (d = 1: 4.2 TB/s written), the fp64 vector pipe from d ~ 12 up (d = 32:
// 64 instructions per entry for the distance, ~28 for the exp).  Algorithmic bytes per launch = n*m*sizeof(T)
// written (+ (n+m)*d*sizeof(T) read, negligible).  One workgroup owns a 64 x (64*VEC) output tile; its column points
// are staged in LDS once (transposed), every lane owns VEC consecutive columns so that a wave stores 1 KiB contiguous
// per row (16 B per lane); the row points, the same for every lane of a wave, come through the scalar cache into
// SGPRs (as LDS broadcasts they made the LDS return path the bound: N = 65536, d = 32: 9.1-9.6 -> 8.05 ms).  The
// squared distance is accumulated directly as sum_k (a_k - b_k)^2 (never |a|^2+|b|^2-2ab: that loses the digits near
// r = 0 that the reference keeps).
#include "gpx_common.h"
#include "gpx_kernels_dev.h"
#include <vector>

namespace gpx {

// ---------------------------------------------------------------------------
// member constants, precomputed in f64 exactly as the reference spells them
// gaussian (gaussian_c.pyx): c[0]=c1  c[1]=c2  c[2]=c3  c[3]=c4  c[4]=form
// periodic (periodic_c.pyx): c[0]=h  c[1]=w  c[2]=p
// ---------------------------------------------------------------------------
int make_kparams(int kernel, int member, const double *params, double diag_add, KParams *out)
{
    memset(out, 0, sizeof(*out));
    out->kernel = kernel;
    out->member = member;
    out->diag_add = diag_add;
    if (!params) { set_error("params is NULL"); return GPX_ERR_ARG; }
    if (kernel == GPX_KERNEL_GAUSSIAN) {
        const double h = params[0], w = params[1];
        const double S = sqrt(2.0 / M_PI);          // gaussian_c.pyx:14
        const double h2 = h * h, w2 = w * w;
        out->c[0] = -0.5 / w2;                      // c1
        switch (member) {
        case GPX_K:        out->c[1] = 0.5 * S * h2 / w; out->c[4] = 0; break;            // :28
        case GPX_DK_DH:    out->c[1] = S * h / w; out->c[4] = 0; break;                   // :61
        case GPX_DK_DW:    out->c[1] = 0.5 * S * h2 / w2;                                  // :82-83
                           out->c[2] = 0.5 * S * h2 / pow(w, 4); out->c[4] = 1; break;
        case GPX_D2K_DHDH: out->c[1] = S / w; out->c[4] = 0; break;                        // :105
        case GPX_D2K_DHDW: out->c[1] = S * h / w2;                                          // :126-127
                           out->c[2] = S * h / pow(w, 4); out->c[4] = 1; break;
        case GPX_D2K_DWDW: out->c[1] = S * h2 / pow(w, 3);                                  // :153-155
                           out->c[2] = 2.5 * S * h2 / pow(w, 5);
                           out->c[3] = 0.5 * S * h2 / pow(w, 7); out->c[4] = 2; break;
        default: set_error("gaussian kernel has no member %d", member); return GPX_ERR_ARG;
        }
    } else if (kernel == GPX_KERNEL_PERIODIC) {
        if (member < GPX_K || member > GPX_D2K_DPDP) {
            set_error("periodic kernel has no member %d", member);
            return GPX_ERR_ARG;
        }
        out->c[0] = params[0];
        out->c[1] = params[1];
        out->c[2] = params[2];
    } else {
        set_error("unknown kernel family %d", kernel);
        return GPX_ERR_ARG;
    }
    return GPX_OK;
}

struct KP32 { float c[6]; };

constexpr int KM_ROWS = 64;        // tile rows
constexpr int KM_RB = 8;           // rows register-blocked per lane (round 6: 4 -> 8: half the LDS reads and loop overhead per entry,
                                   // N = 65536 d = 32: 7.65 -> 7.3 ms)

// MODE 0..2: gaussian FORM 0..2 ; MODE 3: periodic K, any d ; MODE 4: periodic member, d == 1
template <typename T, int MODE>
__global__ __launch_bounds__(256) void kmat_kernel(const T *__restrict__ x1, int64_t n,
                                                   const T *__restrict__ x2, int64_t m, int d,
                                                   KParams kp, int tri, int aligned,
                                                   T *__restrict__ out, int64_t ld, int rblk)
{
    // rblk: row blocks of KM_ROWS per workgroup.  The column points are staged (transposed) ONCE and serve rblk row
    // blocks (round 6: one block per workgroup paid the 32 KB staging pass and its barrier per 64 rows, and half of
    // the 524 k workgroups of a lower-only N = 65536 build existed only to find themselves above the diagonal)
    constexpr int VEC = Vec<T>::N;
    constexpr int TN = 64 * VEC;
    constexpr int TNP = TN + VEC;                     // padded row of s2: the transposing stores below
                                                      // (consecutive threads -> consecutive k) spread over the banks
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // (the row points are staged only where lanes read them as vectors, MODE 4; elsewhere they come through SGPRs)
    T *s1 = reinterpret_cast<T *>(smem_raw);          // [KM_ROWS][d]
    T *s2 = s1 + (MODE == 4 ? (size_t)KM_ROWS * d : (size_t)0);   // [d][TNP]  (transposed: lanes contiguous)

    const int64_t row00 = (int64_t)blockIdx.y * KM_ROWS * rblk;
    const int64_t col0 = (int64_t)blockIdx.x * TN;
    if (tri == GPX_LOWER && col0 > row00 + (int64_t)KM_ROWS * rblk - 1) return;   // every row block strictly above the diagonal

    const int tid = threadIdx.x;
    // stage the two point sets (coalesced: consecutive threads -> consecutive elements)
    // (the tile's points are contiguous in memory: element idx of the tile is x[row0 * d + idx])
    {
        const int64_t lim1 = (n - row00) * d;           // (MODE 4 is launched with rblk = 1)
        const T *g1 = x1 + row00 * d;
        if (MODE == 4)
            for (int idx = tid; idx < KM_ROWS * d; idx += 256) s1[idx] = (idx < lim1) ? g1[idx] : (T)0;
        const int64_t lim2 = (m - col0) * d;
        const T *g2 = x2 + col0 * d;
        const int qd = 256 / d, rd = 256 - qd * d;    // idx += 256  <=>  (c, k) += (qd, rd) with carry
        int c = tid / d, k = tid - c * d;
        for (int idx = tid; idx < TN * d; idx += 256) {
            s2[(size_t)k * TNP + c] = (idx < lim2) ? g2[idx] : (T)0;
            c += qd; k += rd;
            if (k >= d) { k -= d; ++c; }
        }
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // (uniform by construction; this tells the compiler)
    const int cbase = lane * VEC;
    const T c1 = (T)kp.c[0], c2 = (T)kp.c[1], c3 = (T)kp.c[2], c4 = (T)kp.c[3];
    const T dadd = (T)kp.diag_add;
#pragma unroll 1
  for (int q = 0; q < rblk; ++q) {
    const int64_t row0 = row00 + (int64_t)q * KM_ROWS;
    if (row0 >= n) break;
    if (tri == GPX_LOWER && col0 > row0 + KM_ROWS - 1) continue;  // this row block is strictly above the diagonal
    // workgroup-uniform: all 64 x TN entries exist, stores are 16-byte aligned, the diagonal (where diag_add
    // goes) does not cross the tile
    const bool interior = aligned && row0 + KM_ROWS <= n && col0 + TN <= m &&
                          (kp.diag_add == 0.0 || col0 >= row0 + KM_ROWS || col0 + TN <= row0);

#pragma unroll 1
    for (int rb = 0; rb < 16; rb += KM_RB) {
        const int rloc = wave * 16 + rb;
        T acc[KM_RB][VEC];
#pragma unroll
        for (int r = 0; r < KM_RB; ++r)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[r][v] = (T)0;

        if (MODE == 4) {
            // d == 1: acc holds the signed difference
#pragma unroll
            for (int r = 0; r < KM_RB; ++r)
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[r][v] = s1[rloc + r] - s2[cbase + v];
        } else {
            // the row points are the same for every lane of a wave: they come through the scalar cache into SGPRs
            // (wave-uniform addresses), not as four 512-byte LDS broadcasts per coordinate -- with those the LDS return
            // path, shared by the four SIMDs, was the bound (24 of its cycles per 64 VALU cycles and SIMD)
            const T *arow[KM_RB];
#pragma unroll
            for (int r = 0; r < KM_RB; ++r)
                arow[r] = x1 + min(row0 + wave_u * 16 + rb + r, n - 1) * d;
#pragma unroll 4
            for (int k = 0; k < d; ++k) {
                T b[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) b[v] = s2[(size_t)k * TNP + cbase + v];
#pragma unroll
                for (int r = 0; r < KM_RB; ++r) {
                    const T a = arow[r][k];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        if (MODE == 3) {
                            const T sn = sin((T)0.5 * (a - b[v]) / (T)kp.c[2]);
                            acc[r][v] = fma(sn, sn, acc[r][v]);
                        } else {
                            const T t = a - b[v];
                            acc[r][v] = fma(t, t, acc[r][v]);
                        }
                    }
                }
            }
        }

        if (interior) {
            // the tile lies strictly inside the matrix and off the diagonal: no row / column bounds, no
            // diagonal test, one 16-byte store per lane and row (the guarded path below costs ~30 more vector
            // instructions per entry: per-element predicates and the byte-wise assembly of a partial store)
#pragma unroll
            for (int r = 0; r < KM_RB; ++r) {
                typename Vec<T>::type pk;
                T *pv = reinterpret_cast<T *>(&pk);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    if (MODE <= 2) {
                        pv[v] = gaussian_entry<T, MODE>(acc[r][v], c1, c2, c3, c4);
                    } else if (MODE == 3) {
                        const T h = (T)kp.c[0], w = (T)kp.c[1];
                        pv[v] = (h * h) * dev_exp<T>((T)-2.0 * acc[r][v] / (w * w));
                    } else {
                        pv[v] = periodic_entry<T>(kp.member, acc[r][v], (T)kp.c[0], (T)kp.c[1], (T)kp.c[2]);
                    }
                }
                *reinterpret_cast<typename Vec<T>::type *>(out + (row0 + rloc + r) * ld + col0 + cbase) = pk;
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < KM_RB; ++r) {
            const int64_t gi = row0 + rloc + r;
            if (gi >= n) continue;
            T val[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                T x;
                if (MODE <= 2) {
                    x = gaussian_entry<T, MODE>(acc[r][v], c1, c2, c3, c4);
                } else if (MODE == 3) {
                    const T h = (T)kp.c[0], w = (T)kp.c[1];
                    x = (h * h) * dev_exp<T>((T)-2.0 * acc[r][v] / (w * w));
                } else {
                    x = periodic_entry<T>(kp.member, acc[r][v], (T)kp.c[0], (T)kp.c[1], (T)kp.c[2]);
                }
                if (gi == col0 + cbase + v) x += dadd;
                val[v] = x;
            }
            T *dst = out + gi * ld + col0 + cbase;
            if (aligned && col0 + cbase + VEC <= m) {
                typename Vec<T>::type pk;
                memcpy(&pk, val, sizeof(pk));
                *reinterpret_cast<typename Vec<T>::type *>(dst) = pk;
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (col0 + cbase + v < m) dst[v] = val[v];
            }
        }
    }
  }
}

template <typename T>
static int launch_kmat(const void *x1, int64_t n, const void *x2, int64_t m, int d,
                       const KParams &kp, int tri, void *out, int64_t ld, hipStream_t st)
{
    constexpr int VEC = Vec<T>::N;
    constexpr int TN = 64 * VEC;
    int mode;
    if (kp.kernel == GPX_KERNEL_GAUSSIAN) mode = (int)kp.c[4];
    else mode = (kp.member == GPX_K) ? 3 : 4;
    if (mode == 4 && d != 1) {
        set_error("periodic derivative members need d == 1 (got %d)", d);
        return GPX_ERR_UNSUPPORTED;
    }
    const size_t smem = ((mode == 4 ? (size_t)KM_ROWS * d : (size_t)0) + (size_t)d * (TN + VEC)) * sizeof(T);
    if (smem > 96 * 1024) {
        set_error("kmat: d = %d too large for the LDS-staged tile (96 KiB of column points)", d);
        return GPX_ERR_UNSUPPORTED;
    }
    const int aligned = (ld % VEC == 0) && (((uintptr_t)out) % 16 == 0);
    // four row blocks per workgroup once that still leaves several workgroups per CU
    const int rblk = (mode != 4 && cdiv(m, TN) * cdiv(n, 4 * KM_ROWS) >= 2048) ? 4 : 1;
    dim3 grid((unsigned)cdiv(m, TN), (unsigned)cdiv(n, (int64_t)KM_ROWS * rblk));
    dim3 block(256);
    const T *a = (const T *)x1;
    const T *b = (const T *)x2;
    T *o = (T *)out;
    double bytes = 0;
    if (g_prof_on) {
        // bytes actually written: every tile that is not strictly above the diagonal
        for (int64_t r0 = 0; r0 < n; r0 += KM_ROWS) {
            const int64_t rows = std::min<int64_t>(KM_ROWS, n - r0);
            int64_t cols = m;
            if (tri == GPX_LOWER) cols = std::min<int64_t>(m, ((r0 + KM_ROWS - 1) / TN + 1) * TN);
            bytes += (double)rows * cols * sizeof(T);
        }
    }
    ProfScope prof(PC_KMAT, bytes, st);
#define GPX_KM_LAUNCH(MODE)                                                                   \
    do {                                                                                      \
        if (smem > 48 * 1024)                                                                 \
            GPX_HIP(hipFuncSetAttribute((const void *)kmat_kernel<T, MODE>,                   \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL((kmat_kernel<T, MODE>), grid, block, smem, st, a, n, b, m, d, kp,  \
                           tri, aligned, o, ld, rblk);                                        \
    } while (0)
    switch (mode) {
    case 0: GPX_KM_LAUNCH(0); break;
    case 1: GPX_KM_LAUNCH(1); break;
    case 2: GPX_KM_LAUNCH(2); break;
    case 3: GPX_KM_LAUNCH(3); break;
    default: GPX_KM_LAUNCH(4); break;
    }
#undef GPX_KM_LAUNCH
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}

// ---------------------------------------------------------------------------
// fused posterior mean: out[i] = sum_j K(xo[i], x[j]) * alpha[j]   (gp/gp.py:597)
// Workgroup (bx, by) owns MP test points and the by-th slice of the training set,
// which streams through LDS in chunks of 256 points (one per thread, transposed and
// padded so both the staging stores and the reads are conflict-free).  Sums are
// kept in f64; a slice's partial sums go to `partial[by][i]` and a second launch
// adds the slices in a fixed order (deterministic: no atomics).
// ---------------------------------------------------------------------------
constexpr int MP = 8;
constexpr int MCP = 257;           // padded chunk row

// KIND: kernel family; FORM: gaussian member form 0..2 (see gaussian_entry), or for the periodic family
// 0 = K for any d, 1 = any member at d == 1 (kp.member) -- out = member(xo, x) @ alpha.
template <typename T, int KIND, int FORM>
__global__ __launch_bounds__(256) void mean_kernel(const T *__restrict__ xo, int64_t m,
                                                   const T *__restrict__ x, int64_t n, int d,
                                                   KParams kp, const T *__restrict__ alpha,
                                                   int64_t slice_len, double *__restrict__ partial,
                                                   T *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *sx = reinterpret_cast<T *>(smem_raw);            // [d][MCP] chunk of x, transposed
    __shared__ double red[4][MP];

    const int tid = threadIdx.x;
    const int64_t p0 = (int64_t)blockIdx.x * MP;
    double acc[MP];
#pragma unroll
    for (int pp = 0; pp < MP; ++pp) acc[pp] = 0.0;
    const T *orow[MP];
#pragma unroll
    for (int pp = 0; pp < MP; ++pp) orow[pp] = xo + min(p0 + pp, m - 1) * d;

    const T c1 = (T)kp.c[0], c2 = (T)kp.c[1], c3 = (T)kp.c[2], c4 = (T)kp.c[3];
    const int qd = 256 / d, rd = 256 - qd * d;          // idx += 256  <=>  (c, k) += (qd, rd) with carry
    const int cst = tid / d, kst = tid - cst * d;
    const int64_t jbeg = (int64_t)blockIdx.y * slice_len, jend = min(n, jbeg + slice_len);
    for (int64_t j0 = jbeg; j0 < jend; j0 += 256) {
        __syncthreads();
        {
            const int64_t lim = (jend - j0) * d;
            const T *g = x + j0 * d;
            int c = cst, k = kst;
            for (int idx = tid; idx < 256 * d; idx += 256) {
                sx[(size_t)k * MCP + c] = (idx < lim) ? g[idx] : (T)0;
                c += qd; k += rd;
                if (k >= d) { k -= d; ++c; }
            }
        }
        __syncthreads();
        const int64_t j = j0 + tid;
        if (j < jend) {
            const T aj = alpha[j];
            T r[MP];
#pragma unroll
            for (int pp = 0; pp < MP; ++pp) r[pp] = (T)0;
            // (the test points are the same for every lane: SGPR operands through the scalar cache, as in kmat_kernel)
            for (int k = 0; k < d; ++k) {
                const T b = sx[(size_t)k * MCP + tid];
#pragma unroll
                for (int pp = 0; pp < MP; ++pp) {
                    const T a = orow[pp][k];
                    if (KIND == GPX_KERNEL_GAUSSIAN) {
                        const T t = a - b;
                        r[pp] = fma(t, t, r[pp]);
                    } else if (FORM == 1) {
                        r[pp] = a - b;                                // d == 1: the signed difference
                    } else {
                        const T sn = sin((T)0.5 * (a - b) / (T)kp.c[2]);
                        r[pp] = fma(sn, sn, r[pp]);
                    }
                }
            }
#pragma unroll
            for (int pp = 0; pp < MP; ++pp) {
                T kv;
                if (KIND == GPX_KERNEL_GAUSSIAN) {
                    kv = gaussian_entry<T, FORM>(r[pp], c1, c2, c3, c4);
                } else if (FORM == 1) {
                    kv = periodic_entry<T>(kp.member, r[pp], (T)kp.c[0], (T)kp.c[1], (T)kp.c[2]);
                } else {
                    const T h = (T)kp.c[0], w = (T)kp.c[1];
                    kv = (h * h) * dev_exp<T>((T)-2.0 * r[pp] / (w * w));
                }
                acc[pp] += (double)kv * (double)aj;
            }
        }
    }
    // wave reduction (64 lanes), then across the 4 waves in a fixed order
#pragma unroll
    for (int pp = 0; pp < MP; ++pp) {
        double v = acc[pp];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((tid & 63) == 0) red[tid >> 6][pp] = v;
    }
    __syncthreads();
    if (tid < MP && p0 + tid < m) {
        const double sum = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        if (partial) partial[(int64_t)blockIdx.y * m + p0 + tid] = sum;
        else out[p0 + tid] = (T)sum;
    }
}

template <typename T>
__global__ void mean_reduce_kernel(const double *__restrict__ partial, int nslice, int64_t m, T *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double sum = 0.0;
    for (int y = 0; y < nslice; ++y) sum += partial[(int64_t)y * m + i];
    out[i] = (T)sum;
}

// grow-only device scratch for the slice partial sums (one per host thread)
struct MeanScratch { void *p = nullptr; size_t bytes = 0; int device = -1; };
static thread_local MeanScratch g_mean_scr;
static int mean_scratch(size_t bytes, void **out)
{
    int dev = 0;
    GPX_HIP(hipGetDevice(&dev));
    if (g_mean_scr.device != dev || g_mean_scr.bytes < bytes) {
        if (g_mean_scr.p && g_mean_scr.device == dev) { GPX_HIP(hipDeviceSynchronize()); (void)hipFree(g_mean_scr.p); }
        g_mean_scr.p = nullptr; g_mean_scr.bytes = 0; g_mean_scr.device = dev;
        GPX_HIP(hipMalloc(&g_mean_scr.p, bytes));
        g_mean_scr.bytes = bytes;
    }
    *out = g_mean_scr.p;
    return GPX_OK;
}

template <typename T>
static int launch_mean(int kernel, const void *xo, int64_t m, const void *x, int64_t n, int d,
                       const KParams &kp, const void *alpha, void *out, hipStream_t st)
{
    const size_t smem = (size_t)d * MCP * sizeof(T);
    if (smem > 96 * 1024) {
        set_error("mean: d = %d too large", d);
        return GPX_ERR_UNSUPPORTED;
    }
    // enough workgroups to fill the chip: slices of the training set when m alone is too small
    const int64_t gx = cdiv(m, MP);
    int64_t nslice = std::max<int64_t>(1, std::min<int64_t>(cdiv(2048, gx), cdiv(n, 256)));
    const int64_t slice_len = cdiv(cdiv(n, nslice), 256) * 256;
    nslice = cdiv(n, slice_len);
    double *partial = nullptr;
    if (nslice > 1) {
        void *scr = nullptr;
        GPX_TRY(mean_scratch((size_t)nslice * m * sizeof(double), &scr));
        partial = (double *)scr;
    }
    dim3 grid((unsigned)gx, (unsigned)nslice), block(256);
    ProfScope prof(PC_MEAN, (double)m * n, st);
#define GPX_MEAN_LAUNCH(KIND, FORM)                                                                 \
    do {                                                                                            \
        if (smem > 48 * 1024)                                                                       \
            GPX_HIP(hipFuncSetAttribute((const void *)mean_kernel<T, KIND, FORM>,                   \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));    \
        hipLaunchKernelGGL((mean_kernel<T, KIND, FORM>), grid, block, smem, st, (const T *)xo, m,   \
                           (const T *)x, n, d, kp, (const T *)alpha, slice_len, partial, (T *)out); \
    } while (0)
    if (kernel == GPX_KERNEL_GAUSSIAN) {
        switch ((int)kp.c[4]) {
        case 0: GPX_MEAN_LAUNCH(GPX_KERNEL_GAUSSIAN, 0); break;
        case 1: GPX_MEAN_LAUNCH(GPX_KERNEL_GAUSSIAN, 1); break;
        default: GPX_MEAN_LAUNCH(GPX_KERNEL_GAUSSIAN, 2); break;
        }
    } else if (kp.member == GPX_K) {
        GPX_MEAN_LAUNCH(GPX_KERNEL_PERIODIC, 0);
    } else {
        if (d != 1) { set_error("periodic derivative members need d == 1 (got %d)", d); return GPX_ERR_UNSUPPORTED; }
        GPX_MEAN_LAUNCH(GPX_KERNEL_PERIODIC, 1);
    }
#undef GPX_MEAN_LAUNCH
    if (partial)
        hipLaunchKernelGGL((mean_reduce_kernel<T>), dim3((unsigned)cdiv(m, 256)), dim3(256), 0, st, partial,
                           (int)nslice, m, (T *)out);
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}


// ---------------------------------------------------------------------------
// Fused gradient reduction (gp/ext/gp_c.pyx:34-49 without its dense products):
//   P[p]  = sum_{j,k} (alpha_j alpha_k - W[j,k]) * dK_p(x_j, x_k)      p < n_kp
//   P[n_kp] = trace(W)
// with W = K^-1 (lower triangle read; both W and dK_p are symmetric, so the sum runs over
// the lower triangle with weight 2 off the diagonal).  The kernel derivatives are evaluated
// on the fly from the squared distance (no n x n Jacobian is materialised).  Deterministic:
// a fixed grid of workgroups walks the tiles in a fixed order, per-workgroup partial sums
// are written out and added up by the host in index order.
// ---------------------------------------------------------------------------
struct GradParams {
    double c1, c2h, c2w, c3w;      // gaussian: e = c1*d2 ; dK_dh = c2h*exp(e) ; dK_dw = exp(e)*(c3w*d2 - c2w)
    double h, w, p;                // periodic
    int kernel, nkp;
};

constexpr int GR_T = 64;           // tile edge
constexpr int GR_BLOCKS = 1024;

template <typename T>
__global__ __launch_bounds__(256) void dloglh_reduce_kernel(const T *__restrict__ x, int64_t n, int d,
                                                            const T *__restrict__ alpha,
                                                            const T *__restrict__ W, int64_t ldw,
                                                            GradParams gp, int64_t ntiles_r,
                                                            double *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *s1 = reinterpret_cast<T *>(smem_raw);          // [GR_T][d]   rows
    T *s2 = s1 + (size_t)GR_T * d;                    // [d][GR_T]   columns, transposed
    T *sa1 = s2 + (size_t)GR_T * d;                   // alpha rows
    T *sa2 = sa1 + GR_T;                              // alpha cols
    __shared__ double red[4][4];
    const int tid = threadIdx.x, col = tid & 63, rg = tid >> 6;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};             // up to 3 kernel params + trace
    const int64_t total = ntiles_r * (ntiles_r + 1) / 2;
    for (int64_t t = blockIdx.x; t < total; t += gridDim.x) {
        int64_t tr = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((tr + 1) * (tr + 2) / 2 <= t) ++tr;
        while (tr * (tr + 1) / 2 > t) --tr;
        const int64_t tc = t - tr * (tr + 1) / 2;
        const int64_t r0 = tr * GR_T, c0 = tc * GR_T;
        __syncthreads();
        for (int idx = tid; idx < GR_T * d; idx += 256) {
            const int r = idx / d, k = idx - r * d;
            s1[idx] = (r0 + r < n) ? x[(r0 + r) * d + k] : (T)0;
            s2[(size_t)k * GR_T + r] = (c0 + r < n) ? x[(c0 + r) * d + k] : (T)0;
        }
        if (tid < GR_T) {
            sa1[tid] = (r0 + tid < n) ? alpha[r0 + tid] : (T)0;
            sa2[tid] = (c0 + tid < n) ? alpha[c0 + tid] : (T)0;
        }
        __syncthreads();
        const int64_t gc = c0 + col;
        const T ak = sa2[col];
        for (int i = 0; i < 16; ++i) {
            const int r = rg + 4 * i;
            const int64_t gr = r0 + r;
            if (gr >= n || gc >= n || gc > gr) continue;
            const double wt = (gc == gr) ? 1.0 : 2.0;
            const double wjk = (double)W[gr * ldw + gc];
            const double coef = wt * ((double)sa1[r] * (double)ak - wjk);
            if (gc == gr) acc[3] += wjk;
            if (gp.kernel == GPX_KERNEL_GAUSSIAN) {
                T d2 = (T)0;
                for (int k = 0; k < d; ++k) {
                    const T tt = s1[r * d + k] - s2[(size_t)k * GR_T + col];
                    d2 = fma(tt, tt, d2);
                }
                const double e = gp.c1 * (double)d2;
                if (!(e < GPX_MIN_LOG)) {
                    const double ex = exp(e);
                    acc[0] += coef * (gp.c2h * ex);
                    acc[1] += coef * (ex * (gp.c3w * (double)d2 - gp.c2w));
                }
            } else {
                const T dd = s1[r] - s2[col];          // d == 1
                acc[0] += coef * (double)periodic_entry<T>(GPX_DK_DH, dd, (T)gp.h, (T)gp.w, (T)gp.p);
                acc[1] += coef * (double)periodic_entry<T>(GPX_DK_DW, dd, (T)gp.h, (T)gp.w, (T)gp.p);
                acc[2] += coef * (double)periodic_entry<T>(GPX_DK_DP, dd, (T)gp.h, (T)gp.w, (T)gp.p);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double v = acc[q];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((tid & 63) == 0) red[tid >> 6][q] = v;
    }
    __syncthreads();
    if (tid < 4) partial[(int64_t)blockIdx.x * 4 + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

// partial: GR_BLOCKS * 4 doubles of DEVICE memory; out4: host, [P0, P1, P2, trace W]
int dloglh_reduce(int dtype, int kernel, const void *x, int64_t n, int d, const double *params,
                  const void *alpha, const void *W, int64_t ldw, double *partial_dev, double *out4,
                  hipStream_t st)
{
    GradParams gpar;
    memset(&gpar, 0, sizeof(gpar));
    gpar.kernel = kernel;
    if (kernel == GPX_KERNEL_GAUSSIAN) {
        const double h = params[0], w = params[1], S = sqrt(2.0 / M_PI);
        gpar.nkp = 2;
        gpar.c1 = -0.5 / (w * w);
        gpar.c2h = S * h / w;                          // gaussian_c.pyx:61
        gpar.c2w = 0.5 * S * h * h / (w * w);          // :82
        gpar.c3w = 0.5 * S * h * h / pow(w, 4);        // :83
    } else {
        if (d != 1) { set_error("periodic gradient needs d == 1"); return GPX_ERR_UNSUPPORTED; }
        gpar.nkp = 3; gpar.h = params[0]; gpar.w = params[1]; gpar.p = params[2];
    }
    const size_t es = esize(dtype);
    const size_t smem = ((size_t)2 * GR_T * d + 2 * GR_T) * es;
    const int64_t ntr = cdiv(n, GR_T);
    const int blocks = (int)std::min<int64_t>(GR_BLOCKS, ntr * (ntr + 1) / 2);
    GPX_HIP(hipMemsetAsync(partial_dev, 0, (size_t)GR_BLOCKS * 4 * sizeof(double), st));
    if (dtype == GPX_F64) {
        if (smem > 48 * 1024)
            GPX_HIP(hipFuncSetAttribute((const void *)dloglh_reduce_kernel<double>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL((dloglh_reduce_kernel<double>), dim3(blocks), dim3(256), smem, st, (const double *)x, n,
                           d, (const double *)alpha, (const double *)W, ldw, gpar, ntr, partial_dev);
    } else {
        if (smem > 48 * 1024)
            GPX_HIP(hipFuncSetAttribute((const void *)dloglh_reduce_kernel<float>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL((dloglh_reduce_kernel<float>), dim3(blocks), dim3(256), smem, st, (const float *)x, n, d,
                           (const float *)alpha, (const float *)W, ldw, gpar, ntr, partial_dev);
    }
    GPX_LAUNCH_CHECK();
    std::vector<double> host((size_t)GR_BLOCKS * 4);
    GPX_HIP(hipMemcpyAsync(host.data(), partial_dev, host.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    GPX_HIP(hipStreamSynchronize(st));
    for (int q = 0; q < 4; ++q) {
        double v = 0.0;
        for (int b = 0; b < GR_BLOCKS; ++b) v += host[(size_t)b * 4 + q];
        out4[q] = v;
    }
    return GPX_OK;
}

}  // namespace gpx

using namespace gpx;

extern "C" {

int gpx_d_kmat(int dtype, int kernel, int member, const void *x1, int64_t n, const void *x2,
               int64_t m, int d, const double *params, double diag_add, int tri, void *out,
               int64_t ld, void *stream)
{
    GPX_TRY(ensure_device());
    GPX_ARG(dtype == GPX_F64 || dtype == GPX_F32, "dtype must be GPX_F64 or GPX_F32");
    GPX_ARG(n >= 0 && m >= 0 && d >= 1, "need n, m >= 0 and d >= 1");
    GPX_ARG(ld >= m, "ld < m");
    GPX_ARG(tri == GPX_FULL || tri == GPX_LOWER, "tri must be GPX_FULL or GPX_LOWER");
    if (n == 0 || m == 0) return GPX_OK;
    GPX_ARG(x1 && x2 && out, "NULL pointer");
    KParams kp;
    GPX_TRY(make_kparams(kernel, member, params, diag_add, &kp));
    if (dtype == GPX_F64) return launch_kmat<double>(x1, n, x2, m, d, kp, tri, out, ld, S(stream));
    return launch_kmat<float>(x1, n, x2, m, d, kp, tri, out, ld, S(stream));
}

int gpx_d_mean_member(int dtype, int kernel, int member, const void *xo, int64_t m, const void *x,
                      int64_t n, int d, const double *params, const void *alpha, void *out, void *stream)
{
    gpx::StreamTurn turn__((hipStream_t)stream);     // (this thread's scratch buffers: one stream at a time, gpx_common.h)
    GPX_TRY(ensure_device());
    GPX_ARG(dtype == GPX_F64 || dtype == GPX_F32, "dtype must be GPX_F64 or GPX_F32");
    GPX_ARG(n >= 0 && m >= 0 && d >= 1, "need n, m >= 0 and d >= 1");
    if (m == 0) return GPX_OK;
    GPX_ARG(xo && out && (n == 0 || (x && alpha)), "NULL pointer");
    KParams kp;
    GPX_TRY(make_kparams(kernel, member, params, 0.0, &kp));
    if (dtype == GPX_F64) return launch_mean<double>(kernel, xo, m, x, n, d, kp, alpha, out, S(stream));
    return launch_mean<float>(kernel, xo, m, x, n, d, kp, alpha, out, S(stream));
}

int gpx_d_mean(int dtype, int kernel, const void *xo, int64_t m, const void *x, int64_t n, int d,
               const double *params, const void *alpha, void *out, void *stream)
{
    gpx::StreamTurn turn__((hipStream_t)stream);     // (this thread's scratch buffers: one stream at a time, gpx_common.h)
    return gpx_d_mean_member(dtype, kernel, GPX_K, xo, m, x, n, d, params, alpha, out, stream);
}

}  // extern "C"
