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
// ONE batched launch (trinv64_kernel); the solve then advances 512 columns per
// launch (trsv_fwd_fused / trsv_bwd_fused below).
#include "gpx_common.h"
#include "gpx_leaf.h"

namespace gpx {

constexpr int SB = 64;

// ---- batched inverse of the 64 x 64 diagonal blocks --------------------------
// One 256-thread workgroup per block (all blocks in one launch): the register-resident 4 x 4-tile sweep of the
// factorisation's leaf in its "L is given" mode (gpx_leaf.h) carries X = inv(L_jj) -- 16 steps of two barriers,
// ~10 us per block however many there are (the earlier one-lane-per-column forward substitution through LDS took
// 69 us for the 128 blocks of n = 8192 and 1.16 ms for the 1024 of n = 65536).  Blocks shorter than 64 are
// padded with the identity.  Outputs (each optional): `out` W[blk][64][64] laid out for coalesced mat-vec
// reads, transposed != 0: out[i * 64 + c] = X[c][i] (forward sweep, z = X v), else out[i * 64 + c] = X[i][c]
// (backward, a = X^T v); W / Wt: the diagonal 64-blocks of the 512-block operators (X and X^T).
template <typename T>
__global__ __launch_bounds__(256) void inv64_kernel(const T *__restrict__ L, int64_t ldl, int64_t ncols,
                                                    T *__restrict__ out, int transposed, T *__restrict__ W,
                                                    T *__restrict__ Wt, int64_t bsL, int64_t bsOut)
{
    L += (int64_t)blockIdx.y * bsL;                  // batched: matrix blockIdx.y
    const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
    const int64_t k0 = (int64_t)blockIdx.x * SB;
    const int jb = (int)min((int64_t)SB, ncols - k0);
    const T *blk = L + k0 * ldl + k0;
    T a[4][4], x[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = 4 * tr + r, col = 4 * tc + c;
            T v = (row == col) ? (T)1 : (T)0;
            if (row < jb && col <= row) v = blk[(int64_t)row * ldl + col];
            a[r][c] = v;
            x[r][c] = (row == col) ? (T)1 : (T)0;
        }
    factor64<T, true, true>(a, x, jb, 0, nullptr);
    T *o = out ? out + (int64_t)blockIdx.y * bsOut + (int64_t)blockIdx.x * (SB * SB) : nullptr;
    const int64_t kb = blockIdx.x >> 3, pp = blockIdx.x & 7;
    T *w = W ? W + kb * (int64_t)(512 * 512) + pp * SB * (512 + 1) : nullptr;
    T *wt = Wt ? Wt + kb * (int64_t)(512 * 512) + pp * SB * (512 + 1) : nullptr;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = 4 * tr + r, col = 4 * tc + c;
            const T v = (col <= row) ? x[r][c] : (T)0;
            if (o) o[transposed ? col * SB + row : row * SB + col] = v;
            if (w) { w[(int64_t)row * 512 + col] = v; wt[(int64_t)col * 512 + row] = v; }
        }
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

// ---- fused block steps of the single-right-hand-side solves ----------------
// The solve advances in blocks of TB = 512 columns, ONE launch per block (the chain
// of dependent launches, not bandwidth, bounded the old 64-wide stepping: 2 x 1024
// launches of ~11 us at n = 65536).  In the launch that follows the solution of
// block p:
//   workgroup 0     applies block p to the rows (forward) / columns (backward) of
//                   the NEXT block only -- a TB x TB tile -- and then solves that
//                   block: 64-wide sub-steps, z = inv(L_ss) v by a 64 x 64 mat-vec
//                   with the precomputed inverse, right-looking update inside the
//                   block through LDS;
//   workgroups 1..  stream the rest of block p's panel (everything beyond the next
//                   block) against the same solution -- the HBM-bound part.
// Nothing a later block needs is ever produced by two workgroups of one launch, so
// launch order is the only synchronisation.
constexpr int TB = 512;
constexpr int TBT = 512;          // threads per workgroup
constexpr int TCH = TB / 128;     // 128-column chunks of a block row (2 per lane and chunk)

template <typename T>
__device__ __forceinline__ void load2(const T *p, bool aligned, T &v0, T &v1)
{
    if (aligned) {
        if constexpr (sizeof(T) == 8) {
            const double2 t = *reinterpret_cast<const double2 *>(p);
            v0 = t.x; v1 = t.y;
        } else {
            const float2 t = *reinterpret_cast<const float2 *>(p);
            v0 = t.x; v1 = t.y;
        }
    } else {
        v0 = p[0]; v1 = p[1];
    }
}

// sum over the 16 lanes of a DPP row, result in every lane of the row: two quad
// butterflies, then row_half_mirror and row_mirror (VALU only, no LDS traffic)
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <typename T> __device__ __forceinline__ T row16_sum(T v)
{
    v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);     // row_half_mirror
    v += dpp_mov<0x140>(v);     // row_mirror
    return v;
}
template <typename T> __device__ __forceinline__ T lanes32_sum(T v)
{
    v = row16_sum(v);
    return v + __shfl_xor(v, 16, 64);
}
template <typename T> __device__ __forceinline__ T lanes64_sum(T v)
{
    v = lanes32_sum(v);
    return v + __shfl_xor(v, 32, 64);
}

// out[i] = sum_c L[r + i, p0 + c] * z[c], i < RU, for one wave: lane owns columns
// 128 j + 2 lane (+1); all RU * TCH loads are issued before the first use.
// FULL: the block has all TB columns (no column guards).
template <typename T, int RU, bool FULL>
__device__ __forceinline__ void rows_dot(const T *__restrict__ L, int64_t ldl, int64_t p0, int pjb,
                                         int64_t r, int64_t rlast, const T (&zr)[TCH][2], int lane,
                                         bool aligned, T (&out)[RU])
{
    T l[RU][TCH][2];
#pragma unroll
    for (int i = 0; i < RU; ++i) {
        const T *row = L + min(r + i, rlast) * ldl + p0;          // clamped: in bounds, result unused
#pragma unroll
        for (int j = 0; j < TCH; ++j) {
            const int c = j * 128 + 2 * lane;
            if (FULL) {
                load2(row + c, aligned, l[i][j][0], l[i][j][1]);
            } else {
                l[i][j][0] = (T)0; l[i][j][1] = (T)0;
                if (c + 1 < pjb) load2(row + c, aligned, l[i][j][0], l[i][j][1]);
                else if (c < pjb) l[i][j][0] = row[c];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < RU; ++i) {
        T acc = (T)0;
#pragma unroll
        for (int j = 0; j < TCH; ++j) { acc = fma(l[i][j][0], zr[j][0], acc); acc = fma(l[i][j][1], zr[j][1], acc); }
        out[i] = lanes64_sum(acc);
    }
}

template <typename T> __device__ __forceinline__ T lanes8_sum(T v)
{
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    return v;
}

// z = X v (forward) / a = X^T v (backward) for one 64-block by the whole workgroup:
// lane = output row, wave w sums columns 8 w .. 8 w + 7 (li = that slice of the stored
// inverse, already in registers), fixed-order reduction over the 8 waves.
template <typename T>
__device__ __forceinline__ void inv_matvec(const T (&li)[8], const T *sv_s, T *red, T *out64, T *xout,
                                           int nvalid, int tid, int lane, int wave)
{
    T acc = (T)0;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) acc = fma(li[cc], sv_s[wave * 8 + cc], acc);
    red[wave * SB + lane] = acc;
    __syncthreads();
    if (tid < SB) {
        T z = red[tid];
#pragma unroll
        for (int q = 1; q < 8; ++q) z += red[q * SB + tid];
        out64[tid] = z;
        if (tid < nvalid) xout[tid] = z;
    }
    __syncthreads();
}

// far part of the forward sweep: apply the solved block [p0, p0 + pjb) to 64 rows starting at
// far0 + 64 * slab:  b[r] -= L[r, p0:p0+pjb] . x[p0:p0+pjb]   (whole workgroup of TBT threads; szp: TB + 2 of LDS)
template <typename T>
__device__ __forceinline__ void far_fwd(const T *__restrict__ L, int64_t ldl, T *__restrict__ b,
                                        const T *__restrict__ x, int64_t n, int64_t p0, int pjb, int64_t far0,
                                        int slab, bool aligned, T *szp)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < TB + 2; i += TBT) szp[i] = (i < pjb) ? x[p0 + i] : (T)0;
    __syncthreads();
    T zr[TCH][2];
#pragma unroll
    for (int j = 0; j < TCH; ++j) { zr[j][0] = szp[j * 128 + 2 * lane]; zr[j][1] = szp[j * 128 + 2 * lane + 1]; }
    const int64_t rbeg = far0 + (int64_t)slab * 64, rend = min(n, rbeg + 64);
    constexpr int RU = 8;
    for (int64_t r = rbeg + wave * RU; r < rend; r += (TBT / 64) * RU) {
        T acc[RU];
        if (pjb == TB) rows_dot<T, RU, true>(L, ldl, p0, pjb, r, rend - 1, zr, lane, aligned, acc);
        else rows_dot<T, RU, false>(L, ldl, p0, pjb, r, rend - 1, zr, lane, aligned, acc);
        if (lane < RU && r + lane < rend) {
            T mine = acc[0];
#pragma unroll
            for (int i = 1; i < RU; ++i) mine = (lane == i) ? acc[i] : mine;
            b[r + lane] -= mine;
        }
    }
}

// forward.  x[p0:p0+pjb] (block p) is final.  Workgroup 0 solves block [k0, k0+jb)
// from b (nothing when jb == 0); workgroups w >= 1 apply block p to 64 rows each of
// [far0, n):  b[r] -= L[r, p0:p0+pjb] . x[p0:p0+pjb].
template <typename T>
__global__ __launch_bounds__(TBT) void trsv_fwd_fused(const T *__restrict__ L, int64_t ldl,
                                                      const T *__restrict__ Linv, T *__restrict__ b,
                                                      T *__restrict__ x, int64_t n, int64_t k0, int jb,
                                                      int64_t p0, int pjb, int64_t far0, int aligned_i, int ablate,
                                                      int64_t bsL, int64_t bsLinv, int64_t bsv)
{
    L += (int64_t)blockIdx.y * bsL; Linv += (int64_t)blockIdx.y * bsLinv;    // batched: system blockIdx.y
    b += (int64_t)blockIdx.y * bsv; x += (int64_t)blockIdx.y * bsv;
    __shared__ T szp[TB + 2];       // far workgroups: solution of block p, zero padded
    __shared__ T sv[TB];            // workgroup 0: right-hand side of the block being solved
    __shared__ T szb[TB];           // workgroup 0: solution of the block so far
    __shared__ T red[8 * SB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool aligned = aligned_i != 0;

    if (blockIdx.x != 0) {
        if (pjb <= 0 || (ablate & 4)) return;
        far_fwd<T>(L, ldl, b, x, n, p0, pjb, far0, (int)blockIdx.x - 1, aligned, szp);
        return;
    }
    if (jb == 0) return;

    // in-block chain, LEFT-looking over 64-wide sub-blocks: sub-block s subtracts
    // L[rows of s, columns 0 .. 64 s) . z[0 .. 64 s) -- 8 lanes per row, a lane takes 2 of
    // every 16 columns, so each row is read as whole 128-byte lines -- and then forms
    // z_s = inv(L_ss) v_s.  The row block of sub-step s+1 and its inverse are requested as
    // soon as the registers of sub-step s are consumed, ahead of the mat-vec and its barriers.
    constexpr int CG = TB / SB - 1;                     // 7 groups of 4 loads at most
    const int ns = (jb + SB - 1) / SB;
    const int row = tid >> 3, j8 = tid & 7;
    const T *Lblk = L + k0 * ldl + k0;
    const T *Li0 = Linv + (k0 >> 6) * (int64_t)(SB * SB);
    for (int i = tid; i < TB; i += TBT) sv[i] = (i < jb) ? b[k0 + i] : (T)0;
    T li[8], ln[8];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) li[cc] = Li0[(wave * 8 + cc) * SB + lane];
    T l0[CG][4], l1[CG][4];
    __syncthreads();
    for (int s = 0; s < ns; ++s) {
        if (s > 0 && !(ablate & 2)) {
            T acc = (T)0;
#pragma unroll
            for (int t = 0; t < CG; ++t) {
                if (t < s) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int c = t * SB + m * 16 + 2 * j8;
                        acc = fma(l0[t][m], szb[c], acc);
                        acc = fma(l1[t][m], szb[c + 1], acc);
                    }
                }
            }
            acc = lanes8_sum(acc);
            if (j8 == 0 && s * SB + row < jb) sv[s * SB + row] -= acc;
        }
        if (s + 1 < ns) {
            const T *Li = Li0 + (s + 1) * (int64_t)(SB * SB);
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) ln[cc] = Li[(wave * 8 + cc) * SB + lane];
            if (!(ablate & 2)) {
                const T *rp = Lblk + (int64_t)min((s + 1) * SB + row, jb - 1) * ldl + 2 * j8;
#pragma unroll
                for (int t = 0; t < CG; ++t) {
                    if (t <= s) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) load2(rp + t * SB + m * 16, aligned, l0[t][m], l1[t][m]);
                    }
                }
            }
        }
        __syncthreads();
        inv_matvec(li, sv + s * SB, red, szb + s * SB, x + k0 + s * SB, jb - s * SB, tid, lane, wave);
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) li[cc] = ln[cc];
    }
}

// far part of the backward sweep: apply the solved row block [q0, q0 + qjb) to CW columns starting at
// c0 + CW * slab:  b[c] -= L[q0:q0+qjb, c] . x[q0:q0+qjb]   (sa: TB, red: 2 * TB of LDS)
template <typename T, int CW>
__device__ __forceinline__ void far_bwd(const T *__restrict__ L, int64_t ldl, T *__restrict__ b,
                                        const T *__restrict__ x, int64_t q0, int qjb, int64_t c0, int64_t c1,
                                        int slab, bool aligned, T *sa, T *red)
{
    const int tid = threadIdx.x;
        constexpr int LP = CW / 2;                  // lanes (column pairs) per row
        constexpr int NG = TBT / LP;                // row groups
        constexpr int PER = TB / NG;                // rows per thread
        constexpr int BU = PER < 32 ? PER : 32;     // loads in flight per thread and batch
        for (int i = tid; i < TB; i += TBT) sa[i] = (i < qjb) ? x[q0 + i] : (T)0;
        __syncthreads();
        const int64_t cbeg = c0 + (int64_t)slab * CW, cend = min(c1, cbeg + CW);
        const int lp = tid % LP, g = tid / LP;
        const int64_t c = min(cbeg + 2 * lp, cend - 1);
        const bool pair = c + 1 < cend;
        const T *col = L + q0 * ldl + c;
        T a0 = (T)0, a1 = (T)0;
#pragma unroll 1
        for (int u0 = 0; u0 < PER; u0 += BU) {
            T l0[BU], l1[BU];
#pragma unroll
            for (int u = 0; u < BU; ++u) {
                const int i = min(g + (u0 + u) * NG, qjb - 1);
                if (pair) load2(col + (int64_t)i * ldl, aligned, l0[u], l1[u]);
                else { l0[u] = col[(int64_t)i * ldl]; l1[u] = (T)0; }
            }
#pragma unroll
            for (int u = 0; u < BU; ++u) {
                const int i = g + (u0 + u) * NG;
                const T av = (i < qjb) ? sa[i] : (T)0;
                a0 = fma(l0[u], av, a0);
                a1 = fma(l1[u], av, a1);
            }
        }
        red[g * CW + 2 * lp] = a0;
        red[g * CW + 2 * lp + 1] = a1;
        __syncthreads();
        if (tid < CW && cbeg + tid < cend) {
            T sum = red[tid];
            for (int q = 1; q < NG; ++q) sum += red[q * CW + tid];
            b[cbeg + tid] -= sum;
        }
}

// backward (L^T a = v).  x[q0:q0+qjb] (block q, below the block being solved) is final.
// Workgroup 0 solves block [k0, k0+jb) from b (nothing when jb == 0); workgroups
// w >= 1 apply block q to CW columns each of [c0, c1):  b[c] -= L[q0:q0+qjb, c] . x[q0:q0+qjb].
// CW = 128 for the streaming far part (1 KiB per row and wave), 32 for the 512 x 512 tile
// next to the diagonal (16 workgroups instead of 4).
template <typename T, int CW>
__global__ __launch_bounds__(TBT) void trsv_bwd_fused(const T *__restrict__ L, int64_t ldl,
                                                      const T *__restrict__ Linv, T *__restrict__ b,
                                                      T *__restrict__ x, int64_t k0, int jb, int64_t q0,
                                                      int qjb, int64_t c0, int64_t c1, int aligned_i, int ablate,
                                                      int64_t bsL, int64_t bsLinv, int64_t bsv)
{
    L += (int64_t)blockIdx.y * bsL; Linv += (int64_t)blockIdx.y * bsLinv;
    b += (int64_t)blockIdx.y * bsv; x += (int64_t)blockIdx.y * bsv;
    __shared__ T sa[TB];            // far workgroups: solution of block q
    __shared__ T sv[TB];
    __shared__ T sz[SB];
    __shared__ T red[2 * TB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool aligned = aligned_i != 0;

    if (blockIdx.x != 0) {
        if (qjb <= 0 || (ablate & 4)) return;
        far_bwd<T, CW>(L, ldl, b, x, q0, qjb, c0, c1, (int)blockIdx.x - 1, aligned, sa, red);
        return;
    }
    if (jb == 0) return;

    // in-block chain, last sub-block first, right-looking: after a_s = inv(L_ss)^T v_s the
    // rows of sub-block s update every column to their left (thread = column pair, two
    // groups of 32 rows: whole rows are read contiguously).  The tile of a sub-step is
    // requested before the mat-vec that produces its multipliers.
    const int ns = (jb + SB - 1) / SB;
    const int cpair = tid & 255, g = tid >> 8;
    const T *Li0 = Linv + (k0 >> 6) * (int64_t)(SB * SB);
    for (int i = tid; i < TB; i += TBT) sv[i] = (i < jb) ? b[k0 + i] : (T)0;
    T li[8];
    __syncthreads();
    for (int s = ns - 1; s >= 0; --s) {
        const T *Li = Li0 + s * (int64_t)(SB * SB);
        const int ncol = s * SB;
        const int ilim = min(SB, jb - s * SB);
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) li[cc] = Li[(wave * 8 + cc) * SB + lane];
        T l0[32], l1[32];
        const bool act = 2 * cpair < ncol && !(ablate & 2);     // ncol is a multiple of 64: always a pair
        if (act) {
            const T *col = L + (k0 + s * SB) * ldl + k0 + 2 * cpair;
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int i = min(g * 32 + u, ilim - 1);
                load2(col + (int64_t)i * ldl, aligned, l0[u], l1[u]);
            }
        }
        inv_matvec(li, sv + s * SB, red, sz, x + k0 + s * SB, jb - s * SB, tid, lane, wave);
        if (ncol > 0) {
            T a0 = (T)0, a1 = (T)0;
            if (act) {
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    const int i = g * 32 + u;
                    const T av = (i < ilim) ? sz[i] : (T)0;     // rows >= ilim: identity padding, unused
                    a0 = fma(l0[u], av, a0);
                    a1 = fma(l1[u], av, a1);
                }
            }
            red[g * TB + 2 * cpair] = a0;
            red[g * TB + 2 * cpair + 1] = a1;
            __syncthreads();
            if (tid < ncol) sv[tid] -= red[tid] + red[TB + tid];
            __syncthreads();
        }
    }
}

// ---- operator form of the block steps ---------------------------------------------------------------
// The sweep above needs two dependent launches per 512-column block: the near tile (previous block's solution
// into this block's rows) and the in-block chain (8 sub-steps of a 64-wide mat-vec by ONE workgroup, 19 - 31 us:
// at n = 8192 that chain, not bandwidth, is the whole solve).  With per-block operators, precomputed once per
// factor on the MFMA kernel,
//     W_k  = inv(L_kk)             (512 x 512, from the 64 x 64 inverses by recursive doubling, batched over k)
//     Tf_k = W_k L_{k,k-1}         forward :  x_k = W_k  w_k - Tf_k x_{k-1}
//     Tb_k = W_k^T L_{k+1,k}^T     backward:  a_k = W_k^T z_k - Tb_k a_{k+1}
// a block step is ONE launch with no dependency inside it: 8 workgroups do the two 512-wide mat-vecs of block k
// (64 rows each, 8 lanes per row, whole 128-byte lines), all others stream the far panel of the neighbour block
// exactly as before.  w_k / z_k already hold every far contribution (blocks two or more away were streamed by
// earlier launches); the neighbour's contribution comes through Tf / Tb.  A ragged last block (n % 512) keeps
// the old kernels.  Cost of the operators: 2 (n / 512) products of 512^3 + the doubling, ~1.3 ms at n = 65536.
constexpr int OB = 512;

// LT_k[j][m] = L[(OB k + m), OB (k - 1) + j]   for k = 1 .. nfull - 1  (grid: (OB / 64)^2 tiles, k - 1)
template <typename T>
__global__ __launch_bounds__(256) void transpose_subdiag_kernel(const T *__restrict__ L, int64_t ldl, T *__restrict__ LT)
{
    __shared__ T tile[64][65];
    const int64_t k = blockIdx.y + 1;
    const int tm = blockIdx.x >> 3, tj = blockIdx.x & 7;            // tile of rows m (of L), columns j
    const T *src = L + (k * OB + tm * 64) * ldl + (k - 1) * OB + tj * 64;
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        tile[r][c] = src[(int64_t)r * ldl + c];
    }
    __syncthreads();
    T *dst = LT + k * (int64_t)(OB * OB) + (int64_t)(tj * 64) * OB + tm * 64;
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        dst[(int64_t)r * OB + c] = tile[c][r];
    }
}

// one block step: workgroups [0, nchain) solve block k (rows k0 ..), the others stream the neighbour's far panel
// PARTS lanes per row: 8 (64 rows per workgroup, 8 workgroups a block) or 32 (16 rows, 32 workgroups -- a workgroup
// streams 128 KB of operators instead of 512: the block's two mat-vecs are bound by what ONE CU can pull)
template <typename T, bool FWD, int PARTS>
__global__ __launch_bounds__(TBT) void trsv_op_kernel(const T *__restrict__ Wk, const T *__restrict__ Tk,
                                                      const T *__restrict__ L, int64_t ldl, T *__restrict__ rhs,
                                                      T *__restrict__ x, int64_t n, int64_t k0, int64_t p0, int pjb,
                                                      int64_t f0, int64_t f1, int nchain, int aligned_i)
{
    __shared__ T szp[TB + 2];
    __shared__ T sv[TB];
    __shared__ T red[2 * TB];
    const int tid = threadIdx.x;
    const bool aligned = aligned_i != 0;
    if ((int)blockIdx.x >= nchain) {
        if (pjb <= 0) return;
        if (FWD) far_fwd<T>(L, ldl, rhs, x, n, p0, pjb, f0, (int)blockIdx.x - nchain, aligned, szp);
        else far_bwd<T, 128>(L, ldl, rhs, x, p0, pjb, f0, f1, (int)blockIdx.x - nchain, aligned, szp, red);
        return;
    }
    for (int i = tid; i < TB; i += TBT) {
        sv[i] = rhs[k0 + i];
        szp[i] = (Tk && i < pjb) ? x[p0 + i] : (T)0;
    }
    __syncthreads();
    constexpr int ROWS = TBT / PARTS;                 // rows per workgroup
    const int r = tid / PARTS, part = tid % PARTS;
    const int row = ROWS * (int)blockIdx.x + r;
    const T *wrow = Wk + (int64_t)row * OB, *trow = Tk ? Tk + (int64_t)row * OB : nullptr;
    T acc = (T)0;
    // branch-free on purpose (the zero half of the triangular W_k is read too): with a per-row triangle test
    // the loads cannot be batched and every iteration pays a memory round trip (measured 30 us instead of 8)
    if (trow) {
#pragma unroll 8
        for (int i = 0; i < OB / (2 * PARTS); ++i) {
            const int c = 2 * PARTS * i + 2 * part;
            T w0, w1, t0, t1;
            load2(wrow + c, true, w0, w1);
            load2(trow + c, true, t0, t1);
            acc = fma(w0, sv[c], acc);
            acc = fma(w1, sv[c + 1], acc);
            acc = fma(-t0, szp[c], acc);
            acc = fma(-t1, szp[c + 1], acc);
        }
    } else {
#pragma unroll 8
        for (int i = 0; i < OB / (2 * PARTS); ++i) {
            const int c = 2 * PARTS * i + 2 * part;
            T w0, w1;
            load2(wrow + c, true, w0, w1);
            acc = fma(w0, sv[c], acc);
            acc = fma(w1, sv[c + 1], acc);
        }
    }
    acc = PARTS == 8 ? lanes8_sum(acc) : lanes32_sum(acc);
    if (part == 0) x[k0 + row] = acc;
}

// thread-local, grow-only home of the operators when the caller brings no cache of its own
struct OpsScratch { void *p = nullptr; size_t bytes = 0; int device = -1; };
static thread_local OpsScratch g_ops;
static int ops_scratch(size_t bytes, void **out)
{
    int dev = 0;
    GPX_HIP(hipGetDevice(&dev));
    if (g_ops.device != dev || g_ops.bytes < bytes) {
        if (g_ops.p && g_ops.device == dev) { GPX_HIP(hipDeviceSynchronize()); (void)hipFree(g_ops.p); }
        g_ops.p = nullptr; g_ops.bytes = 0; g_ops.device = dev;
        GPX_HIP(hipMalloc(&g_ops.p, bytes));
        g_ops.bytes = bytes;
    }
    *out = g_ops.p;
    return GPX_OK;
}

size_t trsv_ops_bytes(int dtype, int64_t n)
{
    const int64_t nfull = n / OB;
    return (size_t)(5 * nfull * (int64_t)OB * OB + nfull * 8 * (int64_t)SB * SB) * esize(dtype) + 256;
}

// build W, Wt, Tf, Tb of the 512-blocks [kbeg, kend) of L into `buf` (trsv_ops_bytes): everything is batched over the
// blocks of the range with the pointers moved to its first block.  Block k needs L_kk, L_{k,k-1} and L_{k+1,k}: block
// columns <= k of the factor.  kbeg = 0 also clears the triangular operators' zero halves for ALL blocks, so ranges go
// out in increasing order.
template <typename T>
static int trsv_ops_prepare(const T *L, int64_t n, int64_t ldl, void *buf, hipStream_t st, int dtype, int64_t kbeg = 0,
                            int64_t kend = -1)
{
    const int64_t nfull = n / OB, rag = n - nfull * OB;
    if (kend < 0 || kend > nfull) kend = nfull;
    const int64_t cnt = kend - kbeg;
    const int64_t BS = (int64_t)OB * OB;
    T *W = (T *)buf, *Wt = W + nfull * BS, *P = Wt + nfull * BS, *Tf = P + nfull * BS, *Tb = Tf + nfull * BS;
    if (kbeg == 0) GPX_HIP(hipMemsetAsync(W, 0, (size_t)2 * nfull * BS * sizeof(T), st));
    if (cnt <= 0) return GPX_OK;
    const int64_t dLk = (int64_t)OB * (ldl + 1);                     // L_kk -> L_{k+1,k+1}
    const T *Lk = L + kbeg * dLk;
    T *Wk = W + kbeg * BS, *Wtk = Wt + kbeg * BS, *Pk = P + kbeg * BS;
    hipLaunchKernelGGL((inv64_kernel<T>), dim3((unsigned)(cnt * 8)), dim3(256), 0, st, Lk, ldl, cnt * OB, (T *)nullptr, 0,
                       Wk, Wtk, (int64_t)0, (int64_t)0);
    GPX_LAUNCH_CHECK();
    for (int64_t s2 = SB; s2 < OB; s2 *= 2) {
        Batch b;
        b.count = (int)cnt; b.count2 = (int)(OB / (2 * s2));
        const int64_t tW = 2 * s2 * OB + 2 * s2, tL = 2 * s2 * ldl + 2 * s2;
        // Pt = W11^T L21^T
        b.sA = BS; b.sB = dLk; b.sC = BS; b.tA = tW; b.tB = tL; b.tC = tW;
        GPX_TRY(gemm_nt(dtype, s2, s2, s2, Wtk, OB, Lk + s2 * ldl, ldl, Pk, OB, 1.0, GPX_FULL, 0, 0, st, 1, 0, &b));
        // W21 = -W22 Pt^T
        b.sA = BS; b.sB = BS; b.sC = BS; b.tA = tW; b.tB = tW; b.tC = tW;
        GPX_TRY(gemm_nt(dtype, s2, s2, s2, Wk + s2 * OB + s2, OB, Pk, OB, Wk + s2 * OB, OB, -1.0, GPX_FULL, 0, 0, st, 1, 0, &b));
        // (W^T)12 = -Pt W22^T
        GPX_TRY(gemm_nt(dtype, s2, s2, s2, Pk, OB, Wk + s2 * OB + s2, OB, Wtk + s2, OB, -1.0, GPX_FULL, 0, 0, st, 1, 0, &b));
    }
    const int64_t ka = std::max<int64_t>(kbeg, 1);                   // Tf_k = W_k L_{k,k-1} = W_k LT_k^T,  k = ka .. kend - 1
    if (kend > ka) {
        // (the transpose kernel numbers its blocks from 1 relative to the base it is given)
        hipLaunchKernelGGL((transpose_subdiag_kernel<T>), dim3(64, (unsigned)(kend - ka)), dim3(256), 0, st, L + (ka - 1) * dLk, ldl,
                           P + (ka - 1) * BS);
        GPX_LAUNCH_CHECK();
        Batch b;
        b.count = (int)(kend - ka);
        b.sA = BS; b.sB = BS; b.sC = BS;
        GPX_TRY(gemm_nt(dtype, OB, OB, OB, W + ka * BS, OB, P + ka * BS, OB, Tf + ka * BS, OB, 1.0, GPX_FULL, 0, 0, st, 1, 0, &b));
    }
    const int64_t kb_end = std::min(kend, nfull - 1);                // Tb_k = W_k^T L_{k+1,k}^T,  k = kbeg .. kb_end - 1
    if (kb_end > kbeg) {
        Batch b;
        b.count = (int)(kb_end - kbeg);
        b.sA = BS; b.sB = dLk; b.sC = BS;
        GPX_TRY(gemm_nt(dtype, OB, OB, OB, Wtk, OB, Lk + (int64_t)OB * ldl, ldl, Tb + kbeg * BS, OB, 1.0, GPX_FULL, 0, 0, st, 1, 0, &b));
    }
    if (rag > 0 && kend == nfull) {                                  // the last full block against the ragged one
        T *tb = Tb + (nfull - 1) * BS;
        GPX_HIP(hipMemsetAsync(tb, 0, (size_t)BS * sizeof(T), st));
        GPX_TRY(gemm_nt(dtype, OB, rag, OB, Wt + (nfull - 1) * BS, OB, L + nfull * OB * ldl + (nfull - 1) * OB, ldl, tb, OB,
                        1.0, GPX_FULL, 0, 0, st, 1, 0));
    }
    return GPX_OK;
}

static bool trsv_ops_enabled()
{
    return tune().trsv_ops != 0;
}
// below this the ~0.7 ms of operator products (11 under-filled launches) costs what the shorter steps save
// (n = 8192: 1.17 vs 1.10 ms for both sweeps; n = 16384: 1.78 vs 2.49; n = 65536: 9.7 vs 12.1)
static int64_t trsv_ops_min_n()
{
    return std::max<int64_t>(2 * OB, tune().trsv_ops_min);
}

template <typename T>
static int trsv_t(const T *L, int64_t n, int64_t ldl, T *b, T *x, int transpose, hipStream_t st,
                  int64_t ncols = -1, const Batch *bt = nullptr, TrsvOps *ops = nullptr, int dtype = GPX_F64)
{
    // bt: bt->count systems solved by the same launches; sA = stride of L, sB = stride of b and x
    const unsigned nbt = (unsigned)(bt ? bt->count : 1);
    const int64_t sL = bt ? bt->sA : 0, sv = bt ? bt->sB : 0;
    // ncols < n (forward only): the matrix is a trapezoid -- an ncols x ncols lower
    // triangle on top of (n - ncols) further rows; x[0:ncols] is solved and the
    // remaining right-hand side b[ncols:n] is reduced by L[ncols:n, 0:ncols] x.
    if (ncols < 0 || ncols > n) ncols = n;
    ProfScope prof(PC_TRSV, ((double)n * ncols - 0.5 * (double)ncols * (ncols - 1)) * sizeof(T) * nbt, st);
    const int64_t nblk = cdiv(ncols, SB);
    const int64_t sLinv = nblk * SB * SB;
    void *scr = nullptr;
    GPX_TRY(scratch((size_t)nbt * sLinv * sizeof(T), &scr));
    T *Linv = (T *)scr;
    const int aligned = (((uintptr_t)L) % (2 * sizeof(T)) == 0) && (ldl % 2 == 0);
    const int ablate = 0;                          // (timing-only ablations of the step kernels: compile-time edits now)
    const int64_t nb = cdiv(ncols, TB);
    auto width = [&](int64_t blk) { return (int)std::min<int64_t>(TB, ncols - blk * TB); };
    // operator form: square systems of at least two full blocks, aligned rows, one system
    const bool prebuilt = ops && ops->valid && ops->buf && n % OB == 0 && ops->bytes >= trsv_ops_bytes(dtype, n);
    // operators of the LEADING blocks only (built beside the factorisation, gpx_gp_fit; the caller has ordered `st` behind
    // them): a backward sweep takes the trailing blocks by steps and switches to one launch per block where they begin
    const int64_t kpart = (ops && !ops->valid && ops->buf && ops->built > 0 && n % OB == 0 && transpose && !bt && ncols == n &&
                           ops->bytes >= trsv_ops_bytes(dtype, n) && trsv_ops_enabled() && aligned &&
                           ldl % (16 / (int64_t)sizeof(T)) == 0 && ((uintptr_t)L) % 16 == 0) ? std::min(ops->built, n / OB) : 0;
    if (kpart == 0 && trsv_ops_enabled() && !bt && ncols == n && (n >= trsv_ops_min_n() || prebuilt) && aligned &&
        ldl % (16 / (int64_t)sizeof(T)) == 0 && ((uintptr_t)L) % 16 == 0) {
        const int64_t nfull = n / OB, rag = n - nfull * OB, BS = (int64_t)OB * OB;
        route_hit(RT_TRSV_OPS);
        void *buf = nullptr;
        bool fresh = true;
        if (ops) {                                                   // the caller's cache (one factor, many solves)
            if (!ops->buf || ops->bytes < trsv_ops_bytes(dtype, n)) {
                if (ops->buf) { GPX_HIP(hipStreamSynchronize(st)); (void)hipFree(ops->buf); ops->buf = nullptr; }
                GPX_HIP(hipMalloc(&ops->buf, trsv_ops_bytes(dtype, n)));
                ops->bytes = trsv_ops_bytes(dtype, n);
                ops->invalidate();
            }
            buf = ops->buf;
            fresh = !ops->valid;
        } else {
            GPX_TRY(ops_scratch(trsv_ops_bytes(dtype, n), &buf));
        }
        // (a factor whose leading blocks already have their operators: only the rest)
        if (fresh) GPX_TRY(trsv_ops_prepare<T>(L, n, ldl, buf, st, dtype, ops ? std::min(ops->built, nfull) : 0, nfull));
        if (ops) { ops->valid = true; ops->built = nfull; }
        const T *W = (const T *)buf, *Wt = W + nfull * BS, *Tf = Wt + 2 * nfull * BS, *Tb = Tf + nfull * BS;
        const int NCH = OB / (TBT / 32);        // 32 lanes per row (the 8-lane form was measured slower and went in round 6)
        if (!transpose) {
            for (int64_t k = 0; k < nfull; ++k) {
                const int64_t k0 = k * OB, far0 = k0 + OB;
                const int64_t nfar = k > 0 ? cdiv(n - far0, 64) : 0;
                hipLaunchKernelGGL((trsv_op_kernel<T, true, 32>), dim3((unsigned)(NCH + nfar)), dim3(TBT), 0, st, W + k * BS,
                                       k > 0 ? Tf + k * BS : (const T *)nullptr, L, ldl, b, x, n, k0, k0 - OB, k > 0 ? OB : 0, far0, n,
                                       NCH, aligned);
            }
            if (rag > 0) {                                           // ragged last block: the near tile, then the old chain
                const int64_t k0 = nfull * OB, p0 = k0 - OB;
                hipLaunchKernelGGL((inv64_kernel<T>), dim3((unsigned)nblk, nbt), dim3(256), 0, st, L, ldl, ncols, Linv, 1,
                                   (T *)nullptr, (T *)nullptr, sL, sLinv);
                hipLaunchKernelGGL((trsv_fwd_fused<T>), dim3((unsigned)(1 + cdiv(rag, 64))), dim3(TBT), 0, st, L, ldl, Linv, b, x,
                                   n, k0, 0, p0, OB, k0, aligned, ablate, sL, sLinv, sv);
                hipLaunchKernelGGL((trsv_fwd_fused<T>), dim3(1), dim3(TBT), 0, st, L, ldl, Linv, b, x, n, k0, (int)rag, p0, OB, n,
                                   aligned, ablate, sL, sLinv, sv);
            }
        } else {
            if (rag > 0) {
                const int64_t k0 = nfull * OB;
                hipLaunchKernelGGL((inv64_kernel<T>), dim3((unsigned)nblk, nbt), dim3(256), 0, st, L, ldl, ncols, Linv, 0,
                                   (T *)nullptr, (T *)nullptr, sL, sLinv);
                hipLaunchKernelGGL((trsv_bwd_fused<T, 128>), dim3(1), dim3(TBT), 0, st, L, ldl, Linv, b, x, k0, (int)rag, k0 + OB, 0,
                                   (int64_t)0, k0, aligned, ablate, sL, sLinv, sv);
            }
            for (int64_t k = nfull - 1; k >= 0; --k) {
                const int64_t k0 = k * OB, q0 = k0 + OB;
                const int qjb = (k + 1 < nfull) ? OB : (int)rag;
                const int64_t nfar = qjb > 0 ? cdiv(k0, 128) : 0;
                hipLaunchKernelGGL((trsv_op_kernel<T, false, 32>), dim3((unsigned)(NCH + nfar)), dim3(TBT), 0, st, Wt + k * BS,
                                       qjb > 0 ? Tb + k * BS : (const T *)nullptr, L, ldl, b, x, n, k0, q0, qjb, (int64_t)0, k0, NCH,
                                       aligned);
            }
        }
        GPX_LAUNCH_CHECK();
        return GPX_OK;
    }
    // Per block two launches: (N) the 512 x 512 tile that carries the previous block's
    // solution into this block's rows/columns, spread over 8-16 workgroups; (F) workgroup 0
    // solves the block while the other workgroups stream the previous block's far panel.
    route_hit(RT_TRSV_STEPS);
    if (!transpose) {
        hipLaunchKernelGGL((inv64_kernel<T>), dim3((unsigned)nblk, nbt), dim3(256), 0, st, L, ldl, ncols, Linv, 1,
                           (T *)nullptr, (T *)nullptr, sL, sLinv);
        for (int64_t blk = 0; blk < nb; ++blk) {
            const int64_t k0 = blk * TB, p0 = std::max<int64_t>(blk - 1, 0) * TB;
            const int jb = width(blk), pjb = blk > 0 ? TB : 0;
            if (blk > 0)
                hipLaunchKernelGGL((trsv_fwd_fused<T>), dim3((unsigned)(1 + cdiv(jb, 64)), nbt), dim3(TBT), 0, st, L, ldl,
                                   Linv, b, x, k0 + jb, k0, 0, p0, pjb, k0, aligned, ablate, sL, sLinv, sv);
            const int64_t far = blk > 0 ? n - (k0 + jb) : 0;
            hipLaunchKernelGGL((trsv_fwd_fused<T>), dim3((unsigned)(1 + cdiv(far, 64)), nbt), dim3(TBT), 0, st, L, ldl,
                               Linv, b, x, n, k0, jb, p0, pjb, k0 + jb, aligned, ablate, sL, sLinv, sv);
        }
        if (n > ncols)          // trapezoid: the last block's panel below the triangle
            hipLaunchKernelGGL((trsv_fwd_fused<T>), dim3((unsigned)(1 + cdiv(n - ncols, 64)), nbt), dim3(TBT), 0, st, L,
                               ldl, Linv, b, x, n, ncols, 0, (nb - 1) * TB, width(nb - 1), ncols, aligned, ablate,
                               sL, sLinv, sv);
    } else {
        // (kpart > 0: blocks [kpart, nb) here, the leading kpart blocks by their operators below)
        const int64_t b64 = kpart * (OB / SB);
        hipLaunchKernelGGL((inv64_kernel<T>), dim3((unsigned)(nblk - b64), nbt), dim3(256), 0, st, L + b64 * SB * (ldl + 1), ldl,
                           ncols - b64 * SB, Linv + b64 * SB * SB, 0, (T *)nullptr, (T *)nullptr, sL, sLinv);
        for (int64_t blk = nb - 1; blk >= kpart; --blk) {
            const int64_t k0 = blk * TB, q0 = k0 + TB;
            const int jb = width(blk), qjb = blk + 1 < nb ? width(blk + 1) : 0;
            if (qjb > 0)
                hipLaunchKernelGGL((trsv_bwd_fused<T, 32>), dim3((unsigned)(1 + cdiv(jb, 32)), nbt), dim3(TBT), 0, st, L,
                                   ldl, Linv, b, x, k0, 0, q0, qjb, k0, k0 + jb, aligned, ablate, sL, sLinv, sv);
            hipLaunchKernelGGL((trsv_bwd_fused<T, 128>), dim3((unsigned)(1 + (qjb > 0 ? cdiv(k0, 128) : 0)), nbt),
                               dim3(TBT), 0, st, L, ldl, Linv, b, x, k0, jb, q0, qjb, (int64_t)0, k0, aligned,
                               ablate, sL, sLinv, sv);
        }
        if (kpart > 0) {
            route_hit(RT_TRSV_OPS);
            const int64_t nfull = n / OB, BS = (int64_t)OB * OB;
            const T *W = (const T *)ops->buf, *Wt = W + nfull * BS, *Tb = Wt + 3 * nfull * BS;
            const int NCH = OB / (TBT / 32);
            for (int64_t k = kpart - 1; k >= 0; --k) {
                const int64_t k0 = k * OB, q0 = k0 + OB;
                hipLaunchKernelGGL((trsv_op_kernel<T, false, 32>), dim3((unsigned)(NCH + cdiv(k0, 128))), dim3(TBT), 0, st, Wt + k * BS,
                                       Tb + k * BS, L, ldl, b, x, n, k0, q0, OB, (int64_t)0, k0, NCH, aligned);
            }
        }
    }
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}

int trsv_lower(int dtype, const void *L, int64_t n, int64_t ldl, void *b, void *x, int transpose,
               hipStream_t st, const Batch *bt, TrsvOps *ops)
{
    if (n <= 0) return GPX_OK;
    if (dtype == GPX_F64)
        return trsv_t<double>((const double *)L, n, ldl, (double *)b, (double *)x, transpose, st, -1, bt, ops, dtype);
    return trsv_t<float>((const float *)L, n, ldl, (float *)b, (float *)x, transpose, st, -1, bt, ops, dtype);
}

int trsv_ops_build(int dtype, const void *L, int64_t n, int64_t ldl, TrsvOps *ops, hipStream_t st)
{
    if (!ops || n < OB || n % OB != 0 || !trsv_ops_enabled() || ldl % (16 / (int64_t)esize(dtype)) != 0 || ((uintptr_t)L) % 16 != 0)
        return GPX_OK;                                               // not eligible: the solve takes the step route
    const size_t need = trsv_ops_bytes(dtype, n);
    if (!ops->buf || ops->bytes < need) {
        if (ops->buf) { GPX_HIP(hipStreamSynchronize(st)); (void)hipFree(ops->buf); ops->buf = nullptr; }
        GPX_HIP(hipMalloc(&ops->buf, need));
        ops->bytes = need;
    }
    ops->invalidate();
    if (dtype == GPX_F64) GPX_TRY(trsv_ops_prepare<double>((const double *)L, n, ldl, ops->buf, st, dtype));
    else GPX_TRY(trsv_ops_prepare<float>((const float *)L, n, ldl, ops->buf, st, dtype));
    ops->valid = true;
    ops->built = n / OB;
    return GPX_OK;
}

bool trsv_ops_ahead_ok(int dtype, const void *L, int64_t n, int64_t ldl)
{
    return n >= 2 * OB && n % OB == 0 && trsv_ops_enabled() && ldl % (16 / (int64_t)esize(dtype)) == 0 && ((uintptr_t)L) % 16 == 0 &&
           ((uintptr_t)L) % (2 * esize(dtype)) == 0 && ldl % 2 == 0;
}

int trsv_ops_build_upto(int dtype, const void *L, int64_t n, int64_t ldl, TrsvOps *ops, int64_t kend, hipStream_t st)
{
    if (!ops || !trsv_ops_ahead_ok(dtype, L, n, ldl)) return GPX_OK;
    const int64_t nfull = n / OB;
    kend = std::min(kend, nfull);
    const size_t need = trsv_ops_bytes(dtype, n);
    if (ops->built == 0) {
        if (!ops->buf || ops->bytes < need) {
            if (ops->buf) { GPX_HIP(hipDeviceSynchronize()); (void)hipFree(ops->buf); ops->buf = nullptr; }
            GPX_HIP(hipMalloc(&ops->buf, need));
            ops->bytes = need;
        }
        ops->valid = false;
    }
    if (kend <= ops->built || !ops->buf) return GPX_OK;
    if (dtype == GPX_F64) GPX_TRY(trsv_ops_prepare<double>((const double *)L, n, ldl, ops->buf, st, dtype, ops->built, kend));
    else GPX_TRY(trsv_ops_prepare<float>((const float *)L, n, ldl, ops->buf, st, dtype, ops->built, kend));
    ops->built = kend;
    if (kend == nfull) ops->valid = true;
    return GPX_OK;
}

int trsv_lower_cols(int dtype, const void *L, int64_t n, int64_t ldl, int64_t ncols, void *b, void *x,
                    hipStream_t st)
{
    if (n <= 0 || ncols <= 0) return GPX_OK;
    if (dtype == GPX_F64)
        return trsv_t<double>((const double *)L, n, ldl, (double *)b, (double *)x, 0, st, ncols, nullptr, nullptr, dtype);
    return trsv_t<float>((const float *)L, n, ldl, (float *)b, (float *)x, 0, st, ncols, nullptr, nullptr, dtype);
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
// x_upper: X is upper triangular on entry (the identity, when L^-T itself is wanted): rows beyond
// the current block are still zero in its columns, so every step works on the leading k0 + kb rows
// only -- a third of the flops.
// grow-only device scratch of trsm_right_lt's operator route: one m x 512 block of X (one per host thread)
struct TrsmScratch { void *p = nullptr; size_t bytes = 0; int device = -1; };
static thread_local TrsmScratch g_trsm_scr;
static int trsm_scratch(size_t bytes, void **out)
{
    int dev = 0;
    GPX_HIP(hipGetDevice(&dev));
    if (g_trsm_scr.device != dev || g_trsm_scr.bytes < bytes) {
        if (g_trsm_scr.p && g_trsm_scr.device == dev) { GPX_HIP(hipDeviceSynchronize()); (void)hipFree(g_trsm_scr.p); }
        g_trsm_scr.p = nullptr; g_trsm_scr.bytes = 0; g_trsm_scr.device = dev;
        GPX_HIP(hipMalloc(&g_trsm_scr.p, bytes));
        g_trsm_scr.bytes = bytes;
    }
    *out = g_trsm_scr.p;
    return GPX_OK;
}

int trsm_right_lt(int dtype, const void *L, int64_t n, int64_t ldl, void *X, int64_t m, int64_t ldx,
                  hipStream_t st, int x_upper, TrsvOps *ops)
{
    if (n <= 0 || m <= 0) return GPX_OK;
    const size_t es = esize(dtype);
    // Operator route (the caller's TrsvOps of THIS factor, n a multiple of 512): the substitution inside a 512-block,
    //   X[:, blk] <- X[:, blk] inv(L_kk)^T,
    // is ONE product with W_k = inv(L_kk) (into a scratch block, copied back) instead of eight 64-wide substitutions with
    // seven small products between them -- 16 latency-bound launches a block, which were most of a posterior covariance
    // (n = 8192, m = 1024: cov 11.0 -> see DESIGN 3.3).  The operators are completed here if the factor has only some.
    if (ops && trsv_ops_ahead_ok(dtype, L, n, ldl) && tune().trsm_ops != 0 && ldx % (16 / (int64_t)es) == 0 &&
        ((uintptr_t)X) % 16 == 0) {
        const int64_t nfull = n / OB, BS = (int64_t)OB * OB;
        if (!ops->valid) GPX_TRY(trsv_ops_build_upto(dtype, L, n, ldl, ops, nfull, st));
        if (ops->valid && ops->buf) {
            route_hit(RT_TRSM_OPS);
            void *scr = nullptr;
            GPX_TRY(trsm_scratch((size_t)m * OB * es, &scr));
            const char *W = (const char *)ops->buf;
            for (int64_t k = 0; k < nfull; ++k) {
                const int64_t k0 = k * OB, r = k0 + OB;
                const int64_t me = x_upper ? std::min(m, r) : m;
                char *Xk = (char *)X + k0 * es;
                GPX_TRY(gemm_nt(dtype, me, OB, OB, Xk, ldx, W + (size_t)k * BS * es, OB, scr, OB, 1.0, GPX_FULL, 0, 0, st, 1));
                GPX_HIP(hipMemcpy2DAsync(Xk, (size_t)ldx * es, scr, (size_t)OB * es, (size_t)OB * es, (size_t)me,
                                         hipMemcpyDeviceToDevice, st));
                if (r < n)
                    GPX_TRY(gemm_nt(dtype, me, n - r, OB, Xk, ldx, (const char *)L + (r * ldl + k0) * es, ldl, (char *)X + r * es, ldx,
                                    -1.0, GPX_FULL, 0, 0, st));
            }
            return GPX_OK;
        }
    }
    const int64_t NB = n >= 8192 ? 512 : 256;
    auto Lp = [&](int64_t r, int64_t c) { return (const char *)L + (r * ldl + c) * es; };
    auto Xp = [&](int64_t c) { return (char *)X + c * es; };
    for (int64_t k0 = 0; k0 < n; k0 += NB) {
        const int64_t kb = std::min(NB, n - k0);
        const int64_t me = x_upper ? std::min(m, k0 + kb) : m;        // rows that can be non-zero here
        for (int64_t j0 = k0; j0 < k0 + kb; j0 += SB) {
            const int jb = (int)std::min<int64_t>(SB, k0 + kb - j0);
            if (j0 > k0)
                GPX_TRY(gemm_nt(dtype, me, jb, j0 - k0, Xp(k0), ldx, Lp(j0, k0), ldl, Xp(j0), ldx, -1.0,
                                GPX_FULL, 0, 0, st));
            GPX_TRY(trsm_rows(dtype, Xp(j0), ldx, me, Lp(j0, j0), ldl, jb, st));
        }
        const int64_t r = k0 + kb;
        if (r < n)
            GPX_TRY(gemm_nt(dtype, me, n - r, kb, Xp(k0), ldx, Lp(r, k0), ldl, Xp(r), ldx, -1.0, GPX_FULL, 0, 0,
                            st));
    }
    return GPX_OK;
}

// X <- X L^-T for `count` systems in LOCK-STEP (operator route only: n a multiple of 512, every system's block operators
// complete in ops_base + i * sO): every launch covers all systems -- at n = 8192 a single system's far update is 1 - 2 rounds
// of tiles (32 x 32 at most), eight of them fill the chip.  L, X, the operators and the scratch block are sL / sX / sO / (m * 512)
// elements apart.  Same products, same order per system as trsm_right_lt.
int trsm_right_lt_batch(int dtype, const void *L, int64_t sL, int64_t n, int64_t ldl, void *X, int64_t sX, int64_t m, int64_t ldx,
                        hipStream_t st, int x_upper, const void *ops_base, int64_t sO, int count)
{
    if (n <= 0 || m <= 0 || count <= 0) return GPX_OK;
    const size_t es = esize(dtype);
    if (n % OB != 0 || !trsv_ops_ahead_ok(dtype, L, n, ldl)) { set_error("trsm_right_lt_batch: n must be a multiple of %d", OB); return GPX_ERR_ARG; }
    route_hit(RT_TRSM_OPS);
    const int64_t nfull = n / OB, BS = (int64_t)OB * OB, sS = m * OB;
    void *scr = nullptr;
    GPX_TRY(trsm_scratch((size_t)count * sS * es, &scr));
    for (int64_t k = 0; k < nfull; ++k) {
        const int64_t k0 = k * OB, r = k0 + OB;
        const int64_t me = x_upper ? std::min(m, r) : m;
        char *Xk = (char *)X + k0 * es;
        Batch b1; b1.count = count; b1.sA = sX; b1.sB = sO; b1.sC = sS;
        GPX_TRY(gemm_nt(dtype, me, OB, OB, Xk, ldx, (const char *)ops_base + (size_t)k * BS * es, OB, scr, OB, 1.0, GPX_FULL, 0, 0, st, 1, 0, &b1));
        for (int i = 0; i < count; ++i)
            GPX_HIP(hipMemcpy2DAsync(Xk + (size_t)i * sX * es, (size_t)ldx * es, (const char *)scr + (size_t)i * sS * es, (size_t)OB * es,
                                     (size_t)OB * es, (size_t)me, hipMemcpyDeviceToDevice, st));
        if (r < n) {
            Batch b2; b2.count = count; b2.sA = sX; b2.sB = sL; b2.sC = sX;
            GPX_TRY(gemm_nt(dtype, me, n - r, OB, Xk, ldx, (const char *)L + (r * ldl + k0) * es, ldl, (char *)X + r * es, ldx,
                            -1.0, GPX_FULL, 0, 0, st, 0, 0, &b2));
        }
    }
    return GPX_OK;
}

// ---- reductions (single workgroup, fixed order => deterministic) ----------
template <typename T, int MODE>   // MODE 0: sum a[i]*b[i]   1: 2*sum log a[i*stride]
__global__ __launch_bounds__(1024) void reduce_kernel(const T *__restrict__ a, const T *__restrict__ b,
                                                      int64_t n, int64_t stride, double *__restrict__ out,
                                                      int64_t sa, int64_t sb, int64_t so)
{
    // batched: workgroup blockIdx.x reduces vector pair blockIdx.x into out[blockIdx.x * so]
    a += (int64_t)blockIdx.x * sa;
    if (MODE == 0) b += (int64_t)blockIdx.x * sb;
    out += (int64_t)blockIdx.x * so;
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

// count > 1: matrix i at L + i * sL, result in out_dev[i * so]
int logdet_chol(int dtype, const void *L, int64_t n, int64_t ldl, double *out_dev, hipStream_t st, int count,
                int64_t sL, int64_t so)
{
    if (dtype == GPX_F64)
        hipLaunchKernelGGL((reduce_kernel<double, 1>), dim3(count), dim3(1024), 0, st, (const double *)L,
                           (const double *)nullptr, n, ldl + 1, out_dev, sL, (int64_t)0, so);
    else
        hipLaunchKernelGGL((reduce_kernel<float, 1>), dim3(count), dim3(1024), 0, st, (const float *)L,
                           (const float *)nullptr, n, ldl + 1, out_dev, sL, (int64_t)0, so);
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}

// count > 1: pair i is (a + i * sa, b + i * sb), result in out_dev[i * so]
int dot(int dtype, const void *a, const void *b, int64_t n, double *out_dev, hipStream_t st, int count, int64_t sa,
        int64_t sb, int64_t so)
{
    if (dtype == GPX_F64)
        hipLaunchKernelGGL((reduce_kernel<double, 0>), dim3(count), dim3(1024), 0, st, (const double *)a,
                           (const double *)b, n, 1, out_dev, sa, sb, so);
    else
        hipLaunchKernelGGL((reduce_kernel<float, 0>), dim3(count), dim3(1024), 0, st, (const float *)a,
                           (const float *)b, n, 1, out_dev, sa, sb, so);
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}

}  // namespace gpx

using namespace gpx;

extern "C" {

int gpx_d_trsv_lower(int dtype, const void *L, int64_t n, int64_t ldl, void *b, void *x,
                     int transpose, void *stream)
{
    gpx::StreamTurn turn__((hipStream_t)stream);     // (this thread's scratch buffers: one stream at a time, gpx_common.h)
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
    gpx::StreamTurn turn__((hipStream_t)stream);     // (this thread's scratch buffers: one stream at a time, gpx_common.h)
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
    gpx::StreamTurn turn__((hipStream_t)stream);     // (this thread's scratch buffers: one stream at a time, gpx_common.h)
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
