// gpx_leaf.h -- the register-resident 64 x 64 leaf shared by the factorisation (gpx_potrf.hip) and the
// batched inverses of the triangular solves (gpx_solve.hip).
#pragma once
#include "gpx_common.h"
#include <utility>
#include <type_traits>

namespace gpx {

constexpr int IB = 64;

// 1 / sqrt(p).  v_rsq_f64 is good to 2^-24.2 on gfx950 (tools/rsq_probe.hip); ONE third-order step
// y (1 + u/2 + 3u^2/8), u = 1 - p y^2, brings that to 2^-52.7 (1.2 ulp; two Newton steps: 2^-51.9) in a
// dependent chain of 5 instead of 7 operations -- this sits on the critical path of every pivot.
__device__ __forceinline__ double fast_rsqrt(double p)
{
    const double y = __builtin_amdgcn_rsq(p);
    const double u = fma(-(p * y), y, 1.0);
    return fma(fma(0.375, u, 0.5), u * y, y);
}
__device__ __forceinline__ float fast_rsqrt(float p)
{
    float y = __builtin_amdgcn_rsqf(p);
    y = y * fmaf(-0.5f * p * y, y, 1.5f);
    return y;
}

// The sweep itself: a[4][4] (this thread's tile of the block, lower part meaningful, identity padded beyond jb)
// is factored in place; with INV x[4][4] (the identity on entry) becomes the tile of X = L^-1.
// GIVEN: a already holds L (nothing is factored, info is not touched): only X = L^-1 is formed -- the
// batched 64 x 64 inverses of the triangular solves.
template <typename T, bool INV, bool GIVEN = false>
__device__ __forceinline__ void factor64(T (&a)[4][4], T (&x)[4][4], int jb, int64_t j0, int *__restrict__ info)
{
    __shared__ T sD[4][4];            // factored diagonal tile of the step (lower part)
    __shared__ T sR[4];               // its reciprocal pivots
    __shared__ T pan[IB][5];          // pitch 5: the column-tile reads pan[4 tc + c][k] of 16 lanes (stride 4 rows) spread
                                      // over the banks (pitch 4: 128-byte stride, 8-way conflicts)          // the step's 4 finished columns of L, rows below the diagonal tile
    __shared__ T xrow[4][IB];         // INV: the step's 4 finished rows of X
    const int tid = threadIdx.x;
    const int tr = tid >> 4, tc = tid & 15;
#pragma unroll 1
    for (int jt = 0; jt < IB / 4; ++jt) {
        // ---- A: the diagonal tile ----
        if (tr == jt && tc == jt) {
            T rk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const T piv = a[k][k];
                if (GIVEN) { rk[k] = (T)1 / piv; continue; }
                if (4 * jt + k < jb && !(piv > (T)0)) {              // also catches NaN
                    // first failure wins; an atomic because the leaves of one resident-panel launch run in
                    // workgroups on different XCDs (a plain store would sit in one L2)
                    atomicCAS(info, 0, (int)(j0 + 4 * jt + k + 1));
                }
                // 1/sqrt(piv) by v_rsq + Newton steps (error ~1 ulp), sqrt(piv) = piv * rinv
                const T rinv = fast_rsqrt(piv);
                rk[k] = rinv;
                a[k][k] = piv * rinv;
#pragma unroll
                for (int r = k + 1; r < 4; ++r) a[r][k] *= rinv;
#pragma unroll
                for (int c = k + 1; c < 4; ++c)
#pragma unroll
                    for (int r = c; r < 4; ++r) a[r][c] = fma(-a[r][k], a[c][k], a[r][c]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sR[r] = rk[r];
#pragma unroll
                for (int c = 0; c < 4; ++c) sD[r][c] = (c <= r) ? a[r][c] : (T)0;
            }
        }
        __syncthreads();
        // ---- B: tiles below the diagonal tile: P <- P L_dd^-T (forward over the 4 columns) ----
        if (tc == jt && tr > jt) {
            if (!GIVEN) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        T v = a[r][c];
#pragma unroll
                        for (int k = 0; k < c; ++k) v = fma(-a[r][k], sD[c][k], v);
                        a[r][c] = v * sR[c];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) pan[4 * tr + r][c] = a[r][c];
        }
        if (INV && tr == jt) {
            // rows 4jt.. of X: X_d <- L_dd^-1 X_d (forward over the 4 rows), final
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    T v = x[r][c];
#pragma unroll
                    for (int k = 0; k < r; ++k) v = fma(-sD[r][k], x[k][c], v);
                    x[r][c] = v * sR[r];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) xrow[r][4 * tc + c] = x[r][c];
        }
        __syncthreads();
        // ---- C: rank-4 update of everything below ----
        if (tr > jt) {
            T lr[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) lr[r][k] = pan[4 * tr + r][k];
            if (!GIVEN && tc > jt) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    T lc[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) lc[k] = pan[4 * tc + c][k];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int k = 0; k < 4; ++k) a[r][c] = fma(-lr[r][k], lc[k], a[r][c]);
                }
            }
            if (INV) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    T xs[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) xs[k] = xrow[k][4 * tc + c];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int k = 0; k < 4; ++k) x[r][c] = fma(-lr[r][k], xs[k], x[r][c]);
                }
            }
        }
        // (no barrier: the next step's A touches only its own registers and sD / sR, which
        //  nobody reads in C; pan / xrow are rewritten only after the next step's first barrier)
    }
}


// ---- the same leaf with a dedicated PIVOT WAVE (320 threads: waves 0-3 as above, wave 4 owns the 16 diagonal
// tiles, lane t the tile (t, t)).  The dependent chain of the leaf -- the four pivots of a diagonal tile, each a
// reciprocal square root plus its updates, ~150 cycles a pivot -- used to sit between the two barriers of every
// step with 255 threads waiting for one; here the pivot wave runs ONE STEP AHEAD of the others:
//   waves 0-3, step jt:  [barrier 1]  B: column jt's tiles against sD(jt) -> pan(jt), X rows -> xrow(jt)
//                        [barrier 2]  C: rank-4 update of their off-diagonal tiles (and of X) with pan(jt) / xrow(jt)
//   wave 4,    step jt:  [barrier 1]  --                                       [barrier 2]  rank-4 update of the diagonal
//                        tiles t > jt with pan(jt), then lane jt + 1 factors ITS tile -> sD(jt + 1), under the others' C.
// A step costs max(C, diagonal update + 4 pivots) + B instead of their sum.
// Who a thread is in factor64_pipe.  Non-pivot threads own tile (tr, tc) of the block: its strictly-lower part of
// A when tc < tr, its part of X when tc <= tr; pivot lane t < 16 owns the diagonal tile (t, t) of A.  Threads
// without a tile (tr = -1, tc = 99 / t >= 16) only keep the barriers.
struct LeafRole { int tr, tc; bool pivot; int t; };
// 320 threads: waves 0-3 the full 16 x 16 grid of tiles, wave 4 the pivots
__device__ __forceinline__ LeafRole leaf_role_320(int tid)
{
    LeafRole r; r.pivot = tid >= 256; r.t = tid - 256; r.tr = r.pivot ? -1 : (tid >> 4); r.tc = r.pivot ? 99 : (tid & 15);
    return r;
}
// 256 threads: waves 0-2 the 120 strictly-lower tiles (row by row) and the 16 diagonal tiles of X, wave 3 the pivots
__device__ __forceinline__ LeafRole leaf_role_256(int tid)
{
    LeafRole r; r.pivot = tid >= 192; r.t = tid - 192; r.tr = -1; r.tc = 99;
    if (tid < 120) {
        int tr = 1;
        while ((tr + 1) * tr / 2 <= tid) ++tr;
        r.tr = tr; r.tc = tid - tr * (tr - 1) / 2;
    } else if (tid < 136) {
        r.tr = r.tc = tid - 120;
    }
    return r;
}

template <typename T, bool INV>
__device__ __forceinline__ void factor64_pipe(T (&a)[4][4], T (&x)[4][4], int jb, int64_t j0, int *__restrict__ info,
                                              const LeafRole ro, int nsteps = IB / 4, unsigned long long *stamps = nullptr)
{
    // a: non-pivot threads: their OFF-diagonal tile (tr, tc), tc < tr (others unused); pivot lane t < 16: the
    // diagonal tile (t, t).  x: non-pivot threads: tile (tr, tc) of X, tc <= tr; unused by the pivot wave.
    __shared__ T sD[4][4];
    __shared__ T sR[4];
    __shared__ T pan[IB][5];          // pitch 5: the column-tile reads pan[4 tc + c][k] of 16 lanes (stride 4 rows) spread
                                      // over the banks (pitch 4: 128-byte stride, 8-way conflicts)
    __shared__ T xrow[4][IB];
    const int tid = threadIdx.x;
    const bool pivot = ro.pivot;
    const int tr = ro.tr, tc = ro.tc;
    const int t = ro.t;
    auto factor_tile = [&](int jt) {                  // by wave 4, lane jt: the four pivots of tile (jt, jt)
        T rk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const T piv = a[k][k];
            if (4 * jt + k < jb && !(piv > (T)0)) atomicCAS(info, 0, (int)(j0 + 4 * jt + k + 1));   // first failure wins
            const T rinv = fast_rsqrt(piv);
            rk[k] = rinv;
            a[k][k] = piv * rinv;
#pragma unroll
            for (int r = k + 1; r < 4; ++r) a[r][k] *= rinv;
#pragma unroll
            for (int c = k + 1; c < 4; ++c)
#pragma unroll
                for (int r = c; r < 4; ++r) a[r][c] = fma(-a[r][k], a[c][k], a[r][c]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sR[r] = rk[r];
#pragma unroll
            for (int c = 0; c < 4; ++c) sD[r][c] = (c <= r) ? a[r][c] : (T)0;
        }
    };
    if (pivot && t == 0) factor_tile(0);
#pragma unroll 1
    for (int jt = 0; jt < nsteps; ++jt) {             // nsteps < 16: timing diagnostics only (GPX_LEAF_ABLATE)
        __syncthreads();                              // barrier 1: sD(jt), sR(jt) are in
        if (stamps && (tid & 63) == 0) stamps[((tid >> 6) * 16 + jt) * 4 + 0] = __builtin_amdgcn_s_memtime();
        if (!pivot) {
            // ---- B ----
            if (tc == jt && tr > jt) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        T v = a[r][c];
#pragma unroll
                        for (int k = 0; k < c; ++k) v = fma(-a[r][k], sD[c][k], v);
                        a[r][c] = v * sR[c];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) pan[4 * tr + r][c] = a[r][c];
            }
            if (INV && tr == jt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        T v = x[r][c];
#pragma unroll
                        for (int k = 0; k < r; ++k) v = fma(-sD[r][k], x[k][c], v);
                        x[r][c] = v * sR[r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) xrow[r][4 * tc + c] = x[r][c];
            }
        }
        if (stamps && (tid & 63) == 0) stamps[((tid >> 6) * 16 + jt) * 4 + 1] = __builtin_amdgcn_s_memtime();
        __syncthreads();                              // barrier 2: pan(jt), xrow(jt) are in
        if (stamps && (tid & 63) == 0) stamps[((tid >> 6) * 16 + jt) * 4 + 2] = __builtin_amdgcn_s_memtime();
        if (pivot) {
            // diagonal tiles below the step: a_tt -= P_t P_t^T (lower part), then the next tile's pivots
            if (t > jt && t < IB / 4) {
                T lr[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int k = 0; k < 4; ++k) lr[r][k] = pan[4 * t + r][k];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = c; r < 4; ++r)
#pragma unroll
                        for (int k = 0; k < 4; ++k) a[r][c] = fma(-lr[r][k], lr[c][k], a[r][c]);
                if (t == jt + 1) factor_tile(jt + 1);
            }
        } else if (tr > jt) {
            // ---- C ----
            T lr[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) lr[r][k] = pan[4 * tr + r][k];
            if (tc > jt && tc < tr) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    T lc[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) lc[k] = pan[4 * tc + c][k];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int k = 0; k < 4; ++k) a[r][c] = fma(-lr[r][k], lc[k], a[r][c]);
                }
            }
            if (INV && tc <= jt) {                    // (rows 4jt.. of X are zero to the right of column tile jt)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    T xs[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) xs[k] = xrow[k][4 * tc + c];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int k = 0; k < 4; ++k) x[r][c] = fma(-lr[r][k], xs[k], x[r][c]);
                }
            }
        }
        if (stamps && (tid & 63) == 0) stamps[((tid >> 6) * 16 + jt) * 4 + 3] = __builtin_amdgcn_s_memtime();
    }
}

// ---- small MFMA products of the panel kernels: 16 x 16 output tiles, operands read as 32-byte runs of k ----
// (lane (li, lq) holds k = lq * SUB .. + SUB of a chunk of EPK values for row li of A and of B; both operands use the
//  same permutation of k, so the product is unchanged)
template <typename T> struct PM;
template <> struct PM<double> {
    typedef double v4 __attribute__((ext_vector_type(4)));
    static constexpr int EPK = 16, SUB = 4;
    __device__ static __forceinline__ v4 mfma(double a, double b, v4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct PM<float> {
    typedef float v4 __attribute__((ext_vector_type(4)));
    static constexpr int EPK = 32, SUB = 8;
    __device__ static __forceinline__ v4 mfma(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

template <typename T>
__device__ __forceinline__ void load_frag32(const T *__restrict__ p, T (&f)[PM<T>::SUB])
{
    struct alignas(16) Q { unsigned w[4]; };
    const Q a = reinterpret_cast<const Q *>(p)[0], b = reinterpret_cast<const Q *>(p)[1];
    memcpy(&f[0], &a, 16);
    memcpy(&f[PM<T>::SUB / 2], &b, 16);
}


// ---- the 64 x 64 leaf on the MFMA pipe (fp64) ------------------------------------------------------------------
// The sweeps above hold the block as 4 x 4 thread tiles and do every rank-4 update on the VALU: a 4-column step costs
// ~2700 cycles, almost all of it instruction issue and LDS round trips (tools/lat_probe.hip: the dependent arithmetic
// of the four pivots is ~250).  v_mfma_f64_16x16x4 has exactly the step's depth K = 4, so here the block STAYS in
// MFMA accumulator layout -- wave w rows 16w.., tiles a[jj] = columns 16jj.. (lane (li, lq): column li, rows lq + 4r)
// -- which is also how the resident panel kernel already holds it, and a step is
//   S1  the step's column strip (64 x 4) goes to LDS;                                            [barrier]
//   S2  ONE lane factors the 4 x 4 diagonal tile and inverts it (the only VALU chain left);      [barrier]
//   S3  every wave: new strip = strip . inv(L_dd)^T, one MFMA (A = strip rows, B = the 4 x 4 inverse, zero padded);
//   S4  the new strip goes to LDS (the diagonal rows take L_dd itself); the wave that owns the diagonal rows forms the
//       step's four rows of X = inv(L): X_d <- inv(L_dd) X_d, four MFMAs whose B operand is ALREADY in place in its
//       accumulator registers (lane (li, lq), register q holds X[c0 + lq][col li]) and whose result lands back in the
//       same registers; X_d goes to LDS;                                                         [barrier]
//   S5  every wave: the strip's final values into its tile, then rank-4 updates  a[jj] -= L_s L_s^T  (tiles right of
//       the step) and  x[jj] -= L_s X_d  (tiles left of it): 5 MFMAs.
// Masks instead of branches: operand rows / columns at or above the step contribute zeros.  Three barriers and ~6
// MFMAs a step.  Measured (tools/panel_stamps.py, stamps of step 9): 2200 cycles a step -- the pivot lane 1220 (one lane,
// ~60 dependent fp64 operations, 8 LDS reads, 12 writes), strip solve + LDS 460, updates 510, barriers -- = 14.7 us a
// leaf against 21.5 us for the pivot-wave VALU sweep (factor64_pipe), and no layout conversion on the way in or out.  Only the lower triangle of the block is read (the upper part of a diagonal block is stale in a
// lower-only factorisation) and only it is meaningful on return; x must hold the identity on entry.
// T = float (round 3): the same sweep on v_mfma_f32_16x16x4_f32.  Its accumulator layout differs -- lane (li, lq)
// holds rows 4 lq + r of column li, not rows lq + 4 r -- which only matters where the fp64 form uses accumulator
// registers as an MFMA operand "in place": the step's four rows of X sit in ONE lane group there, so they go through
// LDS once (sXraw) to become the B operand and come back from sXd; both hops are inside the owning wave and one step
// behind the pivot, off the critical path.  Everything else is layout-blind (operands come from LDS by row / column
// index, results go back into the tile they came from) and is written with PM<T>::row().
__device__ __forceinline__ void factor64_mfma(PM<double>::v4 (&a)[4], PM<double>::v4 (&x)[4], int64_t j0, int *__restrict__ info,
                                              int wave, int lane, unsigned long long *stamps = nullptr)
{
    typedef PM<double> M;
    typedef M::v4 v4;
    __shared__ __attribute__((aligned(16))) double sS[IB][4];   // the step's column strip as it is (S1) -- rows of the block
    __shared__ double sLn[IB][4];         // the step's finished strip of L
    __shared__ __attribute__((aligned(16))) double sWi[2][4][4];   // inv(L_dd) of the step (by parity), zero above the diagonal
    __shared__ __attribute__((aligned(16))) double sDd[4][4];      // L_dd, zero above the diagonal
    __shared__ double sXd[4][IB];         // four rows of X (of the step before: the X side runs one step behind)
    const int li = lane & 15, lq = lane >> 4;
    const v4 zero = {0.0, 0.0, 0.0, 0.0};
    if (threadIdx.x < 16) {
        sWi[0][threadIdx.x >> 2][threadIdx.x & 3] = 0.0; sWi[1][threadIdx.x >> 2][threadIdx.x & 3] = 0.0;
        sDd[threadIdx.x >> 2][threadIdx.x & 3] = 0.0;
    }
    // (the parts above the diagonal stay zero: the pivot lane only ever writes the lower ones; the first barrier of
    //  step 0 orders this against the first reads)
    double al_prev = 0.0;                 // the previous step's strip as (negated, masked) A operand: its X update is applied one step late
    // The X side (the step's four rows X_d <- inv(L_dd) X_d by the wave that owns them, then x -= L_s X_d) is not on the
    // path to the next pivot.  It runs ONE STEP BEHIND: X_d of step jt - 1 is formed while the pivot lane -- in another
    // wave -- factors the diagonal tile of step jt, and its rank-4 update joins the accumulator updates of step jt.
    // fully unrolled: the tile (jj0) and register (q) the step touches must be compile-time constants, or the
    // accumulator arrays go to scratch memory (first version: 544 bytes of scratch a lane, 48 us a leaf)
#pragma unroll
    for (int jt = 0; jt < IB / 4; ++jt) {
        const int c0 = 4 * jt, jj0 = jt >> 2, q = jt & 3;
        const int jjp = (jt - 1) >> 2, qp = (jt - 1) & 3;        // the step before (jt > 0)
        const int pw = (jj0 + 1) & 3;                              // the pivot lane's wave: never the wave busy with X_d
        // ---- S1 ----
        if ((li >> 2) == q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sS[16 * wave + lq + 4 * r][li & 3] = a[jj0][r];
        }
        __syncthreads();
        if (stamps && jt == 9 && threadIdx.x == 0) stamps[0] = __builtin_amdgcn_s_memtime();
        if (stamps && threadIdx.x == 0) { stamps[16 + jt] = __builtin_amdgcn_s_memrealtime(); stamps[32 + jt] = __builtin_amdgcn_s_memtime(); }
        // ---- S2 ----
        if (wave == pw && lane == 0) {
            double d[4][4], wi[4][4], rk[4];
            int bad = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double row[4];
                load_frag32<double>(&sS[c0 + i][0], row);       // two 16-byte LDS reads a row
#pragma unroll
                for (int k = 0; k < 4; ++k) d[i][k] = (k <= i) ? row[k] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double piv = d[k][k];
                if (bad == 0 && !(piv > 0.0)) bad = c0 + k + 1;                      // also catches NaN; reported after the chain
                const double rinv = fast_rsqrt(piv);
                rk[k] = rinv;
                d[k][k] = piv * rinv;
#pragma unroll
                for (int r = k + 1; r < 4; ++r) d[r][k] *= rinv;
#pragma unroll
                for (int c = k + 1; c < 4; ++c)
#pragma unroll
                    for (int r = c; r < 4; ++r) d[r][c] = fma(-d[r][k], d[c][k], d[r][c]);
            }
            // inv(L_dd): column by column, wi[i][k] = -rk[i] * sum_{m = k}^{i - 1} l[i][m] wi[m][k]
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < k) { wi[i][k] = 0.0; continue; }
                    if (i == k) { wi[i][k] = rk[i]; continue; }
                    double s = 0.0;
#pragma unroll
                    for (int m = k; m < i; ++m) s = fma(d[i][m], wi[m][k], s);
                    wi[i][k] = -rk[i] * s;
                }
            }
            // lower parts only, in 16-byte pieces where a row has a pair
            struct alignas(16) D2 { double v[2]; };
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int k = 0; k + 1 <= i; k += 2) {
                    *reinterpret_cast<D2 *>(&sWi[jt & 1][i][k]) = D2{{wi[i][k], wi[i][k + 1]}};
                    *reinterpret_cast<D2 *>(&sDd[i][k]) = D2{{d[i][k], (k + 1 <= i) ? d[i][k + 1] : 0.0}};
                }
                if ((i & 1) == 0) { sWi[jt & 1][i][i] = wi[i][i]; sDd[i][i] = d[i][i]; }
            }
            if (bad != 0) atomicCAS(info, 0, (int)(j0 + bad));                         // first failure wins
        }
        if (jt > 0 && wave == jjp) {
            // X_d of the step before (its inverse tile is still in the other half of sWi): B operand in place in the registers
            const double wprev = (li < 4) ? sWi[(jt - 1) & 1][li][lq] : 0.0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (jj > jjp) continue;                         // X is lower triangular: nothing to the right of the step
                const v4 u = M::mfma(wprev, x[jj][qp], zero);
                x[jj][qp] = u[0];
                sXd[lq][16 * jj + li] = u[0];
            }
        }
        if (stamps && jt == 9 && wave == pw && lane == 0) stamps[1] = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (stamps && jt == 9 && threadIdx.x == 0) stamps[2] = __builtin_amdgcn_s_memtime();
        // ---- S3: new strip = strip . inv(L_dd)^T ----
        const double winv = (li < 4) ? sWi[jt & 1][li][lq] : 0.0;   // B operand: B[k][n] = Winv[n][k]
        const v4 t = M::mfma(sS[16 * wave + li][lq], winv, zero);
        // ---- S4 ----
        if (li < 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * wave + lq + 4 * r;
                sLn[row][li] = (row >= c0 && row < c0 + 4) ? sDd[row - c0][li] : t[r];
            }
        }
        if (stamps && jt == 9 && threadIdx.x == 0) stamps[3] = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (stamps && jt == 9 && threadIdx.x == 0) stamps[4] = __builtin_amdgcn_s_memtime();
        // ---- S5 ----
        if ((li >> 2) == q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a[jj0][r] = sLn[16 * wave + lq + 4 * r][li & 3];
        }
        const int arow = 16 * wave + li;
        const double al = (arow > c0 + 3) ? -sLn[arow][lq] : 0.0;   // rows at or above the step: no update
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            // the tile that holds the NEXT step's strip first
            const int tj = (jj + ((jt + 1) >> 2)) & 3;
            if (tj >= jj0) {
                const int bcol = 16 * tj + li;
                const double bl = (bcol > c0 + 3) ? sLn[bcol][lq] : 0.0;          // columns up to the step are final
                a[tj] = M::mfma(al, bl, a[tj]);
            }
        }
        if (jt > 0) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                if (jj <= jjp) x[jj] = M::mfma(al_prev, sXd[lq][16 * jj + li], x[jj]);
        }
        al_prev = al;
        if (stamps && jt == 9 && threadIdx.x == 0) stamps[5] = __builtin_amdgcn_s_memtime();
        if (stamps && jt == 10 && threadIdx.x == 0) stamps[6] = __builtin_amdgcn_s_memtime();
        // (no barrier: the next S1 writes sS, last read in S3 -- before the barrier after S4; sLn / sXd / sDd and the
        //  parity half of sWi are rewritten only after the next step's first barrier, when every wave is past S5)
    }
    // the last step's X rows (its rank-4 update has no rows below it)
    if (wave == 3) {
        const double wprev = (li < 4) ? sWi[1][li][lq] : 0.0;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const v4 u = M::mfma(wprev, x[jj][3], zero);
            x[jj][3] = u[0];
        }
    }
}



// ---- round 4: the leaf in ONE wave, without LDS and without barriers -----------------------------------------------------
// Two variants of factor64_mfma were built first in round 4 and removed again (DESIGN section 3.2b keeps the measurements):
// the pivot lane running AHEAD of the strip -- it forms D(jt + 1) = N - (S Wi^T)(S Wi^T)^T itself, 80 fmas, and factors it
// while the waves finish step jt, two barriers a step instead of three -- fully unrolled (a 92 KB kernel) and as a rolled
// 16-step loop (61 KB).  Per-step stamps (profiles/r04_leaf_steps_v1_v2_v3.log) showed why neither paid: a step takes
// ~1.0 us while the diagonal workgroup has its CU to itself and 3.3 - 4.7 us while workgroups of the trailing update
// share it, whatever the schedule inside the step.
// This one touches LDS only to load the block and to store the results, and meets no barrier in between.  The idea:
// factor the TRANSPOSE.  With
// M = U^T U (U upper, L = U^T) held as accumulator tiles T(ti, tj), ti <= tj -- lane (li, lq), register r:
// M[16 ti + lq + 4 r][16 tj + li] -- the four ROWS c0 .. c0 + 3 of a step are register q = (c0 / 4) % 4 of the tiles
// of tile row jj0, and such a register is, untouched,
//     as an MFMA B operand (4 x 16):  B[k][n] = U[c0 + k][16 tj + n]
//     as an MFMA A operand (16 x 4):  A[i][k] = U[c0 + k][16 ti + i]
// so the rank-4 update  T(ti, tj) -= U_strip(ti)^T U_strip(tj)  takes its operands straight from the accumulator
// registers of the strip: no layout change, no LDS, no other wave.  The strip solve is one MFMA per tile whose A operand
// is the 4 x 4 inverse (every lane computes it, from the diagonal tile gathered with v_readlane) and whose result lands
// in the register it came from.  The same row operations applied to the identity give Y = U^-T = L^-1 = W, the inverse
// the panel kernel publishes.  One wave does all of it (14 MFMAs a step on average, its SIMD's matrix pipe 40 % busy);
// the other three waves of the workgroup wait at the barrier behind it.  16 steps, fully unrolled (static tile indices).
__device__ __forceinline__ constexpr int w1_tix(int ti, int tj) { return ti * 4 - ti * (ti - 1) / 2 + (tj - ti); }   // ti <= tj
__device__ __forceinline__ double w1_bcast(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

template <typename F, int... Is>
__device__ __forceinline__ void w1_steps(F &f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }

// The leaf issues its MFMAs from VOLATILE asm: volatile asm statements keep their source order in the instruction stream,
// which is the whole point (see "Schedule" below) -- hipcc's schedulers otherwise gather a step's MFMAs into one run and the
// step's pivot arithmetic into another (sched_barrier fences do not hold pure VALU code in place, sched_group_barrier
// pipelines were not followed; both tried).  What hipcc's hazard recogniser would do for MFMAs it knows about is settled by
// measurement instead (tools/mfma_hazard_probe_gen.py: full-mantissa operands, every result compared bit for bit with the same
// sequence given 40 wait states, one wave alone and 3 x 16384 waves contending for the matrix pipes; profiles/r04_mfma_hazard_probe.txt).
// v_mfma_f64_16x16x4_f64 on gfx950:
//   * registers 0 - 2 of the result (rows 0 .. 11 of the tile) are interlocked for every reader but LDS: VALU, v_readlane,
//     v_accvgpr_read, a following MFMA's srcA / srcB / srcC -- right with ZERO wait states;
//   * register 3 (rows 12 .. 15) is NOT: a reader inside 17 wait states gets a value of about single precision (the double
//     product is built up over the passes); 18 wait states, or one more MFMA issued in between (it waits for the pipe), are enough;
//   * an LDS store of a result needs 4 wait states for register 0 and 17 for register 3;
//   * an MFMA whose source register (VGPR or AGPR, srcA or srcB) a VALU instruction wrote 0 - 1 wait states earlier reads the
//     old value (2 and more: right) -- hipcc puts such copies in front of an asm operand as it likes, so every asm MFMA
//     starts with an s_nop 3 (4 wait states: twice the measured need);
//   * a VALU write into the destination of an MFMA in flight, or over its sources, is held back correctly (but costs the
//     writer the MFMA's whole duration: the strip MFMAs' destinations stay allocated for that reason);
//   * back-to-back accumulation into one tile is right.
// The leaf uses register 0 of the strip MFMAs only, reads other registers of a tile only at the places marked "register 3"
// below, and keeps every read of a tile behind an anchor that follows its last writer by an MFMA or by 18 wait states.
// The wait states below are MEASURED properties of gfx950's matrix pipe, not architectural guarantees: this file must not be
// compiled for any other target (the Makefile's ARCH is overridable), and the library checks the leaf against the
// compiler-scheduled one on the device it actually runs on before it uses it (leaf_selfcheck, gpx_panel.hip).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "gpx_leaf.h: the asm-scheduled MFMA leaf carries wait states measured on gfx950 only"
#endif
// Margins (round 5): the probe found the thresholds moving under contention (an LDS store of register 3 passes with 16 wait
// states alone and fails contended), so every anchor carries the measured minimum plus a third: register-3 readers 24 wait
// states (measured 18), LDS stores of register 0 at least 8 (measured 4).  ~6 cycles per anchor, ~0.1 us a leaf.
typedef double w1_v4 __attribute__((ext_vector_type(4)));
// ACC_A: the tiles live in AGPRs (the one-workgroup-per-CU instantiation, 512 registers a lane) or in VGPRs (the two-per-CU
// instantiation: with 256 registers a lane, AGPRs set aside for the leaf would be taken from the whole kernel)
template <bool ACC_A>
__device__ __forceinline__ void w1_mfma_acc(w1_v4 &c, double a, double b)             // c += a (16 x 4) . b (4 x 16)
{
    if constexpr (ACC_A) asm volatile("s_nop 3\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("s_nop 3\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <bool ACC_A>
__device__ __forceinline__ w1_v4 w1_mfma_strip(double a, double b)                     // a . b (rows 0 .. 3 = register 0 are what is wanted), b a register of a tile
{
    w1_v4 d;
    if constexpr (ACC_A) asm volatile("s_nop 3\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b));
    else asm volatile("s_nop 3\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
    return d;
}
// anchors on one to four tiles; NOPS: with 24 wait states in front (a register-3 read follows: 18 measured, see above)
template <bool ACC_A, bool NOPS>
__device__ __forceinline__ void w1_anchor(w1_v4 &t0)
{
    if constexpr (ACC_A && NOPS) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(t0));
    if constexpr (ACC_A && !NOPS) asm volatile("" : "+a"(t0));
    if constexpr (!ACC_A && NOPS) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(t0));
    if constexpr (!ACC_A && !NOPS) asm volatile("" : "+v"(t0));
}
template <bool ACC_A, bool NOPS>
__device__ __forceinline__ void w1_anchor(w1_v4 &t0, w1_v4 &t1)
{
    if constexpr (ACC_A && NOPS) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(t0), "+a"(t1));
    if constexpr (ACC_A && !NOPS) asm volatile("" : "+a"(t0), "+a"(t1));
    if constexpr (!ACC_A && NOPS) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(t0), "+v"(t1));
    if constexpr (!ACC_A && !NOPS) asm volatile("" : "+v"(t0), "+v"(t1));
}
template <bool ACC_A, bool NOPS>
__device__ __forceinline__ void w1_anchor(w1_v4 &t0, w1_v4 &t1, w1_v4 &t2)
{
    if constexpr (ACC_A && NOPS) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(t0), "+a"(t1), "+a"(t2));
    if constexpr (ACC_A && !NOPS) asm volatile("" : "+a"(t0), "+a"(t1), "+a"(t2));
    if constexpr (!ACC_A && NOPS) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(t0), "+v"(t1), "+v"(t2));
    if constexpr (!ACC_A && !NOPS) asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2));
}
template <bool ACC_A, bool NOPS>
__device__ __forceinline__ void w1_anchor(w1_v4 &t0, w1_v4 &t1, w1_v4 &t2, w1_v4 &t3)
{
    if constexpr (ACC_A && NOPS) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(t0), "+a"(t1), "+a"(t2), "+a"(t3));
    if constexpr (ACC_A && !NOPS) asm volatile("" : "+a"(t0), "+a"(t1), "+a"(t2), "+a"(t3));
    if constexpr (!ACC_A && NOPS) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
    if constexpr (!ACC_A && !NOPS) asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
}

// ---- the leaf itself: TWO waves of the diagonal workgroup ----
// sM: the block, row-major, lower triangle valid (pitch PT).  On return sM holds L (lower; entries above the diagonal are
// not written) and sW holds W = inv(L) (all 64 x 64, zeros above the diagonal).  Waves 0 and 1 of the workgroup call this
// (role = wave); a workgroup barrier must follow before anybody reads sM / sW.
//
// Schedule (second version, round 4).  The first version issued a step's MFMAs in one run behind its pivots -- v_mfma_f64_16x16x4
// holds the matrix pipe for 64 cycles on gfx950 and a wave issues in order, so a step cost (its 14 MFMAs) + (the pivot chain)
// ~ 2300 cycles with the two never overlapping.  Now
//   * the row operations on the identity (Y = inv) are NOT on the chain of the factorisation at all: wave 1 does them, on its
//     own SIMD's matrix pipe, one step or more behind wave 0, from the operands wave 0 leaves in LDS (per step and lane: the
//     strip solve's A operand and the rank-4 update's A operands, W1_SLOTS doubles; a step counter in LDS says how far wave 0
//     is -- LDS executes one wave's instructions in order, so the counter is simply written behind the data);
//   * wave 0's step is split into the CRITICAL pair -- the strip solve of the tile that holds the next diagonal 4 x 4 and that
//     tile's rank-4 update -- and everything else, DEFERRED: the deferred MFMAs of step jt are issued one at a time BETWEEN the
//     stages of step jt + 1's pivot arithmetic (gather | four pivots, each with its row of the inverse), which depends on the
//     critical pair alone: the VALU chain runs in the shadow of the matrix pipe.  Empty volatile asm statements carrying a
//     stage's values ("anchors") pin each stage between two MFMAs;
//   * the open part of the block is held NEGATED (S = -M, Z = -Y): the rank-4 update is then a plain accumulate whose A and
//     B operands are the strip registers as they stand; the strip solve uses -inv(l); the gather and the pivots work on -M;
//   * only the tile row of the strip needs a mask on its A operand (rows already final), one select;
//   * finished strips go to LDS at once from the registers the MFMA left them in (nothing is copied back into the tiles).
constexpr int W1_SLOTS = 5;                                        // per step and lane: aw, then the update operands of tile rows jn .. 3
constexpr int W1_BUF_DOUBLES = (IB / 4) * W1_SLOTS * 64;          // 40 KB (a slot per step; RING * W1_SLOTS * 64 with a ring)

__device__ __forceinline__ unsigned w1_lds_addr(const void *p)
{
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void *)p;
}

// RING: the number of steps the operand buffer holds (sBuf: RING * W1_SLOTS * 64 doubles).  IB / 4 = 16: every step has its own
// slot and wave 0 never waits for wave 1; 4 (10 KB, for the instantiation that runs two workgroups a CU): wave 0 looks at wave 1's
// counter (sCtl[1]: steps whose operands it has taken) before it reuses a slot.  sCtl[0] / sCtl[1] must be 0 on entry.
// TS: the type the block is STORED in (sM / sW): double, or float -- the fp32 panel kernel hands its diagonal blocks to this
// same fp64 leaf (the leaf is latency, not throughput: the chain of 64 dependent pivots costs the same in either precision,
// and the fp32 factorisation's diagonal blocks come out better than fp32 arithmetic would leave them).
template <int PT, int RING = IB / 4, int STAMP_STEP = -1, typename TS = double>
__device__ __forceinline__ void factor64_wave(TS (*sM)[PT], TS (*sW)[PT], double *sBuf, int *sCtl, int role, int64_t j0,
                                              int *__restrict__ info, int lane, unsigned long long *stamps = nullptr)
{
    typedef w1_v4 v4;
    static_assert(RING >= 2 && RING <= IB / 4, "ring of 2 .. 16 steps");
    constexpr bool ACC_A = RING == IB / 4;                      // (the instantiation with a slot per step is the one with 512 registers a lane)
    const int li = lane & 15, lq = lane >> 4;
    double (*buf)[W1_SLOTS][64] = reinterpret_cast<double (*)[W1_SLOTS][64]>(sBuf);
    unsigned step_addr = w1_lds_addr(sCtl);           // (not const: the step lambdas capture it)
    unsigned done_addr = w1_lds_addr(sCtl + 1);
    if (role == 1) {
        // ================= wave 1: Y = U^-T = L^-1 by the same row operations, from wave 0's operands =================
        v4 Y[10];                          // -Y tiles: Y(ti, tj'), ti >= tj', stored at w1_tix(tj', ti)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) Y[w1_tix(tj, ti)][r] = (ti == tj && lq + 4 * r == li) ? -1.0 : 0.0;
        auto ystep = [&](auto jtc) {
            constexpr int jt = decltype(jtc)::value;
            constexpr int c0 = 4 * jt, jj0 = jt >> 2, q = jt & 3;
            constexpr bool last = jt + 1 == IB / 4;
            constexpr int jn = last ? 4 : (jt + 1) >> 2;          // the first tile row still open after this step (none after the last)
            int seen;
            do {
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(step_addr) : "memory");
            } while (seen <= jt);
            double aw = buf[jt % RING][0][lane];
            double au[4];
#pragma unroll
            for (int ti = jn; ti < 4; ++ti) au[ti] = buf[jt % RING][1 + ti - jn][lane];
            if constexpr (RING < IB / 4)                          // the slot is free again once these loads have landed
                asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %1, %2" : "+v"(aw) : "v"(done_addr), "v"(jt + 1) : "memory");
            // (register 3: the tiles of row jj0 were last written by the previous step's updates, possibly by its very last
            //  MFMA; the poll and the loads above are two LDS round trips, the 18 wait states are on top for q == 3)
            if constexpr (q == 3) {
                if constexpr (jj0 == 0) w1_anchor<ACC_A, true>(Y[w1_tix(0, 0)]);
                if constexpr (jj0 == 1) w1_anchor<ACC_A, true>(Y[w1_tix(0, 1)], Y[w1_tix(1, 1)]);
                if constexpr (jj0 == 2) w1_anchor<ACC_A, true>(Y[w1_tix(0, 2)], Y[w1_tix(1, 2)], Y[w1_tix(2, 2)]);
                if constexpr (jj0 == 3) w1_anchor<ACC_A, true>(Y[w1_tix(0, 3)], Y[w1_tix(1, 3)], Y[w1_tix(2, 3)], Y[w1_tix(3, 3)]);
            }
            v4 ys[4];
#pragma unroll
            for (int tj = 0; tj <= jj0; ++tj) ys[tj] = w1_mfma_strip<ACC_A>(aw, Y[w1_tix(tj, jj0)][q]);
#pragma unroll
            for (int ti = jn; ti < 4; ++ti)
#pragma unroll
                for (int tj = 0; tj <= jj0; ++tj) w1_mfma_acc<ACC_A>(Y[w1_tix(tj, ti)], au[ti], ys[tj][0]);
            // (an LDS store of register 0 needs 4 wait states behind the MFMA; the anchor provides 8)
            if constexpr (jj0 == 0) asm volatile("s_nop 7" : "+v"(ys[0]));
            if constexpr (jj0 == 1) asm volatile("s_nop 7" : "+v"(ys[0]), "+v"(ys[1]));
            if constexpr (jj0 == 2) asm volatile("s_nop 7" : "+v"(ys[0]), "+v"(ys[1]), "+v"(ys[2]));
            if constexpr (jj0 == 3) asm volatile("s_nop 7" : "+v"(ys[0]), "+v"(ys[1]), "+v"(ys[2]), "+v"(ys[3]));
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                const int row = c0 + lq, col = 16 * tj + li;      // W = Y with zeros right of the diagonal
                sW[row][col] = (TS)((tj <= jj0 && col <= row) ? ys[tj <= jj0 ? tj : 0][0] : 0.0);
            }
        };
        w1_steps(ystep, std::make_integer_sequence<int, IB / 4>{});
        return;
    }
    // ================= wave 0: the factorisation =================
    unsigned long long tm[12] = {};                                // STAMP_STEP >= 0 (tools/leaf_probe.hip): issue times inside that step
    v4 T[10];                              // -M tiles (ti <= tj)
    if (stamps && lane == 0) { stamps[16] = __builtin_amdgcn_s_memrealtime(); stamps[32] = __builtin_amdgcn_s_memtime(); }
    // ---- load: T(ti, tj)[r] = -M[16 ti + lq + 4 r][16 tj + li], taken from the lower triangle (symmetric) ----
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = ti; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + lq + 4 * r, col = 16 * tj + li;
                T[w1_tix(ti, tj)][r] = -(double)(row <= col ? sM[col][row] : sM[row][col]);
            }
    int bad_at = 0;
    double d[4][4], wi[4][4], aw = 0.0, aw_next = 0.0;
    // the 4 x 4 diagonal tile of step jt to every lane, as it stands in the tile (negated): d[r][c] (r >= c) = -M[c0 + c][c0 + r]
    // (20 v_readlane; a negation here would be 10 SALU operations each waiting for a VALU-written SGPR)
    auto gather = [&](int jt) {
        const double src = T[w1_tix(jt >> 2, jt >> 2)][jt & 3];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) d[r][c] = w1_bcast(src, 16 * c + 4 * (jt & 3) + r);
    };
    // pivot k of the tile: on entry d[r][c], c >= k, hold the NEGATED open part; column k becomes l = R^T (final), the columns
    // right of it are updated (still negated: e += l l^T).  Then row k of inv(l) -- it needs nothing later than this pivot --
    // and its entries of the strip solve's A operand: A[i][m] = -inv(l)[i][m] at lane (li = i < 4, lq = m)
    auto pivot = [&](int jt, int k) {
        const double npiv = d[k][k];
        if (bad_at == 0 && !(npiv < 0.0)) bad_at = 4 * jt + k + 1;                   // also catches NaN; reported at the end
        const double rinv = fast_rsqrt(-npiv);
        d[k][k] = -npiv * rinv;
#pragma unroll
        for (int r = k + 1; r < 4; ++r) d[r][k] = -d[r][k] * rinv;
#pragma unroll
        for (int c = k + 1; c < 4; ++c)
#pragma unroll
            for (int r = c; r < 4; ++r) d[r][c] = fma(d[r][k], d[c][k], d[r][c]);
        wi[k][k] = rinv;
#pragma unroll
        for (int c = 0; c < k; ++c) {
            double sacc = 0.0;
#pragma unroll
            for (int m = c; m < k; ++m) sacc = fma(d[k][m], wi[m][c], sacc);
            wi[k][c] = -rinv * sacc;
        }
        if (k == 0) aw_next = 0.0;
#pragma unroll
        for (int c = 0; c <= k; ++c) aw_next = (li == k && lq == c) ? -wi[k][c] : aw_next;
    };
    // anchors: the values a stage hands to the next one pass through an empty volatile asm, which sits in the MFMAs' order
    auto anchor_tile = [&](v4 &t) { w1_anchor<ACC_A, false>(t); };
    auto anchor_d = [&]() {
        asm volatile("" : "+s"(d[0][0]), "+s"(d[1][0]), "+s"(d[2][0]), "+s"(d[3][0]), "+s"(d[1][1]), "+s"(d[2][1]), "+s"(d[3][1]),
                          "+s"(d[2][2]), "+s"(d[3][2]), "+s"(d[3][3]));
    };
    auto anchor_p = [&](int k) {
        if (k == 0) asm volatile("" : "+v"(d[1][0]), "+v"(d[2][0]), "+v"(d[3][0]), "+v"(d[1][1]), "+v"(d[2][1]), "+v"(d[3][1]),
                                      "+v"(d[2][2]), "+v"(d[3][2]), "+v"(d[3][3]), "+v"(wi[0][0]), "+v"(aw_next));
        if (k == 1) asm volatile("" : "+v"(d[2][1]), "+v"(d[3][1]), "+v"(d[2][2]), "+v"(d[3][2]), "+v"(d[3][3]), "+v"(wi[1][0]), "+v"(wi[1][1]), "+v"(aw_next));
        if (k == 2) asm volatile("" : "+v"(d[3][2]), "+v"(d[3][3]), "+v"(wi[2][0]), "+v"(wi[2][1]), "+v"(wi[2][2]), "+v"(aw_next));
        if (k == 3) asm volatile("" : "+v"(aw_next));
    };
    gather(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) pivot(0, k);
    aw = aw_next;
    auto step = [&](auto jtc) {                                   // (a step per instantiation: every tile index below is static)
        constexpr int jt = decltype(jtc)::value;
        constexpr int c0 = 4 * jt, jj0 = jt >> 2, q = jt & 3;
        constexpr bool last = jt + 1 == IB / 4;
        constexpr int jn = last ? 3 : (jt + 1) >> 2;              // the tile row of the next step = the first still open after this one
        constexpr int ct = (q == 3 && !last) ? jn : jj0;          // the tile of the critical strip
        // the step's finished strips: rows c0 .. c0 + 3 of U = register 0 of these (the strip MFMA's whole destination stays
        // allocated to the end of the step: hipcc would otherwise hand the three registers nobody reads to the next value it
        // computes while the MFMA is still to write them, and the hardware holds that instruction back until it has)
        v4 us[4];
        double am = 0.0;
        auto mark = [&](int i) { if constexpr (jt == STAMP_STEP) tm[i] = __builtin_amdgcn_s_memtime(); };
        // (register 3: the strips read register q of the tiles of row jj0; their last writers are the previous step's critical
        //  update and the first of its deferred updates, each followed by another MFMA or by the whole pivot chain -- the anchor
        //  keeps hipcc's copies of those registers from moving up behind the writer)
        if constexpr (jj0 == 0) w1_anchor<ACC_A, false>(T[w1_tix(0, 0)], T[w1_tix(0, 1)], T[w1_tix(0, 2)], T[w1_tix(0, 3)]);
        if constexpr (jj0 == 1) w1_anchor<ACC_A, false>(T[w1_tix(1, 1)], T[w1_tix(1, 2)], T[w1_tix(1, 3)]);
        if constexpr (jj0 == 2) w1_anchor<ACC_A, false>(T[w1_tix(2, 2)], T[w1_tix(2, 3)]);
        if constexpr (jj0 == 3) w1_anchor<ACC_A, false>(T[w1_tix(3, 3)]);
        // ---- critical pair ----
        mark(0);
        us[ct] = w1_mfma_strip<ACC_A>(aw, T[w1_tix(jj0, ct)][q]);
        if constexpr (q < 3) am = (li > 4 * q + 3) ? us[jj0][0] : 0.0;   // A operand for the strip's own tile row: rows up to the strip are final
        if constexpr (!last) w1_mfma_acc<ACC_A>(T[w1_tix(jn, jn)], jn == jj0 ? am : us[jn][0], us[jn][0]);
        // ---- deferred: ordinals [from, to) of: the other strips, the updates (row-major: the next step's tile row first) ----
        auto defer = [&](int from, int to) {
            int k = 0;
#pragma unroll
            for (int tj = jj0; tj < 4; ++tj) {
                if (tj == ct) continue;
                if (k >= from && k < to) us[tj] = w1_mfma_strip<ACC_A>(aw, T[w1_tix(jj0, tj)][q]);
                ++k;
            }
            if constexpr (!last) {
#pragma unroll
                for (int ti = jn; ti < 4; ++ti)
#pragma unroll
                    for (int tj = ti; tj < 4; ++tj) {
                        if (ti == jn && tj == jn) continue;
                        if (k >= from && k < to) w1_mfma_acc<ACC_A>(T[w1_tix(ti, tj)], ti == jj0 ? am : us[ti][0], us[tj][0]);
                        ++k;
                    }
            }
        };
        constexpr int defer_count = (3 - jj0) + (last ? 0 : (4 - jn) * (5 - jn) / 2 - 1);     // strips besides the critical one + updates besides it
        // wave 1's operands of this step: the strips are complete (issued at least three stages ago), 8 wait states for the stores
        // (register 0: 4 measured)
        auto publish = [&](unsigned flag_addr) {
            if constexpr (jj0 == 0) asm volatile("s_nop 7" : "+v"(us[0]), "+v"(us[1]), "+v"(us[2]), "+v"(us[3]));
            if constexpr (jj0 == 1) asm volatile("s_nop 7" : "+v"(us[1]), "+v"(us[2]), "+v"(us[3]));
            if constexpr (jj0 == 2) asm volatile("s_nop 7" : "+v"(us[2]), "+v"(us[3]));
            if constexpr (jj0 == 3) asm volatile("s_nop 7" : "+v"(us[3]));
            if constexpr (RING < IB / 4 && jt >= RING) {          // wave 1 has taken the operands of step jt - RING?
                int taken;
                do {
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(taken) : "v"(done_addr) : "memory");
                } while (taken < jt - RING + 1);
            }
            buf[jt % RING][0][lane] = aw;
            if constexpr (!last) {
#pragma unroll
                for (int ti = jn; ti < 4; ++ti) buf[jt % RING][1 + ti - jn][lane] = (ti == jj0) ? am : us[ti][0];
            }
            asm volatile("ds_write_b32 %0, %1" : : "v"(flag_addr), "v"(jt + 1) : "memory");
        };
        if constexpr (!last) {
            mark(1);
            defer(0, 1);
            mark(2);
            // (register 3: the gather reads register (jt + 1) & 3 of the tile the critical update has just written; one deferred
            //  MFMA in between is enough, the last steps have none)
            if constexpr (((jt + 1) & 3) == 3 && defer_count == 0) w1_anchor<ACC_A, true>(T[w1_tix(jn, jn)]);
            else anchor_tile(T[w1_tix(jn, jn)]);
            gather(jt + 1);
            anchor_d();
            mark(3);
            defer(1, 2);
            pivot(jt + 1, 0); anchor_p(0);
            mark(4);
            defer(2, 3);
            pivot(jt + 1, 1); anchor_p(1);
            mark(5);
            defer(3, 4);
            pivot(jt + 1, 2); anchor_p(2);
            mark(6);
            defer(4, 5);
            pivot(jt + 1, 3); anchor_p(3);
            mark(7);
            publish(step_addr);
            defer(5, 64);
            mark(8);
        } else {
            defer(0, 64);
            publish(step_addr);
        }
        // ---- the step's finished rows to LDS: L = U^T (lower part only) ----
        if constexpr (jj0 == 0) asm volatile("s_nop 7" : "+v"(us[0]), "+v"(us[1]), "+v"(us[2]), "+v"(us[3]));
        if constexpr (jj0 == 1) asm volatile("s_nop 7" : "+v"(us[1]), "+v"(us[2]), "+v"(us[3]));
        if constexpr (jj0 == 2) asm volatile("s_nop 7" : "+v"(us[2]), "+v"(us[3]));
        if constexpr (jj0 == 3) asm volatile("s_nop 7" : "+v"(us[3]));
#pragma unroll
        for (int tj = jj0; tj < 4; ++tj) {
            const int row = 16 * tj + li, col = c0 + lq;          // U[col][row] = L[row][col]
            if (col <= row) sM[row][col] = (TS)us[tj][0];
        }
        aw = aw_next;
        mark(9);
    };
    w1_steps(step, std::make_integer_sequence<int, IB / 4>{});
    if (stamps && lane == 0) { stamps[17] = __builtin_amdgcn_s_memrealtime(); stamps[33] = __builtin_amdgcn_s_memtime(); }
    if constexpr (STAMP_STEP >= 0) if (stamps && lane == 0) for (int i = 0; i < 10; ++i) stamps[40 + i] = tm[i];
    if (lane == 0 && bad_at != 0) atomicCAS(info, 0, (int)(j0 + bad_at));          // first failure wins
}


template <typename T>
__device__ __forceinline__ void factor64_mfma_t(typename PM<T>::v4 (&a)[4], typename PM<T>::v4 (&x)[4], int64_t j0,
                                              int *__restrict__ info, int wave, int lane, unsigned long long *stamps = nullptr)
{
    typedef PM<T> M;
    typedef typename M::v4 v4;
    constexpr bool F64 = sizeof(T) == 8;
    __shared__ __attribute__((aligned(16))) T sS[IB][4];        // the step's column strip as it is (S1) -- rows of the block
    __shared__ T sLn[IB][4];              // the step's finished strip of L
    __shared__ __attribute__((aligned(16))) T sWi[2][4][4];     // inv(L_dd) of the step (by parity), zero above the diagonal
    __shared__ __attribute__((aligned(16))) T sDd[4][4];        // L_dd, zero above the diagonal
    __shared__ T sXd[4][IB];              // four rows of X (of the step before: the X side runs one step behind)
    __shared__ T sXraw[4][IB];            // fp32 only: the same four rows before inv(L_dd) is applied
    const int li = lane & 15, lq = lane >> 4;
    const v4 zero = {(T)0, (T)0, (T)0, (T)0};
    if (threadIdx.x < 16) {
        sWi[0][threadIdx.x >> 2][threadIdx.x & 3] = (T)0; sWi[1][threadIdx.x >> 2][threadIdx.x & 3] = (T)0;
        sDd[threadIdx.x >> 2][threadIdx.x & 3] = (T)0;
    }
    // (the parts above the diagonal stay zero: the pivot lane only ever writes the lower ones; the first barrier of
    //  step 0 orders this against the first reads)
    T al_prev = (T)0;                     // the previous step's strip as (negated, masked) A operand: its X update is applied one step late
    // X_d <- inv(L_dd) X_d for the four rows c0p .. c0p + 3 of X (held by wave jjp), by the wave that owns them
    // (publish: the finished rows also go to sXd for the other waves' rank-4 update of X; not after the last step, when
    //  nobody needs them and the others may still be reading the rows of the step before)
    auto finish_x_rows = [&](int jjp, int qp, int parity, bool publish) {
        const T wprev = (li < 4) ? sWi[parity][li][lq] : (T)0;
        if constexpr (F64) {
            // fp64: register qp of lane (li, lq) IS X[c0p + lq][col li]: the B operand is in place, the result lands back
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (jj > jjp) continue;                         // X is lower triangular: nothing to the right of the step
                const v4 u = M::mfma(wprev, x[jj][qp], zero);
                x[jj][qp] = u[0];
                if (publish) sXd[lq][16 * jj + li] = u[0];
            }
        } else {
            // fp32: the four rows are registers 0..3 of lane group qp -- through LDS (sXraw, private to this wave) to
            // become the operand, and back the same way
            if (lq == qp) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sXraw[r][16 * jj + li] = x[jj][r];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: program order + the wait orders its own LDS traffic)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (jj > jjp) continue;
                const v4 u = M::mfma(wprev, sXraw[lq][16 * jj + li], zero);     // rows 0..3 of the result: lane group 0, registers 0..3
                if (lq == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sXraw[r][16 * jj + li] = u[r];
                        if (publish) sXd[r][16 * jj + li] = u[r];
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lq == qp) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    if (jj > jjp) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[jj][r] = sXraw[r][16 * jj + li];
                }
            }
        }
    };
    // The X side (the step's four rows X_d <- inv(L_dd) X_d by the wave that owns them, then x -= L_s X_d) is not on the
    // path to the next pivot.  It runs ONE STEP BEHIND: X_d of step jt - 1 is formed while the pivot lane -- in another
    // wave -- factors the diagonal tile of step jt, and its rank-4 update joins the accumulator updates of step jt.
    // fully unrolled: the tile (jj0) and register (q) the step touches must be compile-time constants, or the
    // accumulator arrays go to scratch memory (first version: 544 bytes of scratch a lane, 48 us a leaf)
#pragma unroll
    for (int jt = 0; jt < IB / 4; ++jt) {
        const int c0 = 4 * jt, jj0 = jt >> 2, q = jt & 3;
        const int jjp = (jt - 1) >> 2, qp = (jt - 1) & 3;        // the step before (jt > 0)
        const int pw = (jj0 + 1) & 3;                              // the pivot lane's wave: never the wave busy with X_d
        // ---- S1 ----
        if ((li >> 2) == q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sS[16 * wave + M::row(lane, r)][li & 3] = a[jj0][r];
        }
        __syncthreads();
        if (stamps && jt == 9 && threadIdx.x == 0) stamps[0] = __builtin_amdgcn_s_memtime();
        // ---- S2 ----
        if (wave == pw && lane == 0) {
            T d[4][4], wi[4][4], rk[4];
            int bad = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                T row[4];
                if constexpr (F64) load_frag32<T>(&sS[c0 + i][0], row);       // two 16-byte LDS reads a row
                else { struct alignas(16) Q4 { T v[4]; }; const Q4 qv = *reinterpret_cast<const Q4 *>(&sS[c0 + i][0]); memcpy(row, &qv, 16); }
#pragma unroll
                for (int k = 0; k < 4; ++k) d[i][k] = (k <= i) ? row[k] : (T)0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const T piv = d[k][k];
                if (bad == 0 && !(piv > (T)0)) bad = c0 + k + 1;                      // also catches NaN; reported after the chain
                const T rinv = fast_rsqrt(piv);
                rk[k] = rinv;
                d[k][k] = piv * rinv;
#pragma unroll
                for (int r = k + 1; r < 4; ++r) d[r][k] *= rinv;
#pragma unroll
                for (int c = k + 1; c < 4; ++c)
#pragma unroll
                    for (int r = c; r < 4; ++r) d[r][c] = fma(-d[r][k], d[c][k], d[r][c]);
            }
            // inv(L_dd): column by column, wi[i][k] = -rk[i] * sum_{m = k}^{i - 1} l[i][m] wi[m][k]
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < k) { wi[i][k] = (T)0; continue; }
                    if (i == k) { wi[i][k] = rk[i]; continue; }
                    T s = (T)0;
#pragma unroll
                    for (int m = k; m < i; ++m) s = fma(d[i][m], wi[m][k], s);
                    wi[i][k] = -rk[i] * s;
                }
            }
            // lower parts only, in 16-byte pieces where a row has a pair (fp64) / one 16-byte row (fp32: the zeros above the
            // diagonal are written too, they are zero anyway)
            if constexpr (F64) {
                struct alignas(16) D2 { T v[2]; };
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int k = 0; k + 1 <= i; k += 2) {
                        *reinterpret_cast<D2 *>(&sWi[jt & 1][i][k]) = D2{{wi[i][k], wi[i][k + 1]}};
                        *reinterpret_cast<D2 *>(&sDd[i][k]) = D2{{d[i][k], (k + 1 <= i) ? d[i][k + 1] : (T)0}};
                    }
                    if ((i & 1) == 0) { sWi[jt & 1][i][i] = wi[i][i]; sDd[i][i] = d[i][i]; }
                }
            } else {
                struct alignas(16) Q4 { T v[4]; };
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    Q4 qw, qd;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { qw.v[k] = (k <= i) ? wi[i][k] : (T)0; qd.v[k] = (k <= i) ? d[i][k] : (T)0; }
                    *reinterpret_cast<Q4 *>(&sWi[jt & 1][i][0]) = qw;
                    *reinterpret_cast<Q4 *>(&sDd[i][0]) = qd;
                }
            }
            if (bad != 0) atomicCAS(info, 0, (int)(j0 + bad));                         // first failure wins
        }
        if (jt > 0 && wave == jjp) finish_x_rows(jjp, qp, (jt - 1) & 1, true);       // (its inverse tile is still in the other half of sWi)
        if (stamps && jt == 9 && wave == pw && lane == 0) stamps[1] = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (stamps && jt == 9 && threadIdx.x == 0) stamps[2] = __builtin_amdgcn_s_memtime();
        // ---- S3: new strip = strip . inv(L_dd)^T ----
        const T winv = (li < 4) ? sWi[jt & 1][li][lq] : (T)0;   // B operand: B[k][n] = Winv[n][k]
        const v4 t = M::mfma(sS[16 * wave + li][lq], winv, zero);
        // ---- S4 ----
        if (li < 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * wave + M::row(lane, r);
                sLn[row][li] = (row >= c0 && row < c0 + 4) ? sDd[row - c0][li] : t[r];
            }
        }
        if (stamps && jt == 9 && threadIdx.x == 0) stamps[3] = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (stamps && jt == 9 && threadIdx.x == 0) stamps[4] = __builtin_amdgcn_s_memtime();
        // ---- S5 ----
        if ((li >> 2) == q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a[jj0][r] = sLn[16 * wave + M::row(lane, r)][li & 3];
        }
        const int arow = 16 * wave + li;
        const T al = (arow > c0 + 3) ? -sLn[arow][lq] : (T)0;   // rows at or above the step: no update
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            // the tile that holds the NEXT step's strip first
            const int tj = (jj + ((jt + 1) >> 2)) & 3;
            if (tj >= jj0) {
                const int bcol = 16 * tj + li;
                const T bl = (bcol > c0 + 3) ? sLn[bcol][lq] : (T)0;          // columns up to the step are final
                a[tj] = M::mfma(al, bl, a[tj]);
            }
        }
        if (jt > 0) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                if (jj <= jjp) x[jj] = M::mfma(al_prev, sXd[lq][16 * jj + li], x[jj]);
        }
        al_prev = al;
        if (stamps && jt == 9 && threadIdx.x == 0) stamps[5] = __builtin_amdgcn_s_memtime();
        if (stamps && jt == 10 && threadIdx.x == 0) stamps[6] = __builtin_amdgcn_s_memtime();
        // (no barrier: the next S1 writes sS, last read in S3 -- before the barrier after S4; sLn / sXd / sDd and the
        //  parity half of sWi are rewritten only after the next step's first barrier, when every wave is past S5)
    }
    // the last step's X rows (its rank-4 update has no rows below it)
    if (wave == 3) finish_x_rows(3, 3, 1, false);
}

// fp64: the round-2 function, untouched (its instruction stream is tuned: a refactoring of it into the generic form below
// cost 4 % of an n = 8192 factorisation); fp32: the generic form
template <typename T> struct LeafMfma;
template <> struct LeafMfma<double> {
    __device__ static __forceinline__ void run(PM<double>::v4 (&a)[4], PM<double>::v4 (&x)[4], int64_t j0, int *__restrict__ info,
                                               int wave, int lane, unsigned long long *stamps)
    {
        factor64_mfma(a, x, j0, info, wave, lane, stamps);
    }
};
template <> struct LeafMfma<float> {
    __device__ static __forceinline__ void run(PM<float>::v4 (&a)[4], PM<float>::v4 (&x)[4], int64_t j0, int *__restrict__ info,
                                               int wave, int lane, unsigned long long *stamps)
    {
        factor64_mfma_t<float>(a, x, j0, info, wave, lane, stamps);
    }
};

}  // namespace gpx
