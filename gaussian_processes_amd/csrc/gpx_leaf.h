// gpx_leaf.h -- the register-resident 64 x 64 leaf shared by the factorisation (gpx_potrf.hip) and the
// batched inverses of the triangular solves (gpx_solve.hip).
#pragma once
#include "gpx_common.h"

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



// ---- round 4: the same leaf with the pivot lane running AHEAD of the strip -------------------------------------------
// factor64_mfma above spends 1220 of its 2200 cycles a step with 255 lanes waiting at a barrier for ONE lane: that lane
// reads the step's 4 x 4 diagonal tile from LDS -- which exists only after the previous step's rank-4 update (S5) and
// the strip hand-over (S1) -- factors it, inverts it, and only then can S3 start.  But the NEXT diagonal tile depends on
// very little of step jt: with S = the raw strip rows c0+4 .. c0+7 (4 x 4), N = the raw next tile (4 x 4, as updated
// through step jt - 1) and Wi = inv(L_dd) of step jt,
//         D(jt + 1) = N - (S Wi^T)(S Wi^T)^T                                        80 fused multiply-adds,
// all of whose inputs are in LDS at the START of step jt.  So the pivot lane forms D(jt + 1) itself, while the waves
// do S3 / S4 of step jt, and factors + inverts it while they do S5 of step jt: inverse(jt + 1) is published before
// step jt + 1 begins and nobody ever waits for the pivot chain -- a step costs max(pivot lane's loop, the waves'
// S3 + S4 + S5) with TWO barriers instead of the SUM with three.
//   [B1]  waves: S3 strip solve (1 MFMA), S4 strip -> LDS, X rows of the step before   |  pivot: S, N from LDS, D(jt+1)
//   [B2]  waves: S5 rank-4 updates; raw strip jt+1 and raw tile jt+2 -> LDS            |  pivot: factor + invert D(jt+1), publish
// The pivot lane's rows of the strip are computed twice (its own fmas for D(jt + 1); the S3 MFMA for the strip that is
// stored and used by every update): two roundings of the same exact numbers, both backward stable; the factor L_dd the
// pivot lane publishes is the one that is stored.  The pivot lane lives in wave 0, whose rows are final after four
// steps.  Rank-4 updates of tiles ABOVE the diagonal (column tile > the wave's row tile) are skipped: nothing reads
// them (only the lower triangle of the block is loaded, used and stored).
__device__ __forceinline__ void factor64_mfma2(PM<double>::v4 (&a)[4], PM<double>::v4 (&x)[4], int64_t j0, int *__restrict__ info,
                                               int wave, int lane, unsigned long long *stamps = nullptr)
{
    typedef PM<double> M;
    typedef M::v4 v4;
    __shared__ __attribute__((aligned(16))) double sS[IB][4];   // the step's raw column strip (rows of the block)
    __shared__ __attribute__((aligned(16))) double sN[4][4];    // the NEXT step's raw diagonal tile
    __shared__ double sLn[IB][4];         // the step's finished strip of L
    __shared__ __attribute__((aligned(16))) double sWi[2][4][4];   // inv(L_dd) of the step (by parity), zero above the diagonal
    __shared__ __attribute__((aligned(16))) double sDd[2][4][4];   // L_dd (by parity), zero above the diagonal
    __shared__ double sXd[4][IB];         // four rows of X (of the step before: the X side runs one step behind)
    const int li = lane & 15, lq = lane >> 4;
    const v4 zero = {0.0, 0.0, 0.0, 0.0};
    const bool pivot = wave == 0 && lane == 0;
    if (threadIdx.x < 32) {
        const int h = threadIdx.x >> 4, e = threadIdx.x & 15;
        sWi[h][e >> 2][e & 3] = 0.0; sDd[h][e >> 2][e & 3] = 0.0;
    }
    // raw strip of step `jt` (columns 4 jt ..) and raw diagonal tile of step jt + 1, from the accumulator tiles
    auto hand_over = [&](int jt) {
        const int jj0 = jt >> 2, q = jt & 3;
        if ((li >> 2) == q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sS[16 * wave + lq + 4 * r][li & 3] = a[jj0][r];
        }
        if (jt + 1 < IB / 4) {
            const int jjn = (jt + 1) >> 2, qn = (jt + 1) & 3;
            if (wave == jjn && (li >> 2) == qn) {                 // rows 4 qn + lq of the wave's 16: register qn
#pragma unroll
                for (int r = 0; r < 4; ++r)                        // (a loop with a constant test, not a[jjn][qn]: the direct form
                    if (r == qn) sN[lq][li & 3] = a[jjn][r];       //  cost the kernel 340 bytes of scratch a lane)
            }
        }
    };
    // the pivot lane's state: the current tile's inverse (lower) and the next tile (lower)
    double wi[4][4], dn[4][4];
    int bad_at = 0;
    auto factor_publish = [&](int jt, double (&d)[4][4]) {      // d: lower part of diagonal tile jt -> L_dd, inverse; publish both
        double rk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double piv = d[k][k];
            if (bad_at == 0 && !(piv > 0.0)) bad_at = 4 * jt + k + 1;                // also catches NaN; reported at the end
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
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < k) { wi[i][k] = 0.0; continue; }
                if (i == k) { wi[i][k] = rk[i]; continue; }
                double sacc = 0.0;
#pragma unroll
                for (int m = k; m < i; ++m) sacc = fma(d[i][m], wi[m][k], sacc);
                wi[i][k] = -rk[i] * sacc;
            }
        }
        struct alignas(16) D2 { double v[2]; };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int k = 0; k + 1 <= i; k += 2) {
                *reinterpret_cast<D2 *>(&sWi[jt & 1][i][k]) = D2{{wi[i][k], wi[i][k + 1]}};
                *reinterpret_cast<D2 *>(&sDd[jt & 1][i][k]) = D2{{d[i][k], (k + 1 <= i) ? d[i][k + 1] : 0.0}};
            }
            if ((i & 1) == 0) { sWi[jt & 1][i][i] = wi[i][i]; sDd[jt & 1][i][i] = d[i][i]; }
        }
    };
    // ---- prologue: strip 0 and tile 1 to LDS; the pivot lane factors tile 0 ----
    hand_over(0);
    __syncthreads();
    if (pivot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double row[4];
            load_frag32<double>(&sS[i][0], row);
#pragma unroll
            for (int k = 0; k < 4; ++k) dn[i][k] = (k <= i) ? row[k] : 0.0;
        }
        factor_publish(0, dn);
    }
    double al_prev = 0.0;
#pragma unroll
    for (int jt = 0; jt < IB / 4; ++jt) {
        const int c0 = 4 * jt, jj0 = jt >> 2, q = jt & 3;
        const int jjp = (jt - 1) >> 2, qp = (jt - 1) & 3;        // the step before (jt > 0)
        __syncthreads();                                          // B1: sS = raw strip jt, sN = raw tile jt + 1, inverse jt published
        if (stamps && jt == 9 && threadIdx.x == 0) stamps[0] = __builtin_amdgcn_s_memtime();
        if (stamps && jt == 10 && threadIdx.x == 0) stamps[7] = __builtin_amdgcn_s_memtime();
        // ---- pivot lane: D(jt + 1) ----
        if (pivot && jt + 1 < IB / 4) {
            double S[4][4], ls[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                load_frag32<double>(&sS[c0 + 4 + i][0], S[i]);
                double row[4];
                load_frag32<double>(&sN[i][0], row);
#pragma unroll
                for (int k = 0; k < 4; ++k) dn[i][k] = (k <= i) ? row[k] : 0.0;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    double v = S[r][0] * wi[c][0];
#pragma unroll
                    for (int k = 1; k <= c; ++k) v = fma(S[r][k], wi[c][k], v);
                    ls[r][c] = v;
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c <= r; ++c) {
                    double v = dn[r][c];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v = fma(-ls[r][k], ls[c][k], v);
                    dn[r][c] = v;
                }
        }
        if (stamps && jt == 9 && pivot) stamps[1] = __builtin_amdgcn_s_memtime();
        // ---- waves: X rows of the step before (B operand in place in the registers) ----
        if (jt > 0 && wave == jjp) {
            const double wprev = (li < 4) ? sWi[(jt - 1) & 1][li][lq] : 0.0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (jj > jjp) continue;                         // X is lower triangular: nothing to the right of the step
                const v4 u = M::mfma(wprev, x[jj][qp], zero);
                x[jj][qp] = u[0];
                sXd[lq][16 * jj + li] = u[0];
            }
        }
        // ---- S3: new strip = strip . inv(L_dd)^T;  S4: to LDS (the diagonal rows take L_dd itself) ----
        // (a wave whose 16 rows all lie above the step has nothing left to do but keep the barriers: wave w is active
        //  while jt < 4 (w + 1); from step 4 on wave 0 is the pivot lane's alone)
        const bool active = wave >= jj0;
        if (active) {
            const double winv = (li < 4) ? sWi[jt & 1][li][lq] : 0.0;   // B operand: B[k][n] = Winv[n][k]
            const v4 t = M::mfma(sS[16 * wave + li][lq], winv, zero);
            if (li < 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * wave + lq + 4 * r;
                    sLn[row][li] = (row >= c0 && row < c0 + 4) ? sDd[jt & 1][row - c0][li] : t[r];
                }
            }
        }
        if (stamps && jt == 9 && threadIdx.x == 192) stamps[2] = __builtin_amdgcn_s_memtime();
        __syncthreads();                                          // B2: sLn, sXd are in
        if (stamps && jt == 9 && threadIdx.x == 0) stamps[3] = __builtin_amdgcn_s_memtime();
        // ---- pivot lane: factor + invert tile jt + 1, publish (read at the next B1) ----
        if (pivot && jt + 1 < IB / 4) factor_publish(jt + 1, dn);
        if (stamps && jt == 9 && pivot) stamps[4] = __builtin_amdgcn_s_memtime();
        // ---- S5 ----
        double al = 0.0;
        if (active) {
            if ((li >> 2) == q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a[jj0][r] = sLn[16 * wave + lq + 4 * r][li & 3];
            }
            const int arow = 16 * wave + li;
            al = (arow > c0 + 3) ? -sLn[arow][lq] : 0.0;          // rows at or above the step: no update
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                // the tile that holds the NEXT step's strip first
                const int tj = (jj + ((jt + 1) >> 2)) & 3;
                if (tj >= jj0 && tj <= wave) {                      // (tiles above the diagonal: never read)
                    const int bcol = 16 * tj + li;
                    const double bl = (bcol > c0 + 3) ? sLn[bcol][lq] : 0.0;      // columns up to the step are final
                    a[tj] = M::mfma(al, bl, a[tj]);
                }
            }
        }
        if (jt + 1 < IB / 4 && wave >= ((jt + 1) >> 2)) hand_over(jt + 1);   // raw strip jt + 1 (and raw tile jt + 2) for the next step
        if (jt > 0 && wave >= jjp) {                              // (al_prev is zero for a wave above step jt - 1)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                if (jj <= jjp) x[jj] = M::mfma(al_prev, sXd[lq][16 * jj + li], x[jj]);
        }
        al_prev = al;
        if (stamps && jt == 9 && threadIdx.x == 192) stamps[5] = __builtin_amdgcn_s_memtime();
        if (stamps && jt == 10 && threadIdx.x == 0) stamps[6] = __builtin_amdgcn_s_memtime();
    }
    if (pivot && bad_at != 0) atomicCAS(info, 0, (int)(j0 + bad_at));        // first failure wins
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



// ---- round 4, second step: the same leaf as a ROLLED loop ------------------------------------------------------------
// Both leaves above are fully unrolled -- 16 steps of straight-line code that each run exactly once -- because the tile
// (jt >> 2) and the register (jt & 3) a step touches had to be compile-time constants.  That made the fp64 panel kernel
// 79 - 92 KB of code against a 64 KB instruction cache (shared by two CUs, and with the trailing update's kernel): every
// step fetches its instructions cold, ~12 cycles an instruction instead of the 4 - 7 the arithmetic needs (measured:
// factor64_mfma2's pivot lane took 1228 cycles for 80 independent fmas; the whole step 2712 instead of ~900).
// Here ONE step body runs 16 times:
//   * the current column tile is always a[0]: when a block of four steps ends, a[0] -- final -- is copied into out[]
//     by a select over the block index and the tiles shift left (24 register moves per block);
//   * the register of a tile that holds the step's rows (jt & 3, dynamic now) is read / written through 4-way selects;
//   * everything else that depended on jt is either an LDS address, a lane mask, or a wave-uniform branch.
// Same two-barrier schedule, same arithmetic, same results as factor64_mfma2.
__device__ __forceinline__ double sel4(const PM<double>::v4 &v, int k)
{
    return k == 0 ? v[0] : (k == 1 ? v[1] : (k == 2 ? v[2] : v[3]));
}
__device__ __forceinline__ void put4(PM<double>::v4 &v, int k, double val)
{
    v[0] = k == 0 ? val : v[0]; v[1] = k == 1 ? val : v[1]; v[2] = k == 2 ? val : v[2]; v[3] = k == 3 ? val : v[3];
}

__device__ __forceinline__ void factor64_mfma3(PM<double>::v4 (&a)[4], PM<double>::v4 (&x)[4], int64_t j0, int *__restrict__ info,
                                               int wave, int lane, unsigned long long *stamps = nullptr)
{
    typedef PM<double> M;
    typedef M::v4 v4;
    __shared__ __attribute__((aligned(16))) double sS[IB][4];   // the step's raw column strip (rows of the block)
    __shared__ __attribute__((aligned(16))) double sN[4][4];    // the NEXT step's raw diagonal tile
    __shared__ double sLn[IB][4];         // the step's finished strip of L
    __shared__ __attribute__((aligned(16))) double sWi[2][4][4];   // inv(L_dd) of the step (by parity), zero above the diagonal
    __shared__ __attribute__((aligned(16))) double sDd[2][4][4];   // L_dd (by parity), zero above the diagonal
    __shared__ double sXd[4][IB];         // four rows of X (of the step before: the X side runs one step behind)
    const int li = lane & 15, lq = lane >> 4;
    const v4 zero = {0.0, 0.0, 0.0, 0.0};
    const bool pivot = wave == 0 && lane == 0;
    if (threadIdx.x < 32) {
        const int h = threadIdx.x >> 4, e = threadIdx.x & 15;
        sWi[h][e >> 2][e & 3] = 0.0; sDd[h][e >> 2][e & 3] = 0.0;
    }
    v4 out[4] = {a[0], a[1], a[2], a[3]};             // finished tiles (by column tile); a[] becomes "current tile first"
    // raw strip of step jt (columns 4 jt .., in the CURRENT tile a[0]) and raw diagonal tile of step jt + 1 (in a[0], or
    // in a[1] when step jt + 1 opens the next block; always called with the tiles already shifted for step jt)
    auto hand_over = [&](int jt) {
        const int q = jt & 3;
        if ((li >> 2) == q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sS[16 * wave + lq + 4 * r][li & 3] = a[0][r];
        }
        if (jt + 1 < IB / 4) {
            const int jjn = (jt + 1) >> 2, qn = (jt + 1) & 3;
            if (wave == jjn && (li >> 2) == qn) sN[lq][li & 3] = qn == 0 ? a[1][0] : sel4(a[0], qn);   // (qn == 0: the next block's tile)
        }
    };
    double wi[4][4], dn[4][4];
    int bad_at = 0;
    auto factor_publish = [&](int jt, double (&d)[4][4]) {      // d: lower part of diagonal tile jt -> L_dd, inverse; publish both
        double rk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double piv = d[k][k];
            if (bad_at == 0 && !(piv > 0.0)) bad_at = 4 * jt + k + 1;                // also catches NaN; reported at the end
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
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < k) { wi[i][k] = 0.0; continue; }
                if (i == k) { wi[i][k] = rk[i]; continue; }
                double sacc = 0.0;
#pragma unroll
                for (int m = k; m < i; ++m) sacc = fma(d[i][m], wi[m][k], sacc);
                wi[i][k] = -rk[i] * sacc;
            }
        }
        struct alignas(16) D2 { double v[2]; };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int k = 0; k + 1 <= i; k += 2) {
                *reinterpret_cast<D2 *>(&sWi[jt & 1][i][k]) = D2{{wi[i][k], wi[i][k + 1]}};
                *reinterpret_cast<D2 *>(&sDd[jt & 1][i][k]) = D2{{d[i][k], (k + 1 <= i) ? d[i][k + 1] : 0.0}};
            }
            if ((i & 1) == 0) { sWi[jt & 1][i][i] = wi[i][i]; sDd[jt & 1][i][i] = d[i][i]; }
        }
    };
    // ---- prologue: strip 0 and tile 1 to LDS; the pivot lane factors tile 0 ----
    hand_over(0);
    __syncthreads();
    if (pivot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double row[4];
            load_frag32<double>(&sS[i][0], row);
#pragma unroll
            for (int k = 0; k < 4; ++k) dn[i][k] = (k <= i) ? row[k] : 0.0;
        }
        factor_publish(0, dn);
    }
    double al_prev = 0.0;
#pragma unroll 1
    for (int jt = 0; jt < IB / 4; ++jt) {
        const int c0 = 4 * jt, jj0 = jt >> 2, q = jt & 3;
        const int jjp = (jt - 1) >> 2, qp = (jt - 1) & 3;        // the step before (jt > 0)
        __syncthreads();                                          // B1: sS = raw strip jt, sN = raw tile jt + 1, inverse jt published
        if (stamps && jt == 9 && threadIdx.x == 0) stamps[0] = __builtin_amdgcn_s_memtime();
        if (stamps && jt == 10 && threadIdx.x == 0) stamps[7] = __builtin_amdgcn_s_memtime();
        // ---- pivot lane: D(jt + 1) = N - (S Wi^T)(S Wi^T)^T ----
        if (pivot && jt + 1 < IB / 4) {
            double S[4][4], ls[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                load_frag32<double>(&sS[c0 + 4 + i][0], S[i]);
                double row[4];
                load_frag32<double>(&sN[i][0], row);
#pragma unroll
                for (int k = 0; k < 4; ++k) dn[i][k] = (k <= i) ? row[k] : 0.0;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    double v = S[r][0] * wi[c][0];
#pragma unroll
                    for (int k = 1; k <= c; ++k) v = fma(S[r][k], wi[c][k], v);
                    ls[r][c] = v;
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c <= r; ++c) {
                    double v = dn[r][c];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v = fma(-ls[r][k], ls[c][k], v);
                    dn[r][c] = v;
                }
        }
        if (stamps && jt == 9 && pivot) stamps[1] = __builtin_amdgcn_s_memtime();
        // ---- waves: X rows of the step before (B operand in place in the registers) ----
        if (jt > 0 && wave == jjp) {
            const double wprev = (li < 4) ? sWi[(jt - 1) & 1][li][lq] : 0.0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (jj > jjp) continue;                         // X is lower triangular: nothing to the right of the step
                const v4 u = M::mfma(wprev, sel4(x[jj], qp), zero);
                put4(x[jj], qp, u[0]);
                sXd[lq][16 * jj + li] = u[0];
            }
        }
        // ---- S3: new strip = strip . inv(L_dd)^T;  S4: to LDS (the diagonal rows take L_dd itself) ----
        const bool active = wave >= jj0;                          // (a wave whose rows all lie above the step only keeps the barriers)
        if (active) {
            const double winv = (li < 4) ? sWi[jt & 1][li][lq] : 0.0;   // B operand: B[k][n] = Winv[n][k]
            const v4 t = M::mfma(sS[16 * wave + li][lq], winv, zero);
            if (li < 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * wave + lq + 4 * r;
                    const bool diag = row >= c0 && row < c0 + 4;
                    const double ld = sDd[jt & 1][diag ? row - c0 : 0][li];
                    sLn[row][li] = diag ? ld : t[r];
                }
            }
        }
        if (stamps && jt == 9 && threadIdx.x == 192) stamps[2] = __builtin_amdgcn_s_memtime();
        __syncthreads();                                          // B2: sLn, sXd are in
        if (stamps && jt == 9 && threadIdx.x == 0) stamps[3] = __builtin_amdgcn_s_memtime();
        // ---- pivot lane: factor + invert tile jt + 1, publish (read at the next B1) ----
        if (pivot && jt + 1 < IB / 4) factor_publish(jt + 1, dn);
        if (stamps && jt == 9 && pivot) stamps[4] = __builtin_amdgcn_s_memtime();
        // ---- S5 ----
        double al = 0.0;
        if (active) {
            if ((li >> 2) == q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a[0][r] = sLn[16 * wave + lq + 4 * r][li & 3];
            }
            const int arow = 16 * wave + li;
            al = (arow > c0 + 3) ? -sLn[arow][lq] : 0.0;          // rows at or above the step: no update
#pragma unroll
            for (int t = 0; t < 4; ++t) {                          // a[t]: column tile jj0 + t
                if (jj0 + t <= wave) {                              // (tiles above the diagonal: never read)
                    const int bcol = 16 * (jj0 + t) + li;
                    const double bl = (bcol > c0 + 3) ? sLn[bcol][lq] : 0.0;      // columns up to the step are final
                    a[t] = M::mfma(al, bl, a[t]);
                }
            }
            if (q == 3) {                                           // the block's tile is final: keep it, shift the tiles left
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[t][r] = (t == jj0) ? a[0][r] : out[t][r];
                }
                a[0] = a[1]; a[1] = a[2]; a[2] = a[3];
            }
        }
        if (jt + 1 < IB / 4 && wave >= ((jt + 1) >> 2)) hand_over(jt + 1);   // raw strip jt + 1 (and raw tile jt + 2) for the next step
        if (jt > 0 && wave >= jjp) {                              // (al_prev is zero for a wave above step jt - 1)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                if (jj <= jjp) x[jj] = M::mfma(al_prev, sXd[lq][16 * jj + li], x[jj]);
        }
        al_prev = al;
        if (stamps && jt == 9 && threadIdx.x == 192) stamps[5] = __builtin_amdgcn_s_memtime();
    }
    if (pivot && bad_at != 0) atomicCAS(info, 0, (int)(j0 + bad_at));        // first failure wins
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = out[t];
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
