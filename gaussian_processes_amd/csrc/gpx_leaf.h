// gpx_leaf.h -- the register-resident 64 x 64 leaf shared by the factorisation (gpx_potrf.hip) and the
// batched inverses of the triangular solves (gpx_solve.hip).
#pragma once
#include "gpx_common.h"

namespace gpx {

constexpr int IB = 64;

__device__ __forceinline__ double fast_rsqrt(double p)
{
    double y = __builtin_amdgcn_rsq(p);
    y = y * fma(-0.5 * p * y, y, 1.5);
    y = y * fma(-0.5 * p * y, y, 1.5);
    return y;
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
    __shared__ T pan[IB][4];          // the step's 4 finished columns of L, rows below the diagonal tile
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
                    if (*info == 0) *info = (int)(j0 + 4 * jt + k + 1);
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

}  // namespace gpx
