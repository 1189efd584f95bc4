// gpx_kernels_dev.h -- device-side entries of the two kernel families (shared by gpx_kmat.hip and
// gpx_deriv.hip): gp/ext/gaussian_c.pyx:18-164 and gp/ext/periodic_c.pyx:18-235 per matrix entry.
#pragma once
#include "gpx_common.h"

namespace gpx {

template <typename T> struct Vec;
template <> struct Vec<double> { static constexpr int N = 2; typedef double2 type; };
template <> struct Vec<float>  { static constexpr int N = 4; typedef float4 type; };

template <typename T> __device__ __forceinline__ T dev_exp(T x);
template <> __device__ __forceinline__ double dev_exp<double>(double x) { return exp(x); }
template <> __device__ __forceinline__ float  dev_exp<float>(float x)  { return expf(x); }

// one kernel-matrix entry from the accumulated squared distance (gaussian) --
// FORM 0: c2*exp(e)   1: exp(e)*(c3*d2 - c2)   2: exp(e)*(c4*d2^2 - c3*d2 + c2)
// with the reference's underflow clamp  e < MIN -> 0  (gaussian_c.pyx:31-34)
template <typename T, int FORM>
__device__ __forceinline__ T gaussian_entry(T d2, T c1, T c2, T c3, T c4)
{
    const T e = c1 * d2;
    T v;
    if (FORM == 0)      v = c2 * dev_exp<T>(e);
    else if (FORM == 1) v = dev_exp<T>(e) * (c3 * d2 - c2);
    else                v = dev_exp<T>(e) * (c4 * (d2 * d2) - c3 * d2 + c2);
    return (e < (T)GPX_MIN_LOG) ? (T)0 : v;
}

// periodic members for d == 1 (periodic_c.pyx), dd = x1[i] - x2[j]
template <typename T>
__device__ __forceinline__ T periodic_entry(int member, T dd, T h, T w, T p)
{
    const T h2 = h * h, w2 = w * w, p2 = p * p;
    const T arg = (T)0.5 * dd / p;
    const T sn = sin(arg), cs = cos(arg);
    const T ex = dev_exp<T>((T)-2.0 * (sn * sn) / w2);
    const T w3 = w2 * w, w4 = w2 * w2, p4 = p2 * p2;
    switch (member) {
    case GPX_K:        return h2 * ex;                                                    // :30
    case GPX_DK_DH:    return (T)2.0 * h * ex;                                            // :65
    case GPX_DK_DW:    return (T)4.0 * h2 * ex * (sn * sn) / w3;                          // :80
    case GPX_DK_DP:    return (T)2.0 * dd * h2 * ex * sn * cs / (p2 * w2);                // :96
    case GPX_D2K_DHDH: return (T)2.0 * ex;                                                // :111
    case GPX_D2K_DHDW: return (T)8.0 * h * ex * (sn * sn) / w3;                           // :126
    case GPX_D2K_DHDP: return (T)4.0 * dd * h * ex * sn * cs / (p2 * w2);                 // :142
    case GPX_D2K_DWDW: return (T)-12.0 * h2 * ex * (sn * sn) / w4                         // :172
                              + (T)16.0 * h2 * ex * (sn * sn) * (sn * sn) / (w4 * w2);
    case GPX_D2K_DWDP: return (T)-4.0 * dd * h2 * ex * sn * cs / (p2 * w3)                // :188
                              + (T)8.0 * dd * h2 * ex * (sn * sn * sn) * cs / (p2 * w3 * w2);
    default:           return (dd * dd) * h2 * ex * (sn * sn) / (p4 * w2)                 // :235
                              - (dd * dd) * h2 * ex * (cs * cs) / (p4 * w2)
                              + (T)4.0 * (dd * dd) * h2 * ex * (sn * sn) * (cs * cs) / (p4 * w4)
                              - (T)4.0 * dd * h2 * ex * sn * cs / (p2 * p * w2);
    }
}

}  // namespace gpx
