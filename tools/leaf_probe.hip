// tools/leaf_probe.hip -- the two-wave 64 x 64 leaf (csrc/gpx_leaf.h, factor64_wave) alone: L and W = inv(L) of a random SPD block
// against a host Cholesky, and the leaf's duration in core cycles (s_memtime) with the CU to itself.  diagnostic.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DSTAMP=<step>] [-DPROBE_RING=4] -Igaussian_processes_amd/csrc tools/leaf_probe.hip -o tools/leaf_probe
#include "gpx_leaf.h"
#include <cstdio>
#include <cmath>
#include <vector>
using namespace gpx;
constexpr int PT = 66;
#ifndef STAMP
#define STAMP -1
#endif
#ifndef PROBE_RING
#define PROBE_RING 16
#endif
__global__ __launch_bounds__(256, 1) void k(const double *A, double *L, double *W, int *info, unsigned long long *st, int reps)
{
    __shared__ double sA[64][PT], sB[64][PT], sBuf[W1_BUF_DOUBLES];
    __shared__ int sCtl[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int rep = 0; rep < reps; ++rep) {
        for (int i = tid; i < 64 * 64; i += 256) { sA[i / 64][i % 64] = A[i]; sB[i / 64][i % 64] = -7.0; }
        if (tid < 2) sCtl[tid] = 0;
        __syncthreads();
        if (wave < 2) factor64_wave<PT, PROBE_RING, STAMP>(sA, sB, sBuf, sCtl, wave, 0, info, lane, st);
        __syncthreads();
        if (tid == 0) st[50] = __builtin_amdgcn_s_memtime();
    }
    for (int i = tid; i < 64 * 64; i += 256) { L[i] = sA[i / 64][i % 64]; W[i] = sB[i / 64][i % 64]; }
}
int main()
{
    const int n = 64;
    std::vector<double> A(n * n), G(n * n), L(n * n, 0.0), Lg(n * n), Wg(n * n);
    unsigned s = 12345;
    for (auto &g : G) { s = s * 1664525u + 1013904223u; g = (double)(s >> 8) / 16777216.0 - 0.5; }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double v = 0; for (int k = 0; k < n; ++k) v += G[i * n + k] * G[j * n + k]; A[i * n + j] = v + (i == j ? 1.0 : 0.0); }
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j]; for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
        L[j * n + j] = sqrt(d);
        for (int i = j + 1; i < n; ++i) { double v = A[i * n + j]; for (int k = 0; k < j; ++k) v -= L[i * n + k] * L[j * n + k]; L[i * n + j] = v / L[j * n + j]; }
    }
    double *dA, *dL, *dW; int *dinfo; unsigned long long *dst;
    hipMalloc(&dA, n * n * 8); hipMalloc(&dL, n * n * 8); hipMalloc(&dW, n * n * 8); hipMalloc(&dinfo, 4); hipMalloc(&dst, 64 * 8);
    hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice); hipMemset(dinfo, 0, 4); hipMemset(dst, 0, 64 * 8);
    for (int it = 0; it < 3; ++it) { hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, dA, dL, dW, dinfo, dst, 50); hipDeviceSynchronize(); }
    unsigned long long st[64]; int info;
    hipMemcpy(Lg.data(), dL, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(Wg.data(), dW, n * n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(st, dst, 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
    double eL = 0, eW = 0; int firstbad = -1;
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { const double e = fabs(Lg[i * n + j] - L[i * n + j]); if (e > eL) eL = e; if (e > 1e-9 && firstbad < 0) firstbad = i * 64 + j; }
    // W L = I
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double v = 0; for (int k = 0; k < n; ++k) v += Wg[i * n + k] * (k >= j ? L[k * n + j] : 0.0); const double e = fabs(v - (i == j)); if (e > eW) eW = e; }
    printf("info %d  max |L - L_host| %.3e (first bad entry: row %d col %d)  max |W L - I| %.3e\n", info, eL, firstbad / 64, firstbad % 64, eW);
    for (int i = 0; i < n; i += 4) { for (int j = 0; j <= i; j += 4) { double e = 0; for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) if (j + b <= i + a) e = fmax(e, fabs(Lg[(i + a) * n + j + b] - L[(i + a) * n + j + b])); printf("%c", e < 1e-10 ? '.' : 'X'); } printf("\n"); }
    printf("max error per 4-column step:"); for (int j = 0; j < n; j += 4) { double e = 0; for (int i = j; i < n; ++i) for (int b = 0; b < 4; ++b) if (j + b <= i) e = fmax(e, fabs(Lg[i * n + j + b] - L[i * n + j + b])); printf(" %.1e", e); } printf("\n");
    printf("L[16][0..3] gpu %g %g %g %g host %g %g %g %g\n", Lg[16 * n], Lg[16 * n + 1], Lg[16 * n + 2], Lg[16 * n + 3], L[16 * n], L[16 * n + 1], L[16 * n + 2], L[16 * n + 3]);
    printf("L[17][0..3] gpu %g %g %g %g host %g %g %g %g\n", Lg[17 * n], Lg[17 * n + 1], Lg[17 * n + 2], Lg[17 * n + 3], L[17 * n], L[17 * n + 1], L[17 * n + 2], L[17 * n + 3]);
    printf("leaf: wave 0 %.2f us, %llu core cycles = %llu a step; with wave 1's tail and the barrier %llu core cycles\n", (st[17] - st[16]) / 100.0, st[33] - st[32], (st[33] - st[32]) / 16, st[50] - st[32]);
    if (STAMP >= 0) {
        const char *nm[] = {"crit strip+am+crit update", "D0", "gather", "D1+pivot0", "D2+pivot1", "D3+pivot2", "D4+pivot3", "rest of the MFMAs", "stores"};
        printf("step %d, issue times (core cycles):", STAMP);
        for (int i = 0; i < 9; ++i) printf("  %s %llu", nm[i], st[41 + i] - st[40 + i]);
        printf("  | whole step %llu\n", st[49] - st[40]);
    }
    return (eL < 1e-10 && eW < 1e-9) ? 0 : 1;
}
