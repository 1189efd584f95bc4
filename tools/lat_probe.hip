// tools/lat_probe.hip -- dependent-operation latencies seen by ONE wave (s_memtime ticks per op; diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out, unsigned long long *t, double seed)
{
    __shared__ double sh[64];
    double a = seed + threadIdx.x, b = 1.0000001, c = 1e-9;
    unsigned long long t0, t1;
    // 1. dependent v_fma_f64
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a = fma(a, b, c);
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[0] = t1 - t0;
    // 2. dependent v_rsq_f64 (+1 fma to keep it bounded)
    double r = a * a + 2.0;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) r = __builtin_amdgcn_rsq(r) + 2.0;
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[1] = t1 - t0;
    // 3. dependent f32 fma
    float fa = (float)a, fb = 1.0000001f, fc = 1e-9f;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) fa = fmaf(fa, fb, fc);
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[2] = t1 - t0;
    // 4. LDS write -> read round trip (dependent)
    double v = r;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
        sh[threadIdx.x] = v;
        v = sh[(threadIdx.x + 1) & 63] + 1.0;
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[3] = t1 - t0;
    // 5. 4 independent chains of f64 fma (issue rate)
    double p0 = a, p1 = a + 1, p2 = a + 2, p3 = a + 3;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { p0 = fma(p0, b, c); p1 = fma(p1, b, c); p2 = fma(p2, b, c); p3 = fma(p3, b, c); }
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[4] = t1 - t0;
    // 6. workgroup barrier round trip (4 waves)
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) __syncthreads();
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[5] = t1 - t0;
    unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 4096; ++i) a = fma(a, b, c);
    t1 = __builtin_amdgcn_s_memtime();
    unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { t[6] = t1 - t0; t[7] = rt1 - rt0; }
    out[threadIdx.x] = a + r + fa + v + p0 + p1 + p2 + p3;
}
int main()
{
    double *o; unsigned long long *t, h[8];
    hipMalloc(&o, 256 * 8); hipMalloc(&t, 64);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, o, t, 1.5);
    hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
    printf("dependent v_fma_f64      : %.1f ticks/op\n", h[0] / 1024.0);
    printf("dependent v_rsq_f64 + add: %.1f ticks/pair\n", h[1] / 1024.0);
    printf("dependent v_fma_f32      : %.1f ticks/op\n", h[2] / 1024.0);
    printf("LDS write->read (dep.)   : %.1f ticks/round trip\n", h[3] / 256.0);
    printf("4 independent f64 chains : %.1f ticks per 4 fma\n", h[4] / 1024.0);
    printf("workgroup barrier (4 w.) : %.1f ticks\n", h[5] / 256.0);
    printf("s_memtime ticks per s_memrealtime tick (100 MHz): %.2f -> s_memtime = %.0f MHz\n", (double)h[6] / h[7], 100.0 * h[6] / h[7]);
    return 0;
}
