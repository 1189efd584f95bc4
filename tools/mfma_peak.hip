// tools/mfma_peak.hip -- what does the chip sustain on back-to-back fp64 / fp32 MFMA?
// (diagnostic only; not part of libgpx).  hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512, 2) void k64(double *out, int iters, double seed)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[gridDim.x * blockDim.x] = (double)(t1 - t0);
        out[gridDim.x * blockDim.x + 1] = (double)(r1 - r0);
    }
}

template <int NACC>
__global__ __launch_bounds__(512, 2) void k32(float *out, int iters, float seed)
{
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
    float a = seed + threadIdx.x * 1e-3f, b = seed - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char **argv)
{
    const int blocks = 256, threads = 512, NACC = 16;
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;      // 4000 ~ 1.7 ms; 1000000 ~ 0.43 s sustained
    double *d; hipMalloc(&d, ((size_t)blocks * 4 * threads + 16) * sizeof(double));   // sized for the largest launch below
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k64<NACC>), dim3(blocks), dim3(threads), 0, 0, d, iters, 0.3);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * (threads / 64) * iters * NACC * 2048.0;
        double tm[2]; hipMemcpy(tm, d + blocks * threads, 16, hipMemcpyDeviceToHost);
        printf("f64 1blk/CU(8 waves): %.3f ms  %.2f TF/s  clk=%.0f MHz (memtime/memrealtime*100)\n", ms, fl / ms / 1e9,
               tm[0] / tm[1] * 100.0);
    }
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k64<NACC>), dim3(blocks * 4), dim3(threads), 0, 0, d, iters, 0.3);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * 4 * (threads / 64) * iters * NACC * 2048.0;
        printf("f64 4x blocks: %.3f ms  %.2f TF/s\n", ms, fl / ms / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k32<NACC>), dim3(blocks * 2), dim3(threads), 0, 0, (float *)d, iters, 0.3f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * 2 * (threads / 64) * iters * NACC * 2048.0;
        printf("f32 16x16x4: %.3f ms  %.2f TF/s\n", ms, fl / ms / 1e9);
    }
    return 0;
}
