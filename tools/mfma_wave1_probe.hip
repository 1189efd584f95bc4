// tools/mfma_wave1_probe.hip -- cycles per v_mfma_f64_16x16x4 when ONE wave per SIMD issues them back to back, by the number of
// independent accumulators in rotation (the resident panel kernel's products have four; the GEMM's sixteen), and with two waves
// per SIMD.  Diagnostic only.  hipcc -O3 --offload-arch=gfx950 tools/mfma_wave1_probe.hip -o tools/mfma_wave1_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double *out, int iters, double seed)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (double)(t1 - t0);
}
template <int NACC> void run(double *d, int threads, int blocks, const char *what)
{
    const int iters = 4096 / NACC * 4;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NACC>), dim3(blocks), dim3(threads), 0, 0, d, iters, 0.3);
    hipDeviceSynchronize();
    double cyc; hipMemcpy(&cyc, d + (size_t)blocks * threads, 8, hipMemcpyDeviceToHost);
    printf("%-44s %2d accumulators: %.1f cycles per MFMA per wave (%d MFMAs)\n", what, NACC, cyc / (iters * NACC), iters * NACC);
}
int main()
{
    double *d; hipMalloc(&d, ((size_t)1024 * 512 + 16) * sizeof(double));
    run<1>(d, 256, 256, "one wave a SIMD (256 threads, 1 block a CU)");
    run<2>(d, 256, 256, "one wave a SIMD");
    run<4>(d, 256, 256, "one wave a SIMD");
    run<8>(d, 256, 256, "one wave a SIMD");
    run<16>(d, 256, 256, "one wave a SIMD");
    run<4>(d, 512, 256, "two waves a SIMD (512 threads)");
    run<16>(d, 512, 256, "two waves a SIMD (512 threads)");
    run<4>(d, 64, 256, "one wave a CU");
    return 0;
}
