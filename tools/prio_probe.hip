// tools/prio_probe.hip -- does a workgroup that needs a WHOLE CU's registers (512 VGPRs per lane) on a high-priority
// stream get placed while a long low-priority grid of small workgroups keeps every CU busy?  Prints when the big
// workgroups started relative to the filler's start, and how long the filler took with / without them.
// build: hipcc --offload-arch=gfx950 -O3 tools/prio_probe.hip -o tools/prio_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void filler(unsigned long long *sink, int iters)
{
    // ~132 VGPRs are not needed to make the point: a plain busy loop of FMAs, ~50 us per workgroup
    double a = threadIdx.x * 1e-3, b = 1.000001;
    for (int i = 0; i < iters; ++i) a = fma(a, b, 1e-9);
    if (a == 12345.678) sink[0] = 1;
}

__global__ __launch_bounds__(256) void big(unsigned long long *stamps, int iters)
{
    asm volatile("; claim the whole register file" ::: "v255", "a255");
    if (threadIdx.x == 0) stamps[blockIdx.x * 2] = wall_clock64();
    double a = threadIdx.x * 1e-3, b = 1.000001;
    for (int i = 0; i < iters; ++i) a = fma(a, b, 1e-9);
    if (a == 12345.678) stamps[63] = 1;
    if (threadIdx.x == 0) stamps[blockIdx.x * 2 + 1] = wall_clock64();
}

__global__ void stamp(unsigned long long *p) { p[0] = wall_clock64(); }

int main()
{
    unsigned long long *d = nullptr;
    CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    hipStream_t lo, hi;
    int least, greatest;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    CK(hipStreamCreateWithPriority(&lo, hipStreamNonBlocking, least));
    CK(hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, greatest));
    const int iters = 12000;                 // filler workgroup ~50 us
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int with_big = 0; with_big < 2; ++with_big) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(d, 0, 4096));
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, lo, d + 100);
            CK(hipEventRecord(e0, lo));
            hipLaunchKernelGGL(filler, dim3(512 * 40), dim3(256), 0, lo, d + 64, iters);   // ~40 rounds of 2 WG / CU
            CK(hipEventRecord(e1, lo));
            if (with_big) hipLaunchKernelGGL(big, dim3(4), dim3(256), 0, hi, d, iters / 4);
            CK(hipDeviceSynchronize());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(128);
            CK(hipMemcpy(h.data(), d, 128 * 8, hipMemcpyDeviceToHost));
            printf("with_big=%d filler %.3f ms", with_big, ms);
            if (with_big) for (int w = 0; w < 4; ++w) printf("  wg%d start +%.1f us dur %.1f us", w, (double)(h[2 * w] - h[100]) / 100.0, (double)(h[2 * w + 1] - h[2 * w]) / 100.0);
            printf("\n");
        }
    }
    return 0;
}
