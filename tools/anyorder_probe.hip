// tools/anyorder_probe.hip -- does hipExtAnyOrderLaunch (a dispatch packet WITHOUT the barrier bit) let a kernel overlap its
// predecessor in the same stream on this chip / runtime?  Launch a 300 us spin kernel, then a tiny kernel in the same stream:
//   plain launch        -> the tiny kernel starts after the spin ends (in-order);
//   hipExtAnyOrderLaunch -> if honoured, it starts while the spin is still running.
// Then the pattern the factorisation would use: [barrier kernel P] [any-order kernel U] [barrier kernel P'] ...: P' must start
// only after BOTH P and U ended.
// build: hipcc -O2 --offload-arch=gfx950 tools/anyorder_probe.hip -o tools/anyorder_probe   (diagnostic)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err__)); return 1; } } while (0)
__global__ void stamp(unsigned long long *out, int slot, int us)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[2 * slot] = t0; out[2 * slot + 1] = __builtin_amdgcn_s_memrealtime(); }
}
int main()
{
    unsigned long long *d = nullptr, h[64];
    CK(hipMalloc(&d, sizeof(h)));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(d, 0, sizeof(h)));
        // 0: spin 300 us (plain)   1: tiny, plain      2: spin 300 (plain)   3: tiny, any-order
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, d, 0, 300);
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, d, 1, 1);
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, d, 2, 300);
        hipExtLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, 3, 1);
        // the factorisation's pattern: P(100 us, barrier)  U(300 us, any order)  P'(100 us, barrier)  U'(300, any)  P''(1, barrier)
        hipLaunchKernelGGL(stamp, dim3(4), dim3(64), 0, s, d, 4, 100);
        hipExtLaunchKernelGGL(stamp, dim3(512), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, 5, 300);
        hipLaunchKernelGGL(stamp, dim3(4), dim3(64), 0, s, d, 6, 100);
        hipExtLaunchKernelGGL(stamp, dim3(512), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, 7, 300);
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, d, 8, 1);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        const double t0 = (double)h[0];
        auto us = [&](int i) { return ((double)h[i] - t0) / 100.0; };
        printf("rep %d\n", rep);
        printf("  plain:     spin [%.1f, %.1f]  tiny starts %.1f  (%.1f us after the spin's end)\n", us(0), us(1), us(2), us(2) - us(1));
        printf("  any-order: spin [%.1f, %.1f]  tiny starts %.1f  (%.1f us after the spin's START: overlap %s)\n", us(4), us(5), us(6), us(6) - us(4),
               us(6) < us(5) ? "YES" : "no");
        printf("  pattern:   P [%.1f, %.1f]  U(any) [%.1f, %.1f]  P' [%.1f, %.1f]  U'(any) [%.1f, %.1f]  P'' starts %.1f\n", us(8), us(9), us(10), us(11),
               us(12), us(13), us(14), us(15), us(16));
        printf("             U starts %.1f us after P starts; P' starts %.1f us after max(P, U) ended; U' starts %.1f after P' starts\n", us(10) - us(8),
               us(12) - (us(9) > us(11) ? us(9) : us(11)), us(14) - us(12));
    }
    return 0;
}
