// tools/sync_probe.hip -- what does each kind of dependency between two tiny kernels cost on this runtime?
// build: hipcc -O2 --offload-arch=gfx950 tools/sync_probe.hip -o tools/sync_probe   (diagnostic)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err__)); return 1; } } while (0)
__global__ void tiny(unsigned long long *out, int slot) { if (threadIdx.x == 0) out[slot] = __builtin_amdgcn_s_memrealtime(); }
__global__ void spin(unsigned long long *out, int us) { unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100) __builtin_amdgcn_s_sleep(32); if (threadIdx.x == 0) out[0] = t0; }
int main()
{
    const int N = 64;
    unsigned long long *d = nullptr, h[N];
    CK(hipMalloc(&d, N * 8));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(4 * N);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    auto report = [&](const char *what) {
        hipDeviceSynchronize();
        hipMemcpy(h, d, N * 8, hipMemcpyDeviceToHost);
        double sum = 0; int cnt = 0;
        for (int i = 9; i < N; ++i) { sum += (double)(h[i] - h[i - 1]) / 100.0; ++cnt; }
        printf("%-78s %6.2f us between consecutive kernels\n", what, sum / cnt);
        return 0;
    };
    for (int rep = 0; rep < 2; ++rep) {
        // (a) plain in-order launches on one stream; a long kernel first so that the host is far ahead
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 300);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, d, i);
        report("(a) same stream, nothing in between");
        // (b) hipEventRecord after every kernel
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 300);
        for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, d, i); hipEventRecord(ev[i], s1); }
        report("(b) + hipEventRecord after each");
        // (c) + wait on an event of the OTHER stream that completed long ago (recorded before the spin ended? no: after a short kernel)
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s2, d, 0);
        hipEventRecord(ev[N], s2);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 300);
        for (int i = 0; i < N; ++i) { hipStreamWaitEvent(s1, ev[N], 0); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, d, i); }
        report("(c) hipStreamWaitEvent(other stream's OLD event) before each");
        // (d) record + wait (as the factorisation does): kernel, record, wait(old event of s2)
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 300);
        for (int i = 0; i < N; ++i) { hipStreamWaitEvent(s1, ev[N], 0); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, d, i); hipEventRecord(ev[i], s1); }
        report("(d) wait(old event) + kernel + record");
        // (e) ping-pong: s1 kernel i waits for s2 kernel i-1 and vice versa (a true cross-stream chain)
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 300);
        hipEventRecord(ev[2 * N], s1);
        for (int i = 0; i < N; ++i) {
            hipStream_t a = (i & 1) ? s2 : s1;
            hipStreamWaitEvent(a, ev[2 * N + i], 0);
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d, i);
            hipEventRecord(ev[2 * N + i + 1], a);
        }
        report("(e) chain alternating between two streams (event hand-off each time)");
        // (f) hipExtLaunchKernelGGL with a stop event instead of hipEventRecord
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 300);
        for (int i = 0; i < N; ++i) hipExtLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, nullptr, ev[i], 0, d, i);
        report("(f) hipExtLaunchKernelGGL(..., stopEvent) each");
        // (g) the factorisation's pattern with two live streams: U on s2 (50 us), P on s1 (100 us); P(k+1) waits P(k) [order] and U(k-1) [event]
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 300);
        for (int i = 0; i < N; ++i) {
            if (i >= 2) hipStreamWaitEvent(s1, ev[3 * N + i - 2], 0);
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, d, i);
            hipEventRecord(ev[i], s1);
            hipStreamWaitEvent(s2, ev[i], 0);
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s2, d + 0, 0);
            hipEventRecord(ev[3 * N + i], s2);
        }
        report("(g) look-ahead pattern: P(k+1) after P(k) and U(k-1); U(k) after P(k)");
    }
    return 0;
}
