// tools/cumask_probe.hip -- does hipExtStreamCreateWithCUMask work here, and which CUs does bit i select?
// hipcc -O2 --offload-arch=gfx950 tools/cumask_probe.hip -o tools/cumask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>
#include <vector>
__global__ void probe(unsigned *out, int spin)
{
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        out[blockIdx.x] = ((xcc & 0xF) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF);
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    }
}
static void run(hipStream_t s, const char *name, unsigned *d, int nb)
{
    hipMemsetAsync(d, 0xFF, nb * 4, s);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(64), 0, s, d, 20000);
    hipError_t e = hipStreamSynchronize(s);
    std::vector<unsigned> h(nb);
    hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus; int per_xcc[16] = {0};
    for (unsigned v : h) cus.insert(v);
    for (unsigned v : cus) per_xcc[(v >> 16) & 0xF]++;
    printf("%-28s status=%s distinct CUs=%zu  per XCC:", name, hipGetErrorString(e), cus.size());
    for (int i = 0; i < 8; ++i) printf(" %d", per_xcc[i]);
    printf("\n");
}
int main()
{
    unsigned *d; const int nb = 8192;
    if (hipMalloc(&d, nb * 4) != hipSuccess) return 1;
    hipStream_t s0; hipStreamCreate(&s0);
    run(s0, "unmasked", d, nb);
    struct { const char *name; unsigned m[8]; } cases[] = {
        {"bits 0..15", {0x0000FFFFu, 0, 0, 0, 0, 0, 0, 0}},
        {"bits 0..31", {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0}},
        {"bits 16..255", {0xFFFF0000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}},
        {"every 16th bit", {0x00010001u, 0x00010001u, 0x00010001u, 0x00010001u, 0x00010001u, 0x00010001u, 0x00010001u, 0x00010001u}},
    };
    for (auto &c : cases) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, c.m);
        if (e != hipSuccess) { printf("%-28s create failed: %s\n", c.name, hipGetErrorString(e)); continue; }
        run(s, c.name, d, nb);
        hipStreamDestroy(s);
    }
    return 0;
}
