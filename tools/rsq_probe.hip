// tools/rsq_probe.hip -- accuracy of v_rsq_f64 and of the refinements the leaf could use (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *o0, double *o1, double *o2, double *o3, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double p = x[i];
    double y = __builtin_amdgcn_rsq(p);
    o0[i] = y;
    double y1 = y * fma(-0.5 * p * y, y, 1.5);            // one Newton step
    o1[i] = y1;
    double y2 = y1 * fma(-0.5 * p * y1, y1, 1.5);         // two
    o2[i] = y2;
    double t = p * y, u = fma(-t, y, 1.0);                // third order, one step
    o3[i] = fma(fma(0.375, u, 0.5), u * y, y);
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> h(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 40) - 20); }
    double *dx, *d[4];
    hipMalloc(&dx, n * 8); for (auto &p : d) hipMalloc(&p, n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d[0], d[1], d[2], d[3], n);
    const char *names[4] = {"v_rsq_f64", "+1 Newton", "+2 Newton", "+1 third-order"};
    for (int v = 0; v < 4; ++v) {
        std::vector<double> o(n);
        hipMemcpy(o.data(), d[v], n * 8, hipMemcpyDeviceToHost);
        long double worst = 0;
        for (int i = 0; i < n; ++i) {
            long double ex = 1.0L / sqrtl((long double)h[i]);
            long double e = fabsl(((long double)o[i] - ex) / ex);
            if (e > worst) worst = e;
        }
        printf("%-16s max rel err %.3Le = 2^%.1Lf  (%.2Lf ulp of 2^-53)\n", names[v], worst, log2l(worst), worst / 1.1102230246251565e-16L);
    }
    return 0;
}
