// gpx_deriv.hip -- the second-derivative stack of the GP API, device resident (gfx950).
//
// Replaces the dense-product glue of gp/ext/gp_c.pyx:52-131 (dlh_dtheta, d2lh_dtheta2, dm_dtheta), which
// the reference evaluates as O(n_p^2) np.dot calls on (n, n) matrices.  With W = K^-1 (full, in HBM),
// dK_i the kernel derivative w.r.t. parameter i (dK_s = 2 s I for the noise), alpha = K^-1 y:
//     v_i = dK_i alpha          fused mat-vec, derivative evaluated on the fly (no Jacobian matrix)
//     u_i = K^-1 v_i            two triangular solves with the resident factor
//     M_i = W dK_i              ONE n x n x n product per kernel parameter on the MFMA kernel
//     a_i = alpha . v_i     tr_i = trace(M_i)     vu_ij = v_j . u_i     T_ij = trace(M_j M_i)
//     q_ij = alpha^T d2K_ij alpha     h_ij = trace(W d2K_ij)         (second derivatives on the fly)
// and (gp_c.pyx:70-111, every term restated in these symbols)
//     dlh_i   = 1/2 lh (a_i - tr_i)                                                     gp_c.pyx:52-67
//     d2lh_ij = 1/2 [ dlh_j (a_i - tr_i) + lh (-2 vu_ij + q_ij + T_ij - h_ij) ]
//     dm_i    = dK_i(xo, x) alpha - K(xo, x) u_i                                        gp_c.pyx:114-131
// Only the (n_p + 1)^2 scalars (or the (n_p + 1) x m matrix of dm) return to the host.
#include "gpx_gp_internal.h"
#include "gpx_kernels_dev.h"
#include <cmath>
#include <vector>

extern "C" int gpx_d_mean_member(int dtype, int kernel, int member, const void *xo, int64_t m, const void *x,
                                 int64_t n, int d, const double *params, const void *alpha, void *out,
                                 void *stream);

namespace gpx {

constexpr int DR_T = 64;             // tile edge of the reductions
constexpr int DR_BLOCKS = 1024;      // fixed grid: per-workgroup partial sums, added on the host in index order
constexpr int DR_MAXM = 6;           // members per launch (periodic Hessian: 6 distinct second derivatives)

struct MemberList { KParams kp[DR_MAXM]; int count; int kernel; };

template <typename T>
__device__ __forceinline__ T member_value(const KParams &kp, T r)
{
    // r: squared distance (gaussian) or signed difference (periodic, d == 1)
    if (kp.kernel == GPX_KERNEL_GAUSSIAN) {
        const T c1 = (T)kp.c[0], c2 = (T)kp.c[1], c3 = (T)kp.c[2], c4 = (T)kp.c[3];
        const int form = (int)kp.c[4];
        if (form == 0) return gaussian_entry<T, 0>(r, c1, c2, c3, c4);
        if (form == 1) return gaussian_entry<T, 1>(r, c1, c2, c3, c4);
        return gaussian_entry<T, 2>(r, c1, c2, c3, c4);
    }
    return periodic_entry<T>(kp.member, r, (T)kp.c[0], (T)kp.c[1], (T)kp.c[2]);
}

// For every member p of the list:  Q[p] = sum_ab alpha_a alpha_b mem_p(x_a, x_b),
//                                  H[p] = sum_ab W[a, b] mem_p(x_a, x_b)
// (all members are even in x_a - x_b and W is symmetric: the sums run over the lower triangle with
// weight 2 off the diagonal).  partial: DR_BLOCKS x (2 * DR_MAXM) doubles.
template <typename T>
__global__ __launch_bounds__(256) void member_quad_trace_kernel(const T *__restrict__ x, int64_t n, int d,
                                                                const T *__restrict__ alpha,
                                                                const T *__restrict__ W, int64_t ldw,
                                                                MemberList ml, int64_t ntiles,
                                                                double *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *s1 = reinterpret_cast<T *>(smem_raw);          // [DR_T][d]   row points
    T *s2 = s1 + (size_t)DR_T * d;                    // [d][DR_T]   column points, transposed
    T *sa1 = s2 + (size_t)DR_T * d;
    T *sa2 = sa1 + DR_T;
    __shared__ double red[4][2 * DR_MAXM];
    const int tid = threadIdx.x, col = tid & 63, rg = tid >> 6;
    double accq[DR_MAXM], acch[DR_MAXM];
#pragma unroll
    for (int p = 0; p < DR_MAXM; ++p) { accq[p] = 0.0; acch[p] = 0.0; }
    const int64_t total = ntiles * (ntiles + 1) / 2;
    for (int64_t t = blockIdx.x; t < total; t += gridDim.x) {
        int64_t tr = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((tr + 1) * (tr + 2) / 2 <= t) ++tr;
        while (tr * (tr + 1) / 2 > t) --tr;
        const int64_t tc = t - tr * (tr + 1) / 2;
        const int64_t r0 = tr * DR_T, c0 = tc * DR_T;
        __syncthreads();
        for (int idx = tid; idx < DR_T * d; idx += 256) {
            const int r = idx / d, k = idx - r * d;
            s1[idx] = (r0 + r < n) ? x[(r0 + r) * d + k] : (T)0;
            s2[(size_t)k * DR_T + r] = (c0 + r < n) ? x[(c0 + r) * d + k] : (T)0;
        }
        if (tid < DR_T) {
            sa1[tid] = (r0 + tid < n) ? alpha[r0 + tid] : (T)0;
            sa2[tid] = (c0 + tid < n) ? alpha[c0 + tid] : (T)0;
        }
        __syncthreads();
        const int64_t gc = c0 + col;
        const T ak = sa2[col];
        for (int i = 0; i < 16; ++i) {
            const int r = rg + 4 * i;
            const int64_t gr = r0 + r;
            if (gr >= n || gc >= n || gc > gr) continue;
            const double wt = (gc == gr) ? 1.0 : 2.0;
            const double wab = wt * (double)W[gr * ldw + gc];
            const double aab = wt * (double)sa1[r] * (double)ak;
            T rr;
            if (ml.kernel == GPX_KERNEL_GAUSSIAN) {
                rr = (T)0;
                for (int k = 0; k < d; ++k) {
                    const T tt = s1[r * d + k] - s2[(size_t)k * DR_T + col];
                    rr = fma(tt, tt, rr);
                }
            } else {
                rr = s1[r] - s2[col];                  // d == 1
            }
#pragma unroll
            for (int p = 0; p < DR_MAXM; ++p) {
                if (p < ml.count) {
                    const double v = (double)member_value<T>(ml.kp[p], rr);
                    accq[p] = fma(aab, v, accq[p]);
                    acch[p] = fma(wab, v, acch[p]);
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < DR_MAXM; ++p) {
        double vq = accq[p], vh = acch[p];
        for (int off = 32; off > 0; off >>= 1) { vq += __shfl_down(vq, off, 64); vh += __shfl_down(vh, off, 64); }
        if ((tid & 63) == 0) { red[tid >> 6][2 * p] = vq; red[tid >> 6][2 * p + 1] = vh; }
    }
    __syncthreads();
    if (tid < 2 * DR_MAXM)
        partial[(int64_t)blockIdx.x * (2 * DR_MAXM) + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

// sum_ab A[a, b] * B[b, a] = trace(A B) for two n x n matrices: 64 x 64 tiles, B's tile transposed
// through LDS so that both global reads are row-contiguous.  partial: DR_BLOCKS doubles.
template <typename T>
__global__ __launch_bounds__(256) void trace_prod_kernel(const T *__restrict__ A, int64_t lda,
                                                         const T *__restrict__ B, int64_t ldb, int64_t n,
                                                         int64_t ntiles, double *__restrict__ partial)
{
    __shared__ T sB[DR_T][DR_T + 1];
    __shared__ double red[4];
    const int tid = threadIdx.x, col = tid & 63, rg = tid >> 6;
    double acc = 0.0;
    const int64_t total = ntiles * ntiles;
    for (int64_t t = blockIdx.x; t < total; t += gridDim.x) {
        const int64_t ta = t / ntiles, tb = t - ta * ntiles;
        const int64_t a0 = ta * DR_T, b0 = tb * DR_T;
        __syncthreads();
        for (int i = 0; i < 16; ++i) {                     // B tile rows b0.., columns a0..
            const int r = rg + 4 * i;
            sB[r][col] = (b0 + r < n && a0 + col < n) ? B[(b0 + r) * ldb + a0 + col] : (T)0;
        }
        __syncthreads();
        for (int i = 0; i < 16; ++i) {                     // A tile rows a0.., columns b0..
            const int r = rg + 4 * i;
            if (a0 + r < n && b0 + col < n)
                acc = fma((double)A[(a0 + r) * lda + b0 + col], (double)sB[col][r], acc);
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// out[0] = sum_i a[i * stride]  (trace of a matrix: stride = ld + 1); single workgroup, fixed order
template <typename T>
__global__ __launch_bounds__(1024) void strided_sum_kernel(const T *__restrict__ a, int64_t n, int64_t stride,
                                                           double *__restrict__ out)
{
    __shared__ double red[16];
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (int64_t i = tid; i < n; i += 1024) acc += (double)a[i * stride];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += red[w];
        out[0] = s;
    }
}

template <typename T>
__global__ void scale_copy_kernel(const T *__restrict__ src, T *__restrict__ dst, int64_t n, double f)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (T)((double)src[i] * f);
}

template <typename T>
__global__ void eye_fill_kernel(T *__restrict__ X, int64_t n, int64_t ld)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ld) return;
    for (int64_t r = blockIdx.y; r < n; r += gridDim.y) X[r * ld + c] = (c == r) ? (T)1 : (T)0;
}

// out (m) <- a (m) - b (m)
template <typename T>
__global__ void sub_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out, int64_t n, double fb)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (T)((a ? (double)a[i] : 0.0) - fb * (double)b[i]);
}

static const int kJacGaussian[2] = {GPX_DK_DH, GPX_DK_DW};
static const int kJacPeriodic[3] = {GPX_DK_DH, GPX_DK_DW, GPX_DK_DP};
static int hess_member(int kernel, int i, int j)
{
    if (i > j) std::swap(i, j);
    if (kernel == GPX_KERNEL_GAUSSIAN) {
        static const int t[2][2] = {{GPX_D2K_DHDH, GPX_D2K_DHDW}, {GPX_D2K_DHDW, GPX_D2K_DWDW}};
        return t[i][j];
    }
    static const int t[3][3] = {{GPX_D2K_DHDH, GPX_D2K_DHDW, GPX_D2K_DHDP},
                                {GPX_D2K_DHDW, GPX_D2K_DWDW, GPX_D2K_DWDP},
                                {GPX_D2K_DHDP, GPX_D2K_DWDP, GPX_D2K_DPDP}};
    return t[i][j];
}

static double host_sum(hipStream_t st, const double *dev, int64_t count, int64_t stride, int64_t offset, int *rc)
{
    std::vector<double> h((size_t)count * stride);
    hipError_t e = hipMemcpyAsync(h.data(), dev, h.size() * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { *rc = hip_fail(e, "host_sum", __FILE__, __LINE__); return 0.0; }
    double s = 0.0;
    for (int64_t b = 0; b < count; ++b) s += h[(size_t)b * stride + offset];
    return s;
}

// state shared by the three entry points: W = K^-1 (optional), v_i, u_i
struct DerivState {
    gpx_gp *g;
    int np;                       // kernel parameters; index np is the noise s
    int64_t n, lda;
    size_t es;
    DevBuf W, V, U, tmp, scal, part;
    void *v(int i) { return (char *)V.p + (size_t)i * n * es; }
    void *u(int i) { return (char *)U.p + (size_t)i * n * es; }
};

static const int *jac_members(int kernel) { return kernel == GPX_KERNEL_GAUSSIAN ? kJacGaussian : kJacPeriodic; }

// v_i = dK_i alpha (i < np: fused on-the-fly mat-vec; i = np: 2 s alpha), u_i = K^-1 v_i
static int deriv_vectors(DerivState &D)
{
    gpx_gp *g = D.g;
    const int np = D.np;
    const int64_t n = D.n;
    GPX_TRY(D.V.alloc((size_t)(np + 1) * n * D.es));
    GPX_TRY(D.U.alloc((size_t)(np + 1) * n * D.es));
    GPX_TRY(D.tmp.alloc((size_t)2 * n * D.es));
    const int *jm = jac_members(g->kernel);
    for (int i = 0; i < np; ++i)
        GPX_TRY(gpx_d_mean_member(g->dtype, g->kernel, jm[i], g->x, n, g->x, n, g->d, g->params, g->alpha, D.v(i),
                                  (void *)g->st));
    const unsigned nblk = (unsigned)cdiv(n, 256);
    if (g->dtype == GPX_F64)
        hipLaunchKernelGGL((scale_copy_kernel<double>), dim3(nblk), dim3(256), 0, g->st, (const double *)g->alpha,
                           (double *)D.v(np), n, 2.0 * g->s);
    else
        hipLaunchKernelGGL((scale_copy_kernel<float>), dim3(nblk), dim3(256), 0, g->st, (const float *)g->alpha,
                           (float *)D.v(np), n, 2.0 * g->s);
    GPX_LAUNCH_CHECK();
    for (int i = 0; i <= np; ++i) {
        void *t0 = D.tmp.p, *t1 = (char *)D.tmp.p + (size_t)n * D.es;
        GPX_HIP(hipMemcpyAsync(t0, D.v(i), (size_t)n * D.es, hipMemcpyDeviceToDevice, g->st));
        GPX_TRY(trsv_lower(g->dtype, g->A, n, g->lda, t0, t1, 0, g->st, nullptr, &g->ops));
        GPX_TRY(trsv_lower(g->dtype, g->A, n, g->lda, t1, D.u(i), 1, g->st, nullptr, &g->ops));
    }
    return GPX_OK;
}

static int read_info(gpx_gp *g, int *info, double *logdet, double *yta)
{
    double h4[4];
    GPX_HIP(hipMemcpyAsync(h4, g->scal, sizeof(h4), hipMemcpyDeviceToHost, g->st));
    GPX_HIP(hipStreamSynchronize(g->st));
    memcpy(info, &h4[3], sizeof(int));
    GPX_TRY(check_internal_info(*info));
    if (!g->y_finite) { set_error("array must not contain infs or NaNs (y)"); return GPX_ERR_ARG; }   // gp/gp.py:332-334
    *logdet = h4[0]; *yta = h4[1];
    return GPX_OK;
}

// lh exactly as gp/gp.py:392-396 on top of gp_c.log_lh's clamps
static double lh_of(gpx_gp *g, double logdet, double yta)
{
    if (!(logdet >= GPX_MIN_LOG)) return 0.0;
    const double llh = -0.5 * yta - 0.5 * logdet - 0.5 * (double)g->n * log(2 * M_PI);
    return llh < GPX_MIN_LOG ? 0.0 : exp(llh);
}

template <typename T>
static int d2lh_t(gpx_gp *g, double *dlh_out, double *d2lh_out, double *d2loglh_out)
{
    const int np = g->nparams, P = np + 1;
    const int64_t n = g->n, lda = g->lda;
    int info; double logdet, yta;
    GPX_TRY(read_info(g, &info, &logdet, &yta));
    if (info != 0) {                                      // gp/gp.py:424-428 and :458-462, :493-497: NaN when not PD
        if (dlh_out) for (int i = 0; i < P; ++i) dlh_out[i] = NAN;
        if (d2lh_out) for (int i = 0; i < P * P; ++i) d2lh_out[i] = NAN;
        if (d2loglh_out) for (int i = 0; i < P * P; ++i) d2loglh_out[i] = NAN;
        return GPX_OK;
    }
    const double lh = lh_of(g, logdet, yta);
    DerivState D;
    D.g = g; D.np = np; D.n = n; D.lda = lda; D.es = sizeof(T);
    hipStream_t st = g->st;
    GPX_TRY(deriv_vectors(D));
    // W = K^-1, full: X = L^-T (upper triangular), W = X X^T
    DevBuf X;
    GPX_TRY(D.W.alloc((size_t)n * lda * sizeof(T)));
    GPX_TRY(X.alloc((size_t)n * lda * sizeof(T)));
    GPX_TRY(D.scal.alloc(64 * sizeof(double)));
    GPX_TRY(D.part.alloc((size_t)DR_BLOCKS * 2 * DR_MAXM * sizeof(double)));
    {
        dim3 grid((unsigned)cdiv(lda, 256), (unsigned)std::min<int64_t>(n, 32768)), block(256);
        hipLaunchKernelGGL((eye_fill_kernel<T>), grid, block, 0, st, (T *)X.p, n, lda);
        GPX_LAUNCH_CHECK();
        GPX_HIP(hipMemsetAsync(D.W.p, 0, (size_t)n * lda * sizeof(T), st));
        GPX_TRY(trsm_right_lt(g->dtype, g->A, n, lda, X.p, n, lda, st, 1, &g->ops));
        GPX_TRY(gemm_nt(g->dtype, n, n, n, X.p, lda, X.p, lda, D.W.p, lda, 1.0, GPX_FULL, 0, 0, st, 0, 1));
    }
    // a_i = alpha . v_i ; vu_ij = v_j . u_i   (device dots, one launch each: count = 1)
    std::vector<double> a(P), tr(P), vu((size_t)P * P), q((size_t)P * P, 0.0), hh((size_t)P * P, 0.0), T2((size_t)P * P, 0.0);
    double *sc = (double *)D.scal.p;
    int slot = 0;
    for (int i = 0; i < P; ++i) GPX_TRY(dot(g->dtype, g->alpha, D.v(i), n, sc + slot++, st));
    for (int i = 0; i < P; ++i)
        for (int j = 0; j < P; ++j) GPX_TRY(dot(g->dtype, D.v(j), D.u(i), n, sc + slot++, st));
    hipLaunchKernelGGL((strided_sum_kernel<T>), dim3(1), dim3(1024), 0, st, (const T *)D.W.p, n, lda + 1, sc + slot);
    const int slot_trW = slot++;
    GPX_LAUNCH_CHECK();
    // M_i = W dK_i (i < np), trace(M_i); D_i is built into X (free now), M_i kept
    std::vector<DevBuf> M(np);
    const int *jm = jac_members(g->kernel);
    const int slot_trM = slot;
    for (int i = 0; i < np; ++i) {
        GPX_TRY(M[i].alloc((size_t)n * lda * sizeof(T)));
        GPX_TRY(kmat(g->dtype, g->kernel, jm[i], g->x, n, g->x, n, g->d, g->params, 0.0, GPX_FULL, X.p, lda, st));
        GPX_HIP(hipMemsetAsync(M[i].p, 0, (size_t)n * lda * sizeof(T), st));
        GPX_TRY(gemm_nt(g->dtype, n, n, n, D.W.p, lda, X.p, lda, M[i].p, lda, 1.0, GPX_FULL, 0, 0, st));   // W dK_i^T = W dK_i
        hipLaunchKernelGGL((strided_sum_kernel<T>), dim3(1), dim3(1024), 0, st, (const T *)M[i].p, n, lda + 1, sc + slot++);
        GPX_LAUNCH_CHECK();
    }
    std::vector<double> hs(slot);
    GPX_HIP(hipMemcpyAsync(hs.data(), sc, (size_t)slot * sizeof(double), hipMemcpyDeviceToHost, st));
    GPX_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < P; ++i) a[i] = hs[i];
    for (int i = 0; i < P; ++i) for (int j = 0; j < P; ++j) vu[(size_t)i * P + j] = hs[P + i * P + j];
    const double trW = hs[slot_trW];
    for (int i = 0; i < np; ++i) tr[i] = hs[slot_trM + i];
    tr[np] = 2.0 * g->s * trW;
    // T_ij = trace(M_j M_i): kernel x kernel by the transposed-tile reduction; with the noise: M_s = 2 s W
    const int64_t ntl = cdiv(n, DR_T);
    const int tblocks = (int)std::min<int64_t>(DR_BLOCKS, ntl * ntl);
    int rc = GPX_OK;
    auto trace_prod = [&](const void *Ap, const void *Bp) -> double {
        hipLaunchKernelGGL((trace_prod_kernel<T>), dim3(tblocks), dim3(256), 0, st, (const T *)Ap, lda, (const T *)Bp, lda,
                           n, ntl, (double *)D.part.p);
        return host_sum(st, (const double *)D.part.p, tblocks, 1, 0, &rc);
    };
    for (int i = 0; i < np; ++i)
        for (int j = i; j < np; ++j) {
            const double v = trace_prod(M[j].p, M[i].p);
            T2[(size_t)i * P + j] = T2[(size_t)j * P + i] = v;
        }
    for (int i = 0; i < np; ++i) {
        const double v = 2.0 * g->s * trace_prod(D.W.p, M[i].p);
        T2[(size_t)i * P + np] = T2[(size_t)np * P + i] = v;
    }
    T2[(size_t)np * P + np] = 4.0 * g->s * g->s * trace_prod(D.W.p, D.W.p);
    if (rc != GPX_OK) return rc;
    // q_ij = alpha^T d2K_ij alpha, h_ij = trace(W d2K_ij): second derivatives evaluated on the fly
    {
        MemberList ml;
        memset(&ml, 0, sizeof(ml));
        ml.kernel = g->kernel;
        std::vector<std::pair<int, int>> idx;
        for (int i = 0; i < np; ++i)
            for (int j = i; j < np; ++j) {
                GPX_TRY(make_kparams(g->kernel, hess_member(g->kernel, i, j), g->params, 0.0, &ml.kp[ml.count]));
                ++ml.count;
                idx.push_back({i, j});
            }
        const int64_t ntr = ntl;
        const int blocks = (int)std::min<int64_t>(DR_BLOCKS, ntr * (ntr + 1) / 2);
        const size_t smem = ((size_t)2 * DR_T * g->d + 2 * DR_T) * sizeof(T);
        if (smem > 48 * 1024)
            GPX_HIP(hipFuncSetAttribute((const void *)member_quad_trace_kernel<T>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL((member_quad_trace_kernel<T>), dim3(blocks), dim3(256), smem, st, (const T *)g->x, n, g->d,
                           (const T *)g->alpha, (const T *)D.W.p, lda, ml, ntr, (double *)D.part.p);
        GPX_LAUNCH_CHECK();
        for (int p = 0; p < ml.count; ++p) {
            const double qv = host_sum(st, (const double *)D.part.p, blocks, 2 * DR_MAXM, 2 * p, &rc);
            const double hv = host_sum(st, (const double *)D.part.p, blocks, 2 * DR_MAXM, 2 * p + 1, &rc);
            const int i = idx[p].first, j = idx[p].second;
            q[(size_t)i * P + j] = q[(size_t)j * P + i] = qv;
            hh[(size_t)i * P + j] = hh[(size_t)j * P + i] = hv;
        }
        if (rc != GPX_OK) return rc;
        // noise x noise: d2K = 2 I (gp_c.pyx:93-94); mixed kernel x noise: 0
        GPX_TRY(dot(g->dtype, g->alpha, g->alpha, n, sc, st));
        double ata = 0.0;
        GPX_HIP(hipMemcpyAsync(&ata, sc, sizeof(double), hipMemcpyDeviceToHost, st));
        GPX_HIP(hipStreamSynchronize(st));
        q[(size_t)np * P + np] = 2.0 * ata;
        hh[(size_t)np * P + np] = 2.0 * trW;
    }
    std::vector<double> dlh(P);
    for (int i = 0; i < P; ++i) dlh[i] = 0.5 * lh * (a[i] - tr[i]);          // gp_c.pyx:62-66
    if (dlh_out) for (int i = 0; i < P; ++i) dlh_out[i] = dlh[i];
    if (d2lh_out)
        for (int i = 0; i < P; ++i)
            for (int j = 0; j < P; ++j) {
                const size_t ij = (size_t)i * P + j;
                const double t0 = dlh[j] * (a[i] - tr[i]);                         // gp_c.pyx:104
                const double t1 = lh * (-2.0 * vu[ij] + q[ij] + T2[ij] - hh[ij]);  // gp_c.pyx:105-109
                d2lh_out[ij] = 0.5 * (t0 + t1);
            }
    // Hessian of the LOG marginal likelihood: d2lh / lh - (dlh / lh)(dlh / lh)^T = the bracket above
    // without its lh factors -- finite at any n, where lh itself underflows to 0 (log_lh < MIN)
    if (d2loglh_out)
        for (int ij = 0; ij < P * P; ++ij) d2loglh_out[ij] = 0.5 * (-2.0 * vu[ij] + q[ij] + T2[ij] - hh[ij]);
    return GPX_OK;
}

template <typename T>
static int dm_t(gpx_gp *g, const double *xo, int64_t m, double *out)
{
    const int np = g->nparams, P = np + 1;
    const int64_t n = g->n;
    int info; double logdet, yta;
    GPX_TRY(read_info(g, &info, &logdet, &yta));
    if (info != 0) { set_error("dm_dtheta: the kernel matrix is not positive definite"); return GPX_ERR_ARG; }
    DerivState D;
    D.g = g; D.np = np; D.n = n; D.lda = g->lda; D.es = sizeof(T);
    hipStream_t st = g->st;
    GPX_TRY(deriv_vectors(D));
    DevBuf dxo, o1, o2, res, host64;
    GPX_TRY(dxo.alloc((size_t)m * g->d * sizeof(T)));
    GPX_TRY(o1.alloc((size_t)m * sizeof(T)));
    GPX_TRY(o2.alloc((size_t)m * sizeof(T)));
    GPX_TRY(res.alloc((size_t)P * m * sizeof(T)));
    {
        // upload the test points (host f64 -> dtype)
        std::vector<T> h((size_t)m * g->d);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (T)xo[i];
        GPX_HIP(hipMemcpyAsync(dxo.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st));
        GPX_HIP(hipStreamSynchronize(st));
    }
    const int *jm = jac_members(g->kernel);
    const unsigned mb = (unsigned)cdiv(m, 256);
    for (int i = 0; i < P; ++i) {
        // dm_i = dK_i(xo, x) alpha - K(xo, x) u_i     (gp_c.pyx:121-131; dK_s(xo, x) = 0)
        if (i < np)
            GPX_TRY(gpx_d_mean_member(g->dtype, g->kernel, jm[i], dxo.p, m, g->x, n, g->d, g->params, g->alpha, o1.p,
                                      (void *)st));
        GPX_TRY(gpx_d_mean_member(g->dtype, g->kernel, GPX_K, dxo.p, m, g->x, n, g->d, g->params, D.u(i), o2.p,
                                  (void *)st));
        hipLaunchKernelGGL((sub_kernel<T>), dim3(mb), dim3(256), 0, st, i < np ? (const T *)o1.p : (const T *)nullptr,
                           (const T *)o2.p, (T *)res.p + (size_t)i * m, m, 1.0);
        GPX_LAUNCH_CHECK();
    }
    std::vector<T> h((size_t)P * m);
    GPX_HIP(hipMemcpyAsync(h.data(), res.p, h.size() * sizeof(T), hipMemcpyDeviceToHost, st));
    GPX_HIP(hipStreamSynchronize(st));
    for (size_t i = 0; i < h.size(); ++i) out[i] = (double)h[i];
    return GPX_OK;
}


// =====================================================================================================================
// The dense-matrix glue of gp/ext/gp_c.pyx:34-131 for PLUGIN kernels (any Kernel subclass without a native id): the
// caller holds K^-1, the Jacobian and the Hessian of its own kernel as HOST float64 matrices, exactly the arguments of
// the reference's four functions.  The reference evaluates them as O(n_p^2) dense n x n x n products; here every
// matrix crosses PCIe ONCE, every term that is a quadratic form becomes matrix-vector work (O(n^2)), every trace of a
// product becomes a transposed-tile reduction (O(n^2), no product formed), and only the n_p products N_i = dK_i K^-1
// that the second-derivative trace terms need run as n^3 GEMMs on the MFMA kernel.  With
//     z = Ki^T y      w = Ki y       v_i = dK_i Kiy      u_i = Ki v_i      g_i = dK_i^T z      k_i = dK_i^T Kiy
//     p_i = dK_i w    r_i = Ki p_i   N_i = dK_i Ki       (noise, i = n_p: dK = 2 s I, so each is 2 s times a known vector)
// nothing assumes Ki or dK_i symmetric, and term by term (names of gp_c.pyx in brackets)
//     [y . (Ki dK_i) Kiy]          = z . v_i                      [trace(Ki dK_i)]        = sum_ab Ki[a,b] dK_i[b,a]
//     [t1a = y . dKi_j dK_i Kiy]   = -g_j . u_i                   [t1b = Kiy . d2k Kiy]   = Kiy . (d2K_ij Kiy)
//     [t1c = Kiy . dK_i dKi_j y]   = -k_i . r_j                   [trace(dKi_j dK_i)]     = -sum_ab N_j[a,b] N_i[b,a]
//     [trace(Ki d2k)]              = sum_ab Ki[a,b] d2K_ij[b,a]   [dm_i]                  = dKxox_i w - Kxox r_i
// Only the (n_p + 1) / (n_p + 1)^2 scalars, or the (n_p + 1) x m matrix of dm, return to the host.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T *__restrict__ src, int64_t lds, T *__restrict__ dst,
                                                        int64_t ldd, int64_t n)
{
    __shared__ T tile[DR_T][DR_T + 1];
    const int64_t r0 = (int64_t)blockIdx.y * DR_T, c0 = (int64_t)blockIdx.x * DR_T;
    const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
    for (int i = 0; i < 16; ++i) {
        const int r = rg + 4 * i;
        tile[r][col] = (r0 + r < n && c0 + col < n) ? src[(r0 + r) * lds + c0 + col] : (T)0;
    }
    __syncthreads();
    for (int i = 0; i < 16; ++i) {
        const int r = rg + 4 * i;
        if (c0 + r < n && r0 + col < n) dst[(c0 + r) * ldd + r0 + col] = tile[col][r];
    }
}

// partial[chunk][c] = sum over the chunk's rows r of M[r][c] v[r]   (M^T v in two passes, fixed summation order)
template <typename T>
__global__ __launch_bounds__(256) void gemv_t_partial_kernel(const T *__restrict__ M, int64_t ld, int64_t rows, int64_t cols,
                                                             const T *__restrict__ v, double *__restrict__ partial,
                                                             int64_t rows_per_chunk)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int64_t rb = (int64_t)blockIdx.y * rows_per_chunk, re = rb + rows_per_chunk < rows ? rb + rows_per_chunk : rows;
    double acc = 0.0;
    for (int64_t r = rb; r < re; ++r) acc = fma((double)M[r * ld + c], (double)v[r], acc);
    partial[(int64_t)blockIdx.y * cols + c] = acc;
}
template <typename T>
__global__ __launch_bounds__(256) void gemv_t_sum_kernel(const double *__restrict__ partial, int64_t cols, int nchunks,
                                                         T *__restrict__ out)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    double acc = 0.0;
    for (int k = 0; k < nchunks; ++k) acc += partial[(int64_t)k * cols + c];
    out[c] = (T)acc;
}

// fp64 only: the reference's boundary is float64 host arrays (gp_c.pyx signatures)
struct Glue {
    int64_t n = 0, ld = 0;
    hipStream_t st = nullptr;
    DevBuf part, sc, gpart;
    std::vector<double> hsc;          // host copy of the dot products, filled by flush()
    int nsc = 0;
    static constexpr int SC_MAX = 4096;

    int init(int64_t n_)
    {
        n = n_; ld = round_up(n_, 16);
        GPX_TRY(part.alloc((size_t)DR_BLOCKS * sizeof(double)));
        GPX_TRY(sc.alloc((size_t)SC_MAX * sizeof(double)));
        GPX_TRY(gpart.alloc((size_t)64 * ld * sizeof(double)));
        return GPX_OK;
    }
    // host (rows x cols, dense) -> device (rows x ldd)
    int upload(DevBuf &b, const double *h, int64_t rows, int64_t cols, int64_t ldd)
    {
        if (!b.p) GPX_TRY(b.alloc((size_t)rows * ldd * 8));
        GPX_HIP(hipMemcpy2DAsync(b.p, (size_t)ldd * 8, h, (size_t)cols * 8, (size_t)cols * 8, (size_t)rows, hipMemcpyHostToDevice, st));
        GPX_HIP(hipStreamSynchronize(st));           // (the host buffer is the caller's: done with it on return)
        return GPX_OK;
    }
    int vec(DevBuf &b, int64_t len) { return b.alloc((size_t)round_up(len, 16) * 8); }
    // out (rows) = M (rows x cols, ldm) v (cols)
    int mv(const void *M, int64_t rows, int64_t cols, int64_t ldm, const void *v, void *out)
    {
        GPX_HIP(hipMemsetAsync(out, 0, (size_t)rows * 8, st));
        return gemm_nt(GPX_F64, rows, 1, cols, M, ldm, v, round_up(cols, 16), out, 1, 1.0, GPX_FULL, 0, 0, st);
    }
    // out (cols) = M^T v, M rows x cols
    int mtv(const void *M, int64_t rows, int64_t cols, int64_t ldm, const void *v, void *out)
    {
        const int nch = (int)std::min<int64_t>(64, cdiv(rows, 64));
        const int64_t rpc = cdiv(rows, nch);
        dim3 grid((unsigned)cdiv(cols, 256), (unsigned)nch);
        hipLaunchKernelGGL((gemv_t_partial_kernel<double>), grid, dim3(256), 0, st, (const double *)M, ldm, rows, cols,
                           (const double *)v, (double *)gpart.p, rpc);
        hipLaunchKernelGGL((gemv_t_sum_kernel<double>), dim3(grid.x), dim3(256), 0, st, (const double *)gpart.p, cols, nch, (double *)out);
        GPX_LAUNCH_CHECK();
        return GPX_OK;
    }
    int scale(const void *src, void *dst, int64_t len, double f)
    {
        hipLaunchKernelGGL((scale_copy_kernel<double>), dim3((unsigned)cdiv(len, 256)), dim3(256), 0, st, (const double *)src,
                           (double *)dst, len, f);
        GPX_LAUNCH_CHECK();
        return GPX_OK;
    }
    // queue a . b; returns its slot (value in hsc after flush())
    int qdot(const void *a, const void *b, int64_t len, int *slot)
    {
        if (nsc >= SC_MAX) { set_error("gp_c glue: too many parameters"); return GPX_ERR_ARG; }
        *slot = nsc++;
        return dot(GPX_F64, a, b, len, (double *)sc.p + *slot, st);
    }
    int flush()
    {
        hsc.resize((size_t)std::max(nsc, 1));
        GPX_HIP(hipMemcpyAsync(hsc.data(), sc.p, (size_t)std::max(nsc, 1) * 8, hipMemcpyDeviceToHost, st));
        GPX_HIP(hipStreamSynchronize(st));
        return GPX_OK;
    }
    // sum_ab A[a, b] B[b, a] for n x n device matrices (ld)
    int trace_prod(const void *A, const void *B, double *out)
    {
        const int64_t ntl = cdiv(n, DR_T);
        const int blocks = (int)std::min<int64_t>(DR_BLOCKS, ntl * ntl);
        hipLaunchKernelGGL((trace_prod_kernel<double>), dim3(blocks), dim3(256), 0, st, (const double *)A, ld, (const double *)B, ld,
                           n, ntl, (double *)part.p);
        GPX_LAUNCH_CHECK();
        int rc = GPX_OK;
        *out = host_sum(st, (const double *)part.p, blocks, 1, 0, &rc);
        return rc;
    }
    int trace(const void *A, double *out)
    {
        hipLaunchKernelGGL((strided_sum_kernel<double>), dim3(1), dim3(1024), 0, st, (const double *)A, n, ld + 1, (double *)part.p);
        GPX_LAUNCH_CHECK();
        GPX_HIP(hipMemcpyAsync(out, part.p, 8, hipMemcpyDeviceToHost, st));
        GPX_HIP(hipStreamSynchronize(st));
        return GPX_OK;
    }
};

// a_i = y . (Ki dK_i) Kiy and tr_i = trace(Ki dK_i), i = 0 .. np (np: the noise) -- gp_c.pyx:42-48 and :60-66
static int glue_first_order(const double *y, const double *Ki, const double *Kj, const double *Kiy, double s, int64_t n, int np,
                            double *a, double *tr)
{
    Glue G;
    GPX_TRY(G.init(n));
    DevBuf dKi, dD, dy, dKiy, dz, dv;
    GPX_TRY(G.upload(dKi, Ki, n, n, G.ld));
    GPX_TRY(G.vec(dy, n)); GPX_TRY(G.vec(dKiy, n)); GPX_TRY(G.vec(dz, n)); GPX_TRY(G.vec(dv, n));
    GPX_HIP(hipMemcpyAsync(dy.p, y, (size_t)n * 8, hipMemcpyHostToDevice, G.st));
    GPX_HIP(hipMemcpyAsync(dKiy.p, Kiy, (size_t)n * 8, hipMemcpyHostToDevice, G.st));
    GPX_TRY(G.mtv(dKi.p, n, n, G.ld, dy.p, dz.p));                       // z = Ki^T y
    std::vector<int> slot(np + 1);
    for (int i = 0; i < np; ++i) {
        GPX_TRY(G.upload(dD, Kj + (size_t)i * n * n, n, n, G.ld));
        GPX_TRY(G.mv(dD.p, n, n, G.ld, dKiy.p, dv.p));                  // v_i = dK_i Kiy
        GPX_TRY(G.qdot(dz.p, dv.p, n, &slot[i]));
        GPX_TRY(G.trace_prod(dKi.p, dD.p, &tr[i]));                     // (synchronises: dD may be overwritten next)
    }
    GPX_TRY(G.qdot(dz.p, dKiy.p, n, &slot[np]));
    double trKi = 0.0;
    GPX_TRY(G.trace(dKi.p, &trKi));
    GPX_TRY(G.flush());
    for (int i = 0; i < np; ++i) a[i] = G.hsc[slot[i]];
    a[np] = 2.0 * s * G.hsc[slot[np]];
    tr[np] = 2.0 * s * trKi;
    return GPX_OK;
}

static int glue_d2lh(const double *y, const double *Ki, const double *Kj, const double *Kh, const double *Kiy, double s, double lh,
                     const double *dlh, int64_t n, int np, double *d2lh)
{
    const int P = np + 1;
    Glue G;
    GPX_TRY(G.init(n));
    const int64_t ld = G.ld, vs = round_up(n, 16);                      // vs: stride between the vectors of a family
    DevBuf dKi, dKiT, dH, dy, dKiy, dz, dw, dt;
    std::vector<DevBuf> dD(np), dN(np);
    DevBuf V, U, Gv, Kv, Pv, Rv;                                          // v_i, u_i, g_i, k_i, p_i, r_i: P vectors each
    for (DevBuf *b : {&V, &U, &Gv, &Kv, &Pv, &Rv}) GPX_TRY(b->alloc((size_t)P * vs * 8));
    auto at = [&](DevBuf &b, int i) { return (void *)((double *)b.p + (size_t)i * vs); };
    GPX_TRY(G.upload(dKi, Ki, n, n, ld));
    GPX_TRY(dKiT.alloc((size_t)n * ld * 8));
    {
        dim3 grid((unsigned)cdiv(n, DR_T), (unsigned)cdiv(n, DR_T));
        hipLaunchKernelGGL((transpose_kernel<double>), grid, dim3(256), 0, G.st, (const double *)dKi.p, ld, (double *)dKiT.p, ld, n);
        GPX_LAUNCH_CHECK();
    }
    GPX_TRY(G.vec(dy, n)); GPX_TRY(G.vec(dKiy, n)); GPX_TRY(G.vec(dz, n)); GPX_TRY(G.vec(dw, n)); GPX_TRY(G.vec(dt, n));
    GPX_HIP(hipMemcpyAsync(dy.p, y, (size_t)n * 8, hipMemcpyHostToDevice, G.st));
    GPX_HIP(hipMemcpyAsync(dKiy.p, Kiy, (size_t)n * 8, hipMemcpyHostToDevice, G.st));
    GPX_TRY(G.mv(dKiT.p, n, n, ld, dy.p, dz.p));                         // z = Ki^T y
    GPX_TRY(G.mv(dKi.p, n, n, ld, dy.p, dw.p));                          // w = Ki y
    for (int i = 0; i < np; ++i) {
        GPX_TRY(G.upload(dD[i], Kj + (size_t)i * n * n, n, n, ld));
        GPX_TRY(G.mv(dD[i].p, n, n, ld, dKiy.p, at(V, i)));
        GPX_TRY(G.mv(dKi.p, n, n, ld, at(V, i), at(U, i)));
        GPX_TRY(G.mtv(dD[i].p, n, n, ld, dz.p, at(Gv, i)));
        GPX_TRY(G.mtv(dD[i].p, n, n, ld, dKiy.p, at(Kv, i)));
        GPX_TRY(G.mv(dD[i].p, n, n, ld, dw.p, at(Pv, i)));
        GPX_TRY(G.mv(dKi.p, n, n, ld, at(Pv, i), at(Rv, i)));
        GPX_TRY(dN[i].alloc((size_t)n * ld * 8));
        GPX_HIP(hipMemsetAsync(dN[i].p, 0, (size_t)n * ld * 8, G.st));                   // (beta = 0 exists on the aligned fast path only)
        GPX_TRY(gemm_nt(GPX_F64, n, n, n, dD[i].p, ld, dKiT.p, ld, dN[i].p, ld, 1.0, GPX_FULL, 0, 0, G.st));      // N_i = dK_i Ki
    }
    // the noise: dK = 2 s I
    GPX_TRY(G.scale(dKiy.p, at(V, np), n, 2.0 * s));
    GPX_TRY(G.mv(dKi.p, n, n, ld, at(V, np), at(U, np)));
    GPX_TRY(G.scale(dz.p, at(Gv, np), n, 2.0 * s));
    GPX_TRY(G.scale(dKiy.p, at(Kv, np), n, 2.0 * s));
    GPX_TRY(G.scale(dw.p, at(Pv, np), n, 2.0 * s));
    GPX_TRY(G.mv(dKi.p, n, n, ld, at(Pv, np), at(Rv, np)));
    // scalars: a_i, t1a_ij, t1c_ij
    std::vector<int> sa(P), s1a((size_t)P * P), s1c((size_t)P * P), s1b((size_t)P * P, -1);
    for (int i = 0; i < P; ++i) GPX_TRY(G.qdot(dz.p, at(V, i), n, &sa[i]));
    for (int i = 0; i < P; ++i)
        for (int j = 0; j < P; ++j) {
            GPX_TRY(G.qdot(at(Gv, j), at(U, i), n, &s1a[(size_t)i * P + j]));
            GPX_TRY(G.qdot(at(Kv, i), at(Rv, j), n, &s1c[(size_t)i * P + j]));
        }
    int s_kk = 0;
    GPX_TRY(G.qdot(dKiy.p, dKiy.p, n, &s_kk));
    // traces
    std::vector<double> tr(P), T2((size_t)P * P, 0.0), hh((size_t)P * P, 0.0);
    double trKi = 0.0, trKiKi = 0.0;
    GPX_TRY(G.trace(dKi.p, &trKi));
    for (int i = 0; i < np; ++i) GPX_TRY(G.trace_prod(dKi.p, dD[i].p, &tr[i]));
    tr[np] = 2.0 * s * trKi;
    for (int i = 0; i < np; ++i)
        for (int j = 0; j < np; ++j) {
            if (j < i) { T2[(size_t)i * P + j] = T2[(size_t)j * P + i]; continue; }     // trace(N_j N_i) = trace(N_i N_j)
            GPX_TRY(G.trace_prod(dN[j].p, dN[i].p, &T2[(size_t)i * P + j]));
        }
    for (int i = 0; i < np; ++i) {
        double v = 0.0;
        GPX_TRY(G.trace_prod(dKi.p, dN[i].p, &v));                        // N_noise = 2 s Ki
        T2[(size_t)i * P + np] = T2[(size_t)np * P + i] = 2.0 * s * v;
    }
    GPX_TRY(G.trace_prod(dKi.p, dKi.p, &trKiKi));
    T2[(size_t)np * P + np] = 4.0 * s * s * trKiKi;
    // second kernel derivatives: each (m, m) block of Kh crosses once; t1b and trace(Ki d2K)
    for (int i = 0; i < np; ++i)
        for (int j = 0; j < np; ++j) {
            GPX_TRY(G.upload(dH, Kh + ((size_t)i * np + j) * n * n, n, n, ld));
            GPX_TRY(G.mv(dH.p, n, n, ld, dKiy.p, dt.p));
            GPX_TRY(G.qdot(dKiy.p, dt.p, n, &s1b[(size_t)i * P + j]));
            GPX_TRY(G.trace_prod(dKi.p, dH.p, &hh[(size_t)i * P + j]));   // (synchronises before dH / dt are reused)
        }
    hh[(size_t)np * P + np] = 2.0 * trKi;                                  // d2k = 2 I (gp_c.pyx:93-94)
    GPX_TRY(G.flush());
    for (int i = 0; i < P; ++i) {
        const double a_i = G.hsc[sa[i]];                                   // (the noise's factor 2 s is inside v_np already)
        for (int j = 0; j < P; ++j) {
            const size_t ij = (size_t)i * P + j;
            const double t0 = dlh[j] * (a_i - tr[i]);                                       // gp_c.pyx:100
            const double t1a = -G.hsc[s1a[ij]];
            const double t1b = (i < np && j < np) ? G.hsc[s1b[ij]] : ((i == np && j == np) ? 2.0 * G.hsc[s_kk] : 0.0);
            const double t1c = -G.hsc[s1c[ij]];
            const double t1 = lh * (t1a + t1b + t1c + T2[ij] - hh[ij]);                      // gp_c.pyx:102-105
            d2lh[ij] = 0.5 * (t0 + t1);
        }
    }
    return GPX_OK;
}

static int glue_dm(const double *y, const double *Ki, const double *Kj, const double *Kjxo, const double *Kxox, double s, int64_t n,
                   int np, int64_t m, double *dm)
{
    const int P = np + 1;
    Glue G;
    GPX_TRY(G.init(n));
    const int64_t ld = G.ld;
    DevBuf dKi, dD, dX, dJ, dy, dw, dp, dr, o1, o2;
    GPX_TRY(G.upload(dKi, Ki, n, n, ld));
    GPX_TRY(G.upload(dX, Kxox, m, n, ld));
    GPX_TRY(G.vec(dy, n)); GPX_TRY(G.vec(dw, n)); GPX_TRY(G.vec(dp, n)); GPX_TRY(G.vec(dr, n));
    GPX_TRY(G.vec(o1, m)); GPX_TRY(G.vec(o2, m));
    GPX_HIP(hipMemcpyAsync(dy.p, y, (size_t)n * 8, hipMemcpyHostToDevice, G.st));
    GPX_TRY(G.mv(dKi.p, n, n, ld, dy.p, dw.p));                          // w = Ki y
    std::vector<double> h1((size_t)m), h2((size_t)m);
    for (int i = 0; i < P; ++i) {
        if (i < np) {
            GPX_TRY(G.upload(dD, Kj + (size_t)i * n * n, n, n, ld));
            GPX_TRY(G.upload(dJ, Kjxo + (size_t)i * m * n, m, n, ld));
            GPX_TRY(G.mv(dD.p, n, n, ld, dw.p, dp.p));                   // p = dK_i w
            GPX_TRY(G.mv(dJ.p, m, n, ld, dw.p, o1.p));                   // dKxox_i w
        } else {
            GPX_TRY(G.scale(dw.p, dp.p, n, 2.0 * s));
        }
        GPX_TRY(G.mv(dKi.p, n, n, ld, dp.p, dr.p));                      // r = Ki p
        GPX_TRY(G.mv(dX.p, m, n, ld, dr.p, o2.p));                       // Kxox r
        if (i < np) GPX_HIP(hipMemcpyAsync(h1.data(), o1.p, (size_t)m * 8, hipMemcpyDeviceToHost, G.st));
        GPX_HIP(hipMemcpyAsync(h2.data(), o2.p, (size_t)m * 8, hipMemcpyDeviceToHost, G.st));
        GPX_HIP(hipStreamSynchronize(G.st));
        for (int64_t k = 0; k < m; ++k) dm[(size_t)i * m + k] = (i < np ? h1[(size_t)k] : 0.0) - h2[(size_t)k];   // gp_c.pyx:130-131
    }
    return GPX_OK;
}

}  // namespace gpx

using namespace gpx;

extern "C" {

static int deriv_supported(gpx_gp_t *g)
{
    if (g->kernel == GPX_KERNEL_PERIODIC && g->d != 1) {
        set_error("periodic derivative members need d == 1 (got %d)", g->d);
        return GPX_ERR_UNSUPPORTED;
    }
    if (!g->have_params) {
        set_error("the derivative stack needs a native kernel (set_params), not an uploaded matrix");
        return GPX_ERR_UNSUPPORTED;
    }
    return GPX_OK;
}

int gpx_gp_dlh_d2lh(gpx_gp_t *g, double *dlh, double *d2lh, double *d2loglh)
{
    GP_ENTER(g);
    GPX_ARG(g->fitted && (dlh || d2lh || d2loglh), "gp is not fitted / nothing requested");
    GPX_TRY(deriv_supported(g));
    if (g->dtype == GPX_F64) return d2lh_t<double>(g, dlh, d2lh, d2loglh);
    return d2lh_t<float>(g, dlh, d2lh, d2loglh);
}

int gpx_gp_dm_dtheta(gpx_gp_t *g, const double *xo, int64_t m, double *out)
{
    GP_ENTER(g);
    GPX_ARG(g->fitted, "gp is not fitted");
    GPX_ARG(m >= 0 && (m == 0 || (xo && out)), "bad arguments");
    if (m == 0) return GPX_OK;
    GPX_TRY(deriv_supported(g));
    if (g->dtype == GPX_F64) return dm_t<double>(g, xo, m, out);
    return dm_t<float>(g, xo, m, out);
}


/* ---- gp/ext/gp_c.pyx:34-131 with the reference's own arguments (host float64, C order); see the Glue section ---- */
int gpx_gp_c_dloglh_dtheta(const double *y, const double *Ki, const double *Kj, const double *Kiy, double s, int64_t n, int np,
                           double *dloglh)
{
    GPX_TRY(ensure_device());
    GPX_ARG(n >= 1 && np >= 0 && y && Ki && Kiy && dloglh && (np == 0 || Kj), "bad arguments");
    std::vector<double> a(np + 1), tr(np + 1);
    GPX_TRY(glue_first_order(y, Ki, Kj, Kiy, s, n, np, a.data(), tr.data()));
    for (int i = 0; i <= np; ++i) dloglh[i] = 0.5 * a[i] + -0.5 * tr[i];                  // gp_c.pyx:47-49
    return GPX_OK;
}

int gpx_gp_c_dlh_dtheta(const double *y, const double *Ki, const double *Kj, const double *Kiy, double s, double lh, int64_t n,
                        int np, double *dlh)
{
    GPX_TRY(ensure_device());
    GPX_ARG(n >= 1 && np >= 0 && y && Ki && Kiy && dlh && (np == 0 || Kj), "bad arguments");
    std::vector<double> a(np + 1), tr(np + 1);
    GPX_TRY(glue_first_order(y, Ki, Kj, Kiy, s, n, np, a.data(), tr.data()));
    for (int i = 0; i <= np; ++i) dlh[i] = 0.5 * lh * (a[i] - tr[i]);                     // gp_c.pyx:65-67
    return GPX_OK;
}

int gpx_gp_c_d2lh_dtheta2(const double *y, const double *Ki, const double *Kj, const double *Kh, const double *Kiy, double s,
                          double lh, const double *dlh, int64_t n, int np, double *d2lh)
{
    GPX_TRY(ensure_device());
    GPX_ARG(n >= 1 && np >= 0 && y && Ki && Kiy && dlh && d2lh && (np == 0 || (Kj && Kh)), "bad arguments");
    return glue_d2lh(y, Ki, Kj, Kh, Kiy, s, lh, dlh, n, np, d2lh);
}

int gpx_gp_c_dm_dtheta(const double *y, const double *Ki, const double *Kj, const double *Kjxo, const double *Kxox, double s,
                       int64_t n, int np, int64_t m, double *dm)
{
    GPX_TRY(ensure_device());
    GPX_ARG(n >= 1 && np >= 0 && m >= 0 && y && Ki && (m == 0 || (Kxox && dm)) && (np == 0 || (Kj && (m == 0 || Kjxo))), "bad arguments");
    if (m == 0) return GPX_OK;
    return glue_dm(y, Ki, Kj, Kjxo, Kxox, s, n, np, m, dm);
}

}  // extern "C"
