// gpx_gp.hip -- the fitted-GP device handle and the host-pointer drop-in entry points.
//
// gpx_gp_t keeps one GP resident in HBM and mirrors the memoised properties of
// the reference's gp.GP (gp/gp.py:242-396): Kxx (built lower, + s^2 on the
// diagonal) -> Lxx (in place) -> inv_Kxx_y -> logdet / y^T alpha -> log_lh, then
// posterior mean / covariance (gp/gp.py:574-625).  Data layout in HBM:
//   x      (n, d)  row-major, dtype T
//   y      (n,)
//   A      (n, lda) row-major, lda = round_up(n, 16): lower triangle holds K, then L
//   alpha  (n,)    K^-1 y
//   t0,t1  (n,)    solve scratch
//   scal   3 doubles (logdet, y^T alpha, spare) + 1 int (potrf info)
#include "gpx_common.h"
#include <cmath>
#include <vector>

#include "gpx_gp_internal.h"

namespace gpx {

template <typename T>
__global__ void cvt_from_f64(const double *__restrict__ src, T *__restrict__ dst, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (T)src[i];
}

template <typename T>
__global__ void cvt_to_f64_2d(const T *__restrict__ src, int64_t lds, double *__restrict__ dst,
                              int64_t ldd, int64_t rows, int64_t cols, int lower_only)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
        const double v = (double)src[r * lds + c];
        dst[r * ldd + c] = (lower_only && c > r) ? 0.0 : v;
    }
}

template <typename T>
__global__ void eye_kernel(T *__restrict__ X, int64_t n, int64_t ld)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ld) return;
    for (int64_t r = blockIdx.y; r < n; r += gridDim.y) X[r * ld + c] = (c == r) ? (T)1 : (T)0;
}

// upload a host f64 array into a device buffer of dtype (via a temporary when f32)
static int upload_f64(int dtype, void *dst, const double *src, int64_t count, hipStream_t st)
{
    if (count <= 0) return GPX_OK;
    if (dtype == GPX_F64) {
        GPX_HIP(hipMemcpyAsync(dst, src, count * 8, hipMemcpyHostToDevice, st));
        GPX_HIP(hipStreamSynchronize(st));
        return GPX_OK;
    }
    double *tmp = nullptr;
    GPX_HIP(hipMalloc((void **)&tmp, count * 8));
    hipError_t e = hipMemcpyAsync(tmp, src, count * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((cvt_from_f64<float>), dim3((unsigned)cdiv(count, 256)), dim3(256), 0, st,
                           tmp, (float *)dst, count);
        e = hipStreamSynchronize(st);
    }
    (void)hipFree(tmp);
    if (e != hipSuccess) return hip_fail(e, "upload_f64", __FILE__, __LINE__);
    return GPX_OK;
}

// download a device (rows x cols, lds) matrix of dtype into host f64 (ldh)
static int download_f64(int dtype, double *dst, int64_t ldh, const void *src, int64_t lds, int64_t rows,
                        int64_t cols, int lower_only, hipStream_t st)
{
    if (rows <= 0 || cols <= 0) return GPX_OK;
    if (cols == 1 && lds == 1 && ldh == 1) { cols = rows; rows = 1; lds = cols; ldh = cols; }   // vector
    double *tmp = nullptr;
    GPX_HIP(hipMalloc((void **)&tmp, (size_t)rows * cols * 8));
    dim3 grid((unsigned)cdiv(cols, 256), (unsigned)std::min<int64_t>(rows, 32768)), block(256);
    if (dtype == GPX_F64)
        hipLaunchKernelGGL((cvt_to_f64_2d<double>), grid, block, 0, st, (const double *)src, lds, tmp,
                           cols, rows, cols, lower_only);
    else
        hipLaunchKernelGGL((cvt_to_f64_2d<float>), grid, block, 0, st, (const float *)src, lds, tmp, cols,
                           rows, cols, lower_only);
    hipError_t e = hipMemcpy2DAsync(dst, (size_t)ldh * 8, tmp, (size_t)cols * 8, (size_t)cols * 8,
                                    (size_t)rows, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    if (e != hipSuccess) return hip_fail(e, "download_f64", __FILE__, __LINE__);
    return GPX_OK;
}

int kmat(int dtype, int kernel, int member, const void *x1, int64_t n, const void *x2, int64_t m,
         int d, const double *params, double diag_add, int tri, void *out, int64_t ld, hipStream_t st)
{
    return gpx_d_kmat(dtype, kernel, member, x1, n, x2, m, d, params, diag_add, tri, out, ld, (void *)st);
}

static int nparams_of(int kernel) { return kernel == GPX_KERNEL_PERIODIC ? 3 : 2; }

int check_internal_info(int info)
{
    if (info >= 0) return GPX_OK;
    set_error("internal failure inside the factorisation (info = %d: a hand-off between workgroups of the resident "
              "panel kernel timed out); the factor is not valid -- this is NOT a statement about the matrix", info);
    return GPX_ERR_INTERNAL;
}

// flag[0] |= 1 when v holds a NaN or an infinity (scipy's asarray_chkfinite on the device, O(n))
template <typename T>
__global__ void nonfinite_kernel(const T *__restrict__ v, int64_t n, int *__restrict__ flag)
{
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        bad = bad || !isfinite(v[i]);
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// x_finite / y_finite of the handle from its device arrays (synchronous)
int gp_scan_finite(gpx_gp *g)
{
    int *flags = (int *)(g->scal + 2);                    // two spare words of the scalar block
    GPX_HIP(hipMemsetAsync(flags, 0, 2 * sizeof(int), g->st));
    const int64_t nx = g->n * g->d, ny = g->n;
    const unsigned bx = (unsigned)std::min<int64_t>(cdiv(nx, 256), 1024), by = (unsigned)std::min<int64_t>(cdiv(ny, 256), 1024);
    if (g->dtype == GPX_F64) {
        hipLaunchKernelGGL((nonfinite_kernel<double>), dim3(bx), dim3(256), 0, g->st, (const double *)g->x, nx, flags);
        hipLaunchKernelGGL((nonfinite_kernel<double>), dim3(by), dim3(256), 0, g->st, (const double *)g->y, ny, flags + 1);
    } else {
        hipLaunchKernelGGL((nonfinite_kernel<float>), dim3(bx), dim3(256), 0, g->st, (const float *)g->x, nx, flags);
        hipLaunchKernelGGL((nonfinite_kernel<float>), dim3(by), dim3(256), 0, g->st, (const float *)g->y, ny, flags + 1);
    }
    GPX_LAUNCH_CHECK();
    int h[2] = {0, 0};
    GPX_HIP(hipMemcpyAsync(h, flags, sizeof(h), hipMemcpyDeviceToHost, g->st));
    GPX_HIP(hipStreamSynchronize(g->st));
    g->x_finite = h[0] == 0; g->y_finite = h[1] == 0;
    return GPX_OK;
}

// Would K(x, x) + s^2 I hold only finite numbers for finite x?  The kernel's value at r = 0 and at r = 1 in the
// host's double arithmetic: a NaN or infinite parameter (the reference's setters let both through:
// gp/kernels/gaussian.py:62-69 only reject values < EPS, gp/gp.py:192-193 only s < 0) shows up there.
// The kernel's value at r = 0 (plus s^2) and at r = 1, evaluated in the arithmetic the build will use: an fp32 handle
// whose h^2 exceeds FLT_MAX builds an infinite K although the constants are finite in double.
template <typename T>
static bool kernel_values_finite_t(int kernel, const double *p, double s)
{
    T k0, k1;
    if (kernel == GPX_KERNEL_GAUSSIAN) {
        const T c1 = (T)(-0.5 / (p[1] * p[1])), c2 = (T)(0.5 * sqrt(2.0 / M_PI) * p[0] * p[0] / p[1]);   // gaussian_c.pyx:27-28
        k0 = c2; k1 = c2 * (T)exp((double)c1);
    } else {
        const double sn = sin(0.5 / p[2]);                                                            // periodic_c.pyx:27-29
        k0 = (T)(p[0] * p[0]); k1 = k0 * (T)exp(-2.0 * sn * sn / (p[1] * p[1]));
    }
    const T diag = k0 + (T)(s * s);
    return std::isfinite(diag) && std::isfinite(k1);
}
bool kernel_values_finite(int kernel, const double *p, double s, int dtype)
{
    return dtype == GPX_F32 ? kernel_values_finite_t<float>(kernel, p, s) : kernel_values_finite_t<double>(kernel, p, s);
}

static const char *NONFINITE_MSG = "array must not contain infs or NaNs";      // scipy's text (gp/gp.py:294, 332-334)

#define GP_NEED_FINITE_Y(g)                                                    \
    do { if (!(g)->y_finite) { set_error("%s (y)", NONFINITE_MSG); return GPX_ERR_ARG; } } while (0)

}  // namespace gpx

using namespace gpx;

extern "C" {

// ------------------------------------------------------------- the handle --
int gpx_gp_create(gpx_gp_t **out, int dtype, int kernel, int64_t n, int d)
{
    GPX_TRY(ensure_device());
    GPX_ARG(out, "gp is NULL");
    *out = nullptr;
    GPX_ARG(dtype == GPX_F64 || dtype == GPX_F32, "dtype must be GPX_F64 or GPX_F32");
    GPX_ARG(kernel == GPX_KERNEL_GAUSSIAN || kernel == GPX_KERNEL_PERIODIC, "unknown kernel family");
    GPX_ARG(n >= 1 && d >= 1, "need n >= 1 and d >= 1");
    gpx_gp *g = new gpx_gp();
    memset(g, 0, sizeof(*g));
    g->dtype = dtype; g->kernel = kernel; g->n = n; g->d = d;
    if (hipGetDevice(&g->device) != hipSuccess) { (void)hipGetLastError(); g->device = 0; }
    g->nparams = nparams_of(kernel);
    g->lda = round_up(n, 16);
    const size_t es = esize(dtype);
    int rc = GPX_OK;
    hipError_t e;
#define GP_ALLOC(field, bytes)                                                        \
    if (rc == GPX_OK) {                                                               \
        e = hipMalloc((void **)&g->field, (bytes));                                   \
        if (e != hipSuccess) rc = hip_fail(e, "hipMalloc " #field, __FILE__, __LINE__); \
    }
    GP_ALLOC(x, (size_t)n * d * es);
    GP_ALLOC(y, (size_t)n * es);
    GP_ALLOC(A, (size_t)(n + 1) * g->lda * es);          // (+ one row: the right-hand side rides along in the factorisation)
    GP_ALLOC(alpha, (size_t)n * es);
    GP_ALLOC(t0, (size_t)n * es);
    GP_ALLOC(t1, (size_t)n * es);
    GP_ALLOC(scal, 4 * sizeof(double));
#undef GP_ALLOC
    if (rc == GPX_OK) {
        e = hipStreamCreateWithFlags(&g->st, hipStreamNonBlocking);
        if (e != hipSuccess) rc = hip_fail(e, "hipStreamCreate", __FILE__, __LINE__);
    }
    for (int i = 0; i < 6 && rc == GPX_OK; ++i) {
        e = hipEventCreate(&g->ev[i]);
        if (e != hipSuccess) rc = hip_fail(e, "hipEventCreate", __FILE__, __LINE__);
    }
    if (rc != GPX_OK) { gpx_gp_destroy(g); return rc; }
    *out = g;
    return GPX_OK;
}

int gpx_gp_destroy(gpx_gp_t *g)
{
    if (!g) return GPX_OK;
    gpx::DeviceGuard guard__(g->device);
    if (g->st) (void)hipStreamSynchronize(g->st);
    stream_epoch_bump();                                       // (StreamTurn: a later stream at this one's address is a different stream)
    if (g->st_ops) { (void)hipStreamSynchronize(g->st_ops); (void)hipStreamDestroy(g->st_ops); }
    if (g->ev_ops) (void)hipEventDestroy(g->ev_ops);
    void *bufs[] = {g->x, g->y, g->A, g->alpha, g->t0, g->t1, g->scal, g->bw, g->ops.buf, g->gw, g->bops.buf};
    for (void *b : bufs) if (b) (void)hipFree(b);
    for (int i = 0; i < 6; ++i) if (g->ev[i]) (void)hipEventDestroy(g->ev[i]);
    if (g->st) (void)hipStreamDestroy(g->st);
    delete g;
    return GPX_OK;
}

int gpx_gp_set_data(gpx_gp_t *g, const double *x, const double *y)
{
    GP_ENTER(g);
    GPX_ARG(g && x && y, "NULL argument");
    GPX_TRY(upload_f64(g->dtype, g->x, x, g->n * g->d, g->st));
    GPX_TRY(upload_f64(g->dtype, g->y, y, g->n, g->st));
    g->have_data = true; g->fitted = false;
    return gp_scan_finite(g);
}

int gpx_gp_set_data_device(gpx_gp_t *g, const void *x_dev, const void *y_dev)
{
    GP_ENTER(g);
    GPX_ARG(g && x_dev && y_dev, "NULL argument");
    const size_t es = esize(g->dtype);
    GPX_HIP(hipMemcpyAsync(g->x, x_dev, (size_t)g->n * g->d * es, hipMemcpyDeviceToDevice, g->st));
    GPX_HIP(hipMemcpyAsync(g->y, y_dev, (size_t)g->n * es, hipMemcpyDeviceToDevice, g->st));
    GPX_HIP(hipStreamSynchronize(g->st));      // the caller may free or overwrite the sources on return
    g->have_data = true; g->fitted = false;
    return gp_scan_finite(g);
}

int gpx_gp_set_params(gpx_gp_t *g, const double *params, double s)
{
    GP_ENTER(g);
    GPX_ARG(g && params, "NULL argument");
    GPX_ARG(!(s < 0), "invalid value for s");                  // gp/gp.py:192-193 (`val < 0`: a NaN passes, as in the reference; gpx_gp_fit then rejects it as non-finite)
    for (int i = 0; i < g->nparams; ++i) g->params[i] = params[i];
    g->s = s;
    g->have_params = true; g->fitted = false; g->have_K = false;
    return GPX_OK;
}

int gpx_gp_set_K(gpx_gp_t *g, const double *Kxx, int64_t ld)
{
    GP_ENTER(g);
    GPX_ARG(g && Kxx && ld >= g->n, "bad arguments");
    const int64_t n = g->n;
    if (g->dtype == GPX_F64) {
        GPX_HIP(hipMemcpy2DAsync(g->A, (size_t)g->lda * 8, Kxx, (size_t)ld * 8, (size_t)n * 8, (size_t)n,
                                 hipMemcpyHostToDevice, g->st));
        GPX_HIP(hipStreamSynchronize(g->st));
    } else {
        // stage as f64, convert row by row into the padded f32 matrix
        DevBuf tmp;
        GPX_TRY(tmp.alloc((size_t)n * n * 8));
        GPX_HIP(hipMemcpy2DAsync(tmp.p, (size_t)n * 8, Kxx, (size_t)ld * 8, (size_t)n * 8, (size_t)n,
                                 hipMemcpyHostToDevice, g->st));
        for (int64_t r = 0; r < n; ++r)
            hipLaunchKernelGGL((cvt_from_f64<float>), dim3((unsigned)cdiv(n, 256)), dim3(256), 0, g->st,
                               (const double *)tmp.p + r * n, (float *)g->A + r * g->lda, n);
        GPX_HIP(hipStreamSynchronize(g->st));
    }
    g->have_K = true; g->fitted = false;
    return GPX_OK;
}

// The block operators of the triangular solves (csrc/gpx_solve.hip: W_k = inv(L_kk) and its products with the neighbour
// blocks, 512 columns a block) need nothing but the block columns of L up to their own.  potrf() reports its progress
// (PotrfHook), and every `group` finished blocks their operators are built on a stream of their own, beside the
// rest of the factorisation: when it ends only the last group is still to do, and the solves take the operator route --
// one launch per block with no chain inside it -- at every size (n = 8192: backward solve 0.53 -> see DESIGN 3.3).
struct OpsAhead { gpx_gp *g; int64_t group, last; };
static int ops_ahead_step(void *user, int64_t cols_done, hipEvent_t panel_done)
{
    OpsAhead *o = (OpsAhead *)user;
    gpx_gp *g = o->g;
    const int64_t ready = std::min(cols_done / 512, o->last);   // (the trailing blocks are left to the sweep's own steps)
    if (ready - g->ops.built < o->group) return GPX_OK;
    GPX_HIP(hipStreamWaitEvent(g->st_ops, panel_done, 0));
    return trsv_ops_build_upto(g->dtype, g->A, g->n, g->lda, &g->ops, ready, g->st_ops);
}

int gpx_gp_fit(gpx_gp_t *g, int *info)
{
    GP_ENTER(g);
    GPX_ARG(g, "gp is NULL");
    GPX_ARG(g->have_data && (g->have_params || g->have_K),
            "set_data and set_params (or set_K) must be called before fit");
    // scipy.linalg.cholesky(Kxx, check_finite=True), gp/gp.py:294: a kernel matrix with NaN / inf entries is a
    // ValueError, not "not positive definite".  K is finite iff x, the kernel's constants and s^2 are.
    if (!g->have_K && (!g->x_finite || !kernel_values_finite(g->kernel, g->params, g->s, g->dtype))) {
        set_error("%s (%s)", NONFINITE_MSG, g->x_finite ? "kernel parameters or s" : "x");
        return GPX_ERR_ARG;
    }
    const size_t es = esize(g->dtype);
    int *info_dev = (int *)(g->scal + 3);
    hipStream_t st = g->st;
    GPX_HIP(hipEventRecord(g->ev[0], st));
    // Kxx = K(x, x) + s^2 I, lower triangle only (gp/gp.py:263-266)
    if (!g->have_K)
        GPX_TRY(kmat(g->dtype, g->kernel, GPX_K, g->x, g->n, g->x, g->n, g->d, g->params, g->s * g->s,
                     GPX_LOWER, g->A, g->lda, st));
    g->have_K = false;   // the factor overwrites it
    GPX_HIP(hipEventRecord(g->ev[1], st));
    // Lxx (gp/gp.py:294), in place
    // Small and mid sizes: y rides along as row n of the matrix -- every panel substitutes it, every update reduces it,
    // exactly what the forward solve L t = y would do afterwards -- so only the backward solve is left (n = 8192: the
    // two solves were 1.0 ms of an 8 ms fit).  At large n the extra row of tiles in every update costs what it saves.
    const int64_t ride_max = tune().fit_ride_max;
    const bool ride = g->n <= ride_max;
    route_hit(ride ? RT_FIT_RIDE : RT_FIT_TWO_SOLVES);
    char *row_n = (char *)g->A + (size_t)g->n * g->lda * es;
    if (ride) GPX_HIP(hipMemcpyAsync(row_n, g->y, (size_t)g->n * es, hipMemcpyDeviceToDevice, st));
    g->ops.invalidate();                                  // a new factor: its block operators are rebuilt once
    const bool ahead = tune().fit_ops_ahead != 0 && g->n >= tune().fit_ops_ahead_min &&
                       trsv_ops_ahead_ok(g->dtype, g->A, g->n, g->lda);
    // ONE instalment, when all but the last `tail` blocks are final: the build is a chain of ~11 launches batched over its
    // blocks (~0.6 ms whatever their number), so instalments of 4 blocks cost the factorisation what the solve saves
    // (n = 8192: fit 6.71 -> 6.73 ms with groups of 4, 6.62 with one; n = 12288: 15.57 -> 15.04; n = 4096: no gain), and
    // the last blocks' operators would only be waited for: the backward sweep takes those blocks by steps
    const int64_t nfull = g->n / 512, tail = tune().fit_ops_tail;
    const int64_t group = std::max<int64_t>(1, std::min(nfull, nfull - tail));
    OpsAhead oa = {g, group, std::max<int64_t>(0, nfull - tail)};
    PotrfHook hook = {ops_ahead_step, &oa};
    if (ahead) {
        if (!g->st_ops) {
            int least = 0, greatest = 0;
            GPX_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
            GPX_HIP(hipStreamCreateWithPriority(&g->st_ops, hipStreamNonBlocking, least));
            GPX_HIP(hipEventCreateWithFlags(&g->ev_ops, hipEventDisableTiming));
        }
        const size_t need = trsv_ops_bytes(g->dtype, g->n);
        if (!g->ops.buf || g->ops.bytes < need) {             // (here, not inside the factorisation's launch loop)
            if (g->ops.buf) { GPX_HIP(hipStreamSynchronize(st)); (void)hipFree(g->ops.buf); g->ops.buf = nullptr; g->ops.bytes = 0; }
            GPX_HIP(hipMalloc(&g->ops.buf, need));
            g->ops.bytes = need;
        }
        // (the operator buffer may still be read by solves of the factor before this one, queued on st)
        GPX_HIP(hipEventRecord(g->ev_ops, st));
        GPX_HIP(hipStreamWaitEvent(g->st_ops, g->ev_ops, 0));
        potrf_set_hook(&hook);
    }
    const int prc = potrf(g->dtype, g->A, g->n, g->lda, info_dev, st, nullptr, ride ? 1 : 0, /*may_block=*/true);
    potrf_set_hook(nullptr);
    GPX_TRY(prc);
    if (ahead) {
        // the solves wait for the operator stream.  The backward sweep takes the blocks that have no operators yet by
        // steps; a forward sweep (n > ride_max) and every later solve of this factor complete the set first (trsv_lower)
        GPX_HIP(hipEventRecord(g->ev_ops, g->st_ops));
        GPX_HIP(hipStreamWaitEvent(st, g->ev_ops, 0));
        if (g->ops.built > 0) route_hit(RT_FIT_OPS_AHEAD);
    }
    GPX_HIP(hipEventRecord(g->ev[2], st));
    // inv_Kxx_y = cho_solve((L, True), y) (gp/gp.py:332-334)
    if (ride) {
        GPX_TRY(trsv_lower(g->dtype, g->A, g->n, g->lda, row_n, g->alpha, 1, st, nullptr, &g->ops));
    } else {
        GPX_HIP(hipMemcpyAsync(g->t0, g->y, (size_t)g->n * es, hipMemcpyDeviceToDevice, st));
        GPX_TRY(trsv_lower(g->dtype, g->A, g->n, g->lda, g->t0, g->t1, 0, st, nullptr, &g->ops));
        GPX_TRY(trsv_lower(g->dtype, g->A, g->n, g->lda, g->t1, g->alpha, 1, st, nullptr, &g->ops));
    }
    GPX_HIP(hipEventRecord(g->ev[3], st));
    // logdet (replaces slogdet(K), gp_c.pyx:21) and y^T alpha (gp_c.pyx:26)
    GPX_TRY(logdet_chol(g->dtype, g->A, g->n, g->lda, g->scal + 0, st));
    GPX_TRY(dot(g->dtype, g->y, g->alpha, g->n, g->scal + 1, st));
    GPX_HIP(hipEventRecord(g->ev[4], st));
    g->fitted = true;
    if (info) {
        GPX_HIP(hipMemcpyAsync(info, info_dev, sizeof(int), hipMemcpyDeviceToHost, st));
        GPX_HIP(hipStreamSynchronize(st));
        GPX_TRY(check_internal_info(*info));
    }
    return GPX_OK;
}

static int gp_scalars(gpx_gp_t *g, double *logdet, double *yta, int *info)
{
    GP_ENTER(g);
    GPX_ARG(g && g->fitted, "gp is not fitted");
    double h[4];
    GPX_HIP(hipMemcpyAsync(h, g->scal, sizeof(h), hipMemcpyDeviceToHost, g->st));
    GPX_HIP(hipStreamSynchronize(g->st));
    int inf;
    memcpy(&inf, &h[3], sizeof(int));
    GPX_TRY(check_internal_info(inf));
    if (logdet) *logdet = h[0];
    if (yta) *yta = h[1];
    if (info) *info = inf;
    return GPX_OK;
}

int gpx_gp_log_lh(gpx_gp_t *g, double *log_lh)
{
    GPX_ARG(log_lh, "log_lh is NULL");
    GPX_ARG(g, "gp is NULL");
    GP_NEED_FINITE_Y(g);                                  // cho_solve(..., check_finite=True), gp/gp.py:332-334
    double logdet, yta; int info;
    GPX_TRY(gp_scalars(g, &logdet, &yta, &info));
    // gp/gp.py:362-365 (LinAlgError -> -inf) and gp_c.pyx:22-29 (sign / MIN clamp)
    if (info != 0 || !(logdet >= GPX_MIN_LOG)) { *log_lh = -INFINITY; return GPX_OK; }
    const double data_fit = -0.5 * yta;
    const double complexity_penalty = -0.5 * logdet;
    const double constant = -0.5 * (double)g->n * log(2 * M_PI);
    *log_lh = data_fit + complexity_penalty + constant;
    return GPX_OK;
}

int gpx_gp_logdet(gpx_gp_t *g, double *logdet)
{
    GPX_ARG(logdet, "logdet is NULL");
    return gp_scalars(g, logdet, nullptr, nullptr);
}

int gpx_gp_info(gpx_gp_t *g, int *info)
{
    GPX_ARG(info, "info is NULL");
    return gp_scalars(g, nullptr, nullptr, info);
}

int gpx_gp_mean(gpx_gp_t *g, const double *xo, int64_t m, double *out)
{
    GP_ENTER(g);
    GPX_ARG(g && g->fitted, "gp is not fitted");
    GP_NEED_FINITE_Y(g);
    GPX_ARG(m >= 0 && (m == 0 || (xo && out)), "bad arguments");
    if (m == 0) return GPX_OK;
    const size_t es = esize(g->dtype);
    DevBuf dxo, dout;
    GPX_TRY(dxo.alloc((size_t)m * g->d * es));
    GPX_TRY(dout.alloc((size_t)m * es));
    GPX_TRY(upload_f64(g->dtype, dxo.p, xo, m * g->d, g->st));
    GPX_TRY(gpx_d_mean(g->dtype, g->kernel, dxo.p, m, g->x, g->n, g->d, g->params, g->alpha, dout.p,
                       (void *)g->st));
    return download_f64(g->dtype, out, 1, dout.p, 1, m, 1, 0, g->st);
}

int gpx_gp_cov(gpx_gp_t *g, const double *xo, int64_t m, double *out)
{
    GP_ENTER(g);
    GPX_ARG(g && g->fitted, "gp is not fitted");
    GPX_ARG(m >= 0 && (m == 0 || (xo && out)), "bad arguments");
    if (m == 0) return GPX_OK;
    const size_t es = esize(g->dtype);
    const int64_t ldx = g->lda, ldc = round_up(m, 16);
    DevBuf dxo, X, C;
    GPX_TRY(dxo.alloc((size_t)m * g->d * es));
    GPX_TRY(X.alloc((size_t)m * ldx * es));
    GPX_TRY(C.alloc((size_t)m * ldc * es));
    GPX_TRY(upload_f64(g->dtype, dxo.p, xo, m * g->d, g->st));
    // X = Kxox (m x n); V^T = X L^-T; cov = Kxoxo - V^T V   (gp/gp.py:622-625 without K^-1)
    GPX_TRY(kmat(g->dtype, g->kernel, GPX_K, dxo.p, m, g->x, g->n, g->d, g->params, 0.0, GPX_FULL, X.p,
                 ldx, g->st));
    GPX_TRY(trsm_right_lt(g->dtype, g->A, g->n, g->lda, X.p, m, ldx, g->st, 0, &g->ops));
    GPX_TRY(kmat(g->dtype, g->kernel, GPX_K, dxo.p, m, dxo.p, m, g->d, g->params, 0.0, GPX_FULL, C.p,
                 ldc, g->st));
    GPX_TRY(gemm_nt(g->dtype, m, m, g->n, X.p, ldx, X.p, ldx, C.p, ldc, -1.0, GPX_FULL, 0, 0, g->st));
    return download_f64(g->dtype, out, m, C.p, ldc, m, m, 0, g->st);
}

int gpx_gp_mean_from_K(gpx_gp_t *g, const double *Kxox, int64_t m, double *out)
{
    GP_ENTER(g);
    GPX_ARG(g && g->fitted, "gp is not fitted");
    GP_NEED_FINITE_Y(g);
    GPX_ARG(m >= 0 && (m == 0 || (Kxox && out)), "bad arguments");
    if (m == 0) return GPX_OK;
    const size_t es = esize(g->dtype);
    const int64_t n = g->n, ldx = g->lda;
    DevBuf X, o;
    GPX_TRY(X.alloc((size_t)m * ldx * es));
    GPX_TRY(o.alloc((size_t)m * es));
    GPX_HIP(hipMemsetAsync(o.p, 0, (size_t)m * es, g->st));
    if (g->dtype == GPX_F64) {
        GPX_HIP(hipMemcpy2DAsync(X.p, (size_t)ldx * 8, Kxox, (size_t)n * 8, (size_t)n * 8, (size_t)m,
                                 hipMemcpyHostToDevice, g->st));
    } else {
        for (int64_t r = 0; r < m; ++r)
            GPX_TRY(upload_f64(g->dtype, (float *)X.p + r * ldx, Kxox + r * n, n, g->st));
    }
    GPX_TRY(gemm_nt(g->dtype, m, 1, n, X.p, ldx, g->alpha, ldx, o.p, 1, 1.0, GPX_FULL, 0, 0, g->st));
    return download_f64(g->dtype, out, 1, o.p, 1, m, 1, 0, g->st);
}

int gpx_gp_cov_from_K(gpx_gp_t *g, const double *Kxox, const double *Kxoxo, int64_t m, double *out)
{
    GP_ENTER(g);
    GPX_ARG(g && g->fitted, "gp is not fitted");
    GPX_ARG(m >= 0 && (m == 0 || (Kxox && Kxoxo && out)), "bad arguments");
    if (m == 0) return GPX_OK;
    const size_t es = esize(g->dtype);
    const int64_t n = g->n, ldx = g->lda, ldc = round_up(m, 16);
    DevBuf X, C;
    GPX_TRY(X.alloc((size_t)m * ldx * es));
    GPX_TRY(C.alloc((size_t)m * ldc * es));
    if (g->dtype == GPX_F64) {
        GPX_HIP(hipMemcpy2DAsync(X.p, (size_t)ldx * 8, Kxox, (size_t)n * 8, (size_t)n * 8, (size_t)m,
                                 hipMemcpyHostToDevice, g->st));
        GPX_HIP(hipMemcpy2DAsync(C.p, (size_t)ldc * 8, Kxoxo, (size_t)m * 8, (size_t)m * 8, (size_t)m,
                                 hipMemcpyHostToDevice, g->st));
    } else {
        for (int64_t r = 0; r < m; ++r) {
            GPX_TRY(upload_f64(g->dtype, (float *)X.p + r * ldx, Kxox + r * n, n, g->st));
            GPX_TRY(upload_f64(g->dtype, (float *)C.p + r * ldc, Kxoxo + r * m, m, g->st));
        }
    }
    GPX_TRY(trsm_right_lt(g->dtype, g->A, n, g->lda, X.p, m, ldx, g->st, 0, &g->ops));
    GPX_TRY(gemm_nt(g->dtype, m, m, n, X.p, ldx, X.p, ldx, C.p, ldc, -1.0, GPX_FULL, 0, 0, g->st));
    return download_f64(g->dtype, out, m, C.p, ldc, m, m, 0, g->st);
}

int gpx_gp_get_Kxx(gpx_gp_t *g, double *out, int64_t ld)
{
    GP_ENTER(g);
    GPX_ARG(g && g->have_data && g->have_params && out && ld >= g->n, "bad arguments");
    const size_t es = esize(g->dtype);
    DevBuf K;
    GPX_TRY(K.alloc((size_t)g->n * g->lda * es));
    GPX_TRY(kmat(g->dtype, g->kernel, GPX_K, g->x, g->n, g->x, g->n, g->d, g->params, g->s * g->s,
                 GPX_FULL, K.p, g->lda, g->st));
    return download_f64(g->dtype, out, ld, K.p, g->lda, g->n, g->n, 0, g->st);
}

int gpx_gp_get_Lxx(gpx_gp_t *g, double *out, int64_t ld)
{
    GP_ENTER(g);
    GPX_ARG(g && g->fitted && out && ld >= g->n, "bad arguments");
    return download_f64(g->dtype, out, ld, g->A, g->lda, g->n, g->n, 1, g->st);
}

int gpx_gp_get_alpha(gpx_gp_t *g, double *out)
{
    GP_ENTER(g);
    GPX_ARG(g && g->fitted && out, "bad arguments");
    GP_NEED_FINITE_Y(g);                                  // cho_solve(..., check_finite=True), gp/gp.py:332-334
    return download_f64(g->dtype, out, 1, g->alpha, 1, g->n, 1, 0, g->st);
}

int gpx_gp_get_inv_Kxx(gpx_gp_t *g, double *out, int64_t ld)
{
    GP_ENTER(g);
    GPX_ARG(g && g->fitted && out && ld >= g->n, "bad arguments");
    const size_t es = esize(g->dtype);
    const int64_t n = g->n, lda = g->lda;
    DevBuf X, C;
    GPX_TRY(X.alloc((size_t)n * lda * es));
    GPX_TRY(C.alloc((size_t)n * lda * es));
    dim3 grid((unsigned)cdiv(lda, 256), (unsigned)std::min<int64_t>(n, 32768)), block(256);
    if (g->dtype == GPX_F64) hipLaunchKernelGGL((eye_kernel<double>), grid, block, 0, g->st, (double *)X.p, n, lda);
    else hipLaunchKernelGGL((eye_kernel<float>), grid, block, 0, g->st, (float *)X.p, n, lda);
    GPX_LAUNCH_CHECK();
    GPX_HIP(hipMemsetAsync(C.p, 0, (size_t)n * lda * es, g->st));
    // X = I L^-T = L^-T ; K^-1 = L^-T L^-1 = X X^T   (gp/gp.py:311-312)
    GPX_TRY(trsm_right_lt(g->dtype, g->A, n, lda, X.p, n, lda, g->st, 1, &g->ops));
    GPX_TRY(gemm_nt(g->dtype, n, n, n, X.p, lda, X.p, lda, C.p, lda, 1.0, GPX_FULL, 0, 0, g->st, 0, 1));
    return download_f64(g->dtype, out, ld, C.p, lda, n, n, 0, g->st);
}

// the reduction half of the gradient: W = K^-1 (lower triangle, n x ldw) is in HBM; one fused pass of (alpha alpha^T - W)
// against the kernel derivatives evaluated on the fly (gp_c.pyx:34-49).  Synchronises g->st.
static int grad_reduce(gpx_gp *g, const void *alpha, const double *params, double s_noise, const void *W, int64_t ldw, double *part,
                       double *out)
{
    const int64_t n = g->n;
    double *aa = part + 1024 * 4;
    GPX_TRY(dot(g->dtype, alpha, alpha, n, aa, g->st));
    double p4[4];
    GPX_TRY(dloglh_reduce(g->dtype, g->kernel, g->x, n, g->d, params, alpha, W, ldw, part, p4, g->st));
    double ata = 0.0;
    GPX_HIP(hipMemcpyAsync(&ata, aa, sizeof(double), hipMemcpyDeviceToHost, g->st));
    GPX_HIP(hipStreamSynchronize(g->st));
    for (int i = 0; i < g->nparams; ++i) out[i] = 0.5 * p4[i];
    out[g->nparams] = s_noise * (ata - p4[3]);         // dK/ds = 2 s I  (gp_c.pyx:46)
    return GPX_OK;
}

// d log_lh / d(kernel params..., s) from a factor L (n x n, lower, in HBM) and alpha = K^-1 y: X = L^-T by the blocked
// right-looking TRSM, W = K^-1 = X X^T (lower triangle, triangular k-loop) on the MFMA kernel, then ONE fused pass
// reduces (alpha alpha^T - W) against the kernel derivatives evaluated on the fly.  X, W: n x lda scratch; part:
// 1024 * 4 + 8 doubles; ops: the block operators of THIS factor (completed here).  Synchronises `g->st`.
static int grad_from_factor(gpx_gp *g, const void *L, int64_t lda, const void *alpha, const double *params, double s_noise,
                            void *X, void *W, double *part, TrsvOps *ops, double *out)
{
    const size_t es = esize(g->dtype);
    const int64_t n = g->n;
    dim3 grid((unsigned)cdiv(lda, 256), (unsigned)std::min<int64_t>(n, 32768)), block(256);
    if (g->dtype == GPX_F64) hipLaunchKernelGGL((eye_kernel<double>), grid, block, 0, g->st, (double *)X, n, lda);
    else hipLaunchKernelGGL((eye_kernel<float>), grid, block, 0, g->st, (float *)X, n, lda);
    GPX_LAUNCH_CHECK();
    GPX_HIP(hipMemsetAsync(W, 0, (size_t)n * lda * es, g->st));
    GPX_TRY(trsm_right_lt(g->dtype, L, n, lda, X, n, lda, g->st, 1, ops));
    GPX_TRY(gemm_nt(g->dtype, n, n, n, X, lda, X, lda, W, lda, 1.0, GPX_LOWER, 0, 0, g->st, 0, 1));
    return grad_reduce(g, alpha, params, s_noise, W, lda, part, out);
}

// Gradient of the log marginal likelihood w.r.t. (kernel params..., s), RW06 eq. 5.9
// (gp/gp.py:398-433 + gp_c.pyx:34-49): K^-1 is formed on the device (X = L^-T by the blocked
// right-looking TRSM, W = X X^T lower triangle on the MFMA kernel), then ONE fused pass
// reduces (alpha alpha^T - W) against the kernel derivatives evaluated on the fly.
int gpx_gp_dloglh_dtheta(gpx_gp_t *g, double *out)
{
    GP_ENTER(g);
    GPX_ARG(g && g->fitted && out, "bad arguments");
    GP_NEED_FINITE_Y(g);
    const size_t es = esize(g->dtype);
    const int64_t n = g->n, lda = g->lda;
    double h4[4];
    GPX_HIP(hipMemcpyAsync(h4, g->scal, sizeof(h4), hipMemcpyDeviceToHost, g->st));
    GPX_HIP(hipStreamSynchronize(g->st));
    int info;
    memcpy(&info, &h4[3], sizeof(int));
    GPX_TRY(check_internal_info(info));
    if (info != 0) {                                   // gp/gp.py:424-428: NaN when K is not PD
        for (int i = 0; i <= g->nparams; ++i) out[i] = NAN;
        return GPX_OK;
    }
    DevBuf X, W, part;
    GPX_TRY(X.alloc((size_t)n * lda * es));
    GPX_TRY(W.alloc((size_t)n * lda * es));
    GPX_TRY(part.alloc((size_t)1024 * 4 * sizeof(double) + 64));
    return grad_from_factor(g, g->A, lda, g->alpha, g->params, g->s, X.p, W.p, (double *)part.p, &g->ops, out);
}

// Batched ML-II step (BASELINE config 5; the reference's inner step "set params -> read log_lh",
// gp/gp.py:216-223,337-367, for a whole table of restarts): the kernel matrices of up to `B` parameter
// rows live in HBM side by side and are factored in LOCK-STEP -- every launch of the factorisation and
// of the solves covers all of them (grid dimension y / x = matrix index), so the chain of small
// dependent launches that bounds ONE n = 8192 factorisation is paid once per batch and the chip stays
// filled by the trailing updates of all matrices.  Chunked when B matrices do not fit in free HBM.
static int fit_batch_impl(gpx_gp_t *g, const double *thetas, int64_t B, double *log_lh, double *dloglh, double *logdet_yta,
                          int *info)
{
    GPX_ARG(g->have_data, "set_data must be called before fit_batch");
    GPX_ARG(B >= 0 && (B == 0 || (thetas && log_lh)), "bad arguments");
    if (B == 0) return GPX_OK;
    if (dloglh) {
        if (g->kernel == GPX_KERNEL_PERIODIC && g->d != 1) { set_error("periodic gradient needs d == 1"); return GPX_ERR_UNSUPPORTED; }
        // gradient scratch BEFORE the chunk size is taken from what is free: X = L^-T and W = K^-1 of a GROUP of up to 8 rows
        // at a time (the group's TRSM and SYRK run in lock-step: at n = 8192 one system's far update is 1 - 2 rounds of
        // tiles), as many as a sixth of free HBM holds, + their block operators
        const size_t nl = (size_t)g->n * g->lda * esize(g->dtype);
        const bool group_ok = trsv_ops_ahead_ok(g->dtype, g->A, g->n, g->lda) && tune().trsm_ops != 0;
        size_t freeg = 0, totalg = 0;
        GPX_HIP(hipMemGetInfo(&freeg, &totalg));
        const size_t per_row = 2 * nl + (group_ok ? trsv_ops_bytes(g->dtype, g->n) : 0);
        int G = group_ok ? (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, B), (int64_t)((double)(freeg + g->gw_bytes) / 6.0 / (double)per_row))) : 1;
        if (g->gw_cap >= G && g->gw) G = g->gw_cap;
        const size_t gneed = (size_t)G * per_row + (size_t)(1024 * 4 + 8) * sizeof(double) + 256;
        if (g->gw_bytes < gneed) {
            if (g->gw) { GPX_HIP(hipStreamSynchronize(g->st)); (void)hipFree(g->gw); g->gw = nullptr; g->gw_bytes = 0; g->gw_cap = 0; }
            GPX_HIP(hipMalloc(&g->gw, gneed));
            g->gw_bytes = gneed;
        }
        g->gw_cap = G;
    }
    if (!g->x_finite || !g->y_finite) { set_error("%s (%s)", NONFINITE_MSG, g->x_finite ? "y" : "x"); return GPX_ERR_ARG; }
    const int64_t n = g->n, lda = g->lda;
    const size_t es = esize(g->dtype);
    const int np = g->nparams;
    const int64_t ride_max = tune().fit_ride_max;
    const bool ride = n <= ride_max;                      // y rides along as row n of every matrix (see gpx_gp_fit)
    const size_t per = (size_t)(n + (ride ? 1 : 0)) * lda * es;
    size_t freeb = 0, totalb = 0;
    GPX_HIP(hipMemGetInfo(&freeb, &totalb));
    int64_t Bc = (int64_t)((double)freeb * 0.85 / (double)(per + 4 * (size_t)n * es + 64));
    if (tune().batch_max_set) Bc = std::min<int64_t>(Bc, std::max<int64_t>(1, tune().batch_max));
    Bc = std::max<int64_t>(1, std::min<int64_t>(Bc, B));
    if (!g->bw && (double)per > (double)freeb * 0.85) { set_error("fit_batch: not even one more n x n matrix fits in HBM"); return GPX_ERR_NOMEM; }
    // one block, kept in the handle between calls (an ML-II loop calls this once per sweep; a fresh
    // hipMalloc of tens of GB costs more than the factorisations)
    const size_t vec = ((size_t)n * es + 255) / 256 * 256;
    if (g->bw && g->bw_cap >= Bc) Bc = std::min<int64_t>(g->bw_cap, B);
    const size_t need = (size_t)Bc * (per + 3 * vec) + (size_t)Bc * 2 * sizeof(double) + (size_t)Bc * sizeof(int) + 1024;
    if (g->bw_bytes < need) {
        if (g->bw) { GPX_HIP(hipStreamSynchronize(g->st)); (void)hipFree(g->bw); g->bw = nullptr; g->bw_bytes = 0; g->bw_cap = 0; }
        GPX_HIP(hipMalloc(&g->bw, need));
        g->bw_bytes = need; g->bw_cap = Bc;
    }
    struct Ptr { void *p; } Ab, t0, t1, al, sc, inf;
    {
        char *w = (char *)g->bw;
        Ab.p = w; w += (size_t)Bc * per;
        t0.p = w; w += (size_t)Bc * vec;
        t1.p = w; w += (size_t)Bc * vec;
        al.p = w; w += (size_t)Bc * vec;
        sc.p = w; w += ((size_t)Bc * 2 * sizeof(double) + 255) / 256 * 256;
        inf.p = w;
    }
    const int64_t sV = (int64_t)(vec / es);               // element stride between the vectors of a chunk
    hipStream_t st = g->st;
    const int64_t sM = (n + (ride ? 1 : 0)) * lda;
    std::vector<double> hs((size_t)Bc * 2);
    std::vector<int> hi((size_t)Bc), valid((size_t)Bc);
    const double eps = 2.220446049250313e-16;            // gp/kernels/gaussian.py:62-69: parameter < EPS is invalid
    for (int64_t b0 = 0; b0 < B; b0 += Bc) {
        const int cnt = (int)std::min<int64_t>(Bc, B - b0);
        for (int i = 0; i < cnt; ++i) {
            const double *th = thetas + (b0 + i) * (np + 1);
            bool ok = th[np] >= 0 && std::isfinite(th[np]);
            for (int k = 0; k < np; ++k) ok = ok && std::isfinite(th[k]) && !(th[k] < eps);
            valid[i] = ok;
            const double safe[3] = {1.0, 1.0, 1.0};      // an invalid row still takes part in the lock-step
            const double *prm = ok ? th : safe;
            const double s = ok ? th[np] : 1.0;
            GPX_TRY(kmat(g->dtype, g->kernel, GPX_K, g->x, n, g->x, n, g->d, prm, s * s, GPX_LOWER,
                         (char *)Ab.p + (size_t)i * per, lda, st));
            char *rhs = ride ? (char *)Ab.p + (size_t)i * per + (size_t)n * lda * es : (char *)t0.p + (size_t)i * vec;
            GPX_HIP(hipMemcpyAsync(rhs, g->y, (size_t)n * es, hipMemcpyDeviceToDevice, st));
        }
        Batch bm; bm.count = cnt; bm.sA = bm.sB = bm.sC = sM;
        GPX_TRY(potrf(g->dtype, Ab.p, n, lda, (int *)inf.p, st, cnt > 1 ? &bm : nullptr, ride ? 1 : 0));
        Batch bs; bs.count = cnt; bs.sA = sM; bs.sB = sV; bs.sC = 0;
        if (ride)      // rows n of the matrices (= L^-1 y) side by side as the backward solves' right-hand sides
            GPX_HIP(hipMemcpy2DAsync(t1.p, vec, (char *)Ab.p + (size_t)n * lda * es, per, (size_t)n * es, (size_t)cnt,
                                     hipMemcpyDeviceToDevice, st));
        else
            GPX_TRY(trsv_lower(g->dtype, Ab.p, n, lda, t0.p, t1.p, 0, st, &bs));
        GPX_TRY(trsv_lower(g->dtype, Ab.p, n, lda, t1.p, al.p, 1, st, &bs));
        GPX_TRY(logdet_chol(g->dtype, Ab.p, n, lda, (double *)sc.p, st, cnt, sM, 2));
        GPX_TRY(dot(g->dtype, g->y, al.p, n, (double *)sc.p + 1, st, cnt, 0, sV, 2));
        GPX_HIP(hipMemcpyAsync(hs.data(), sc.p, (size_t)cnt * 2 * sizeof(double), hipMemcpyDeviceToHost, st));
        GPX_HIP(hipMemcpyAsync(hi.data(), inf.p, (size_t)cnt * sizeof(int), hipMemcpyDeviceToHost, st));
        GPX_HIP(hipStreamSynchronize(st));
        for (int i = 0; i < cnt; ++i) GPX_TRY(check_internal_info(hi[i]));
        for (int i = 0; i < cnt; ++i) {
            const double logdet = hs[2 * i], yta = hs[2 * i + 1];
            double v;
            if (!valid[i]) v = NAN;                               // the reference raises ValueError for this row
            else if (hi[i] != 0 || !(logdet >= GPX_MIN_LOG)) v = -INFINITY;     // gp/gp.py:362-365, gp_c.pyx:22-29
            else v = -0.5 * yta - 0.5 * logdet - 0.5 * (double)n * log(2 * M_PI);
            log_lh[b0 + i] = v;
            if (info) info[b0 + i] = valid[i] ? hi[i] : -1;
            if (logdet_yta) { logdet_yta[2 * (b0 + i)] = valid[i] && hi[i] == 0 ? logdet : NAN; logdet_yta[2 * (b0 + i) + 1] = valid[i] && hi[i] == 0 ? yta : NAN; }
        }
        if (dloglh) {
            // gp/gp.py:398-433 per row, on the row's factor and alpha where the lock-step pass left them.  The reference
            // computes the gradient whenever the factorisation succeeds (no logdet < MIN test there); NaN for a row that
            // is not positive definite (gp/gp.py:424-428) or that the reference would have refused (ValueError).
            const size_t nl = (size_t)n * lda * es;
            const int G = (int)g->gw_cap;
            char *Xs = (char *)g->gw, *Ws = Xs + (size_t)G * nl;
            const size_t obytes = G > 1 ? trsv_ops_bytes(g->dtype, n) : 0;
            char *Os = Ws + (size_t)G * nl;
            double *part = (double *)(((uintptr_t)(Os + (size_t)G * obytes) + 255) / 256 * 256);
            for (int i0 = 0; i0 < cnt; i0 += G) {
                const int gc = std::min(G, cnt - i0);
                bool any = false;
                for (int i = i0; i < i0 + gc; ++i) any = any || (valid[i] && hi[i] == 0);
                const bool lock_step = G > 1 && gc > 1 && any;
                if (lock_step) {
                    // the group's K^-1 in lock-step: X = L^-T (operators of every row built first), W = X X^T lower.  Rows that are
                    // invalid or not positive definite take part (their factor is garbage; nothing of theirs is read back).
                    const char *L0 = (const char *)Ab.p + (size_t)i0 * per;
                    dim3 grid((unsigned)cdiv(lda, 256), (unsigned)std::min<int64_t>(n, 32768)), block(256);
                    for (int i = 0; i < gc; ++i) {
                        if (g->dtype == GPX_F64) hipLaunchKernelGGL((eye_kernel<double>), grid, block, 0, st, (double *)(Xs + (size_t)i * nl), n, lda);
                        else hipLaunchKernelGGL((eye_kernel<float>), grid, block, 0, st, (float *)(Xs + (size_t)i * nl), n, lda);
                        GPX_LAUNCH_CHECK();
                        TrsvOps o;                                   // (a view into the group's operator block: not owned, not freed)
                        o.buf = Os + (size_t)i * obytes; o.bytes = obytes;
                        GPX_TRY(trsv_ops_build_upto(g->dtype, L0 + (size_t)i * per, n, lda, &o, n / 512, st));
                    }
                    GPX_HIP(hipMemsetAsync(Ws, 0, (size_t)gc * nl, st));
                    GPX_TRY(trsm_right_lt_batch(g->dtype, L0, (int64_t)(per / es), n, lda, Xs, (int64_t)(nl / es), n, lda, st, 1, Os,
                                                (int64_t)(obytes / es), gc));
                    Batch bw; bw.count = gc; bw.sA = bw.sB = bw.sC = (int64_t)(nl / es);
                    GPX_TRY(gemm_nt(g->dtype, n, n, n, Xs, lda, Xs, lda, Ws, lda, 1.0, GPX_LOWER, 0, 0, st, 0, 1, &bw));
                }
                for (int i = i0; i < i0 + gc; ++i) {
                    double *o = dloglh + (b0 + i) * (np + 1);
                    if (!valid[i] || hi[i] != 0) { for (int k = 0; k <= np; ++k) o[k] = NAN; continue; }
                    const double *th = thetas + (b0 + i) * (np + 1);
                    const void *alpha_i = (char *)al.p + (size_t)i * vec;
                    if (lock_step) {
                        GPX_TRY(grad_reduce(g, alpha_i, th, th[np], Ws + (size_t)(i - i0) * nl, lda, part, o));
                    } else {
                        g->bops.invalidate();
                        GPX_TRY(grad_from_factor(g, (char *)Ab.p + (size_t)i * per, lda, alpha_i, th, th[np], Xs, Ws, part, &g->bops, o));
                    }
                }
            }
        }
    }
    return GPX_OK;
}

int gpx_gp_fit_batch(gpx_gp_t *g, const double *thetas, int64_t B, double *log_lh, int *info)
{
    GP_ENTER(g);
    return fit_batch_impl(g, thetas, B, log_lh, nullptr, nullptr, info);
}

// The batched ML-II step WITH its gradient (what turns config 5 from grid / restart evaluation into
// optimisation; gp/gp.py:398-433, gp_c.pyx:34-49 for every row of the table): the lock-step factorisation of
// gpx_gp_fit_batch, then per row K^-1 from the row's factor and the fused trace / quadratic-form pass of
// gpx_gp_dloglh_dtheta.  dloglh: HOST (B, n_params + 1) row-major, order (kernel params..., s).  logdet_yta (may be
// NULL): HOST (B, 2) = (log det K, y^T K^-1 y) per row, NaN where the factorisation failed -- from them a caller forms
// the UNCLAMPED log marginal likelihood where the reference's logdet < MIN clamp (gp_c.pyx:22-29) returns -inf.
int gpx_gp_fit_batch_grad(gpx_gp_t *g, const double *thetas, int64_t B, double *log_lh, double *dloglh, double *logdet_yta,
                          int *info)
{
    GP_ENTER(g);
    GPX_ARG(B == 0 || dloglh, "dloglh is NULL");
    GP_NEED_FINITE_Y(g);
    return fit_batch_impl(g, thetas, B, log_lh, dloglh, logdet_yta, info);
}

int gpx_gp_last_timing(gpx_gp_t *g, float *ms5)
{
    GP_ENTER(g);
    GPX_ARG(g && g->fitted && ms5, "bad arguments");
    GPX_HIP(hipEventSynchronize(g->ev[4]));
    for (int i = 0; i < 4; ++i) GPX_HIP(hipEventElapsedTime(&ms5[i], g->ev[i], g->ev[i + 1]));
    GPX_HIP(hipEventElapsedTime(&ms5[4], g->ev[0], g->ev[4]));
    return GPX_OK;
}

int gpx_gp_device_ptrs(gpx_gp_t *g, void **A, int64_t *lda, void **x, void **y, void **alpha,
                       void **stream)
{
    GPX_ARG(g, "gp is NULL");
    if (A) *A = g->A;
    if (lda) *lda = g->lda;
    if (x) *x = g->x;
    if (y) *y = g->y;
    if (alpha) *alpha = g->alpha;
    if (stream) *stream = (void *)g->st;
    return GPX_OK;
}

// ----------------------------------------------- host-pointer drop-ins --
int gpx_gaussian_c(int member, double *out, const double *x1, int64_t n, const double *x2, int64_t m,
                   double h, double w)
{
    const double p[2] = {h, w};
    return gpx_kmat_host(GPX_KERNEL_GAUSSIAN, member, out, x1, n, x2, m, 1, p, 0.0);
}

int gpx_gaussian_c_jacobian(double *out, const double *x1, int64_t n, const double *x2, int64_t m,
                            double h, double w)
{
    // gaussian_c.pyx:39-41
    GPX_TRY(gpx_gaussian_c(GPX_DK_DH, out, x1, n, x2, m, h, w));
    return gpx_gaussian_c(GPX_DK_DW, out + n * m, x1, n, x2, m, h, w);
}

int gpx_gaussian_c_hessian(double *out, const double *x1, int64_t n, const double *x2, int64_t m,
                           double h, double w)
{
    // gaussian_c.pyx:44-48
    const int mem[4] = {GPX_D2K_DHDH, GPX_D2K_DHDW, GPX_D2K_DHDW, GPX_D2K_DWDW};
    for (int i = 0; i < 4; ++i) GPX_TRY(gpx_gaussian_c(mem[i], out + (int64_t)i * n * m, x1, n, x2, m, h, w));
    return GPX_OK;
}

int gpx_periodic_c(int member, double *out, const double *x1, int64_t n, const double *x2, int64_t m,
                   double h, double w, double p)
{
    const double prm[3] = {h, w, p};
    return gpx_kmat_host(GPX_KERNEL_PERIODIC, member, out, x1, n, x2, m, 1, prm, 0.0);
}

int gpx_periodic_c_jacobian(double *out, const double *x1, int64_t n, const double *x2, int64_t m,
                            double h, double w, double p)
{
    // periodic_c.pyx:33-36
    const int mem[3] = {GPX_DK_DH, GPX_DK_DW, GPX_DK_DP};
    for (int i = 0; i < 3; ++i) GPX_TRY(gpx_periodic_c(mem[i], out + (int64_t)i * n * m, x1, n, x2, m, h, w, p));
    return GPX_OK;
}

int gpx_periodic_c_hessian(double *out, const double *x1, int64_t n, const double *x2, int64_t m,
                           double h, double w, double p)
{
    // periodic_c.pyx:39-50
    const int mem[9] = {GPX_D2K_DHDH, GPX_D2K_DHDW, GPX_D2K_DHDP, GPX_D2K_DHDW, GPX_D2K_DWDW,
                        GPX_D2K_DWDP, GPX_D2K_DHDP, GPX_D2K_DWDP, GPX_D2K_DPDP};
    for (int i = 0; i < 9; ++i) GPX_TRY(gpx_periodic_c(mem[i], out + (int64_t)i * n * m, x1, n, x2, m, h, w, p));
    return GPX_OK;
}

int gpx_gemm_nt_host(double *C, const double *A, const double *B, int64_t M, int64_t N, int64_t K)
{
    GPX_TRY(ensure_device());
    GPX_ARG(M >= 0 && N >= 0 && K >= 0, "negative dimension");
    if (M == 0 || N == 0) return GPX_OK;
    GPX_ARG(C && (K == 0 || (A && B)), "NULL pointer");
    const int64_t ldk = round_up(std::max<int64_t>(K, 1), 16), ldc = round_up(N, 16);
    DevBuf a, b, c;
    GPX_TRY(a.alloc((size_t)M * ldk * 8));
    GPX_TRY(c.alloc((size_t)M * ldc * 8));
    GPX_HIP(hipMemset(c.p, 0, (size_t)M * ldc * 8));
    if (K > 0) {
        GPX_HIP(hipMemcpy2D(a.p, (size_t)ldk * 8, A, (size_t)K * 8, (size_t)K * 8, (size_t)M, hipMemcpyHostToDevice));
        const void *bp = a.p;
        if (!(B == A && N == M)) {
            GPX_TRY(b.alloc((size_t)N * ldk * 8));
            GPX_HIP(hipMemcpy2D(b.p, (size_t)ldk * 8, B, (size_t)K * 8, (size_t)K * 8, (size_t)N, hipMemcpyHostToDevice));
            bp = b.p;
        }
        GPX_TRY(gemm_nt(GPX_F64, M, N, K, a.p, ldk, bp, ldk, c.p, ldc, 1.0, GPX_FULL, 0, 0, nullptr));
    }
    GPX_HIP(hipMemcpy2D(C, (size_t)N * 8, c.p, (size_t)ldc * 8, (size_t)N * 8, (size_t)M, hipMemcpyDeviceToHost));
    return GPX_OK;
}

int gpx_cholesky(double *L, const double *A, int64_t n, int *info)
{
    GPX_TRY(ensure_device());
    gpx::StreamTurn turn__(nullptr);                           // (this thread's scratch buffers: one stream at a time, gpx_common.h)
    GPX_ARG(n >= 0 && info, "bad arguments");
    *info = 0;
    if (n == 0) return GPX_OK;
    GPX_ARG(L && A, "NULL pointer");
    const int64_t lda = round_up(n, 16);
    DevBuf a, inf;
    GPX_TRY(a.alloc((size_t)n * lda * 8));
    GPX_TRY(inf.alloc(sizeof(int)));
    GPX_HIP(hipMemcpy2D(a.p, (size_t)lda * 8, A, (size_t)n * 8, (size_t)n * 8, (size_t)n, hipMemcpyHostToDevice));
    GPX_TRY(potrf(GPX_F64, a.p, n, lda, (int *)inf.p, nullptr, nullptr, 0, /*may_block=*/true));
    GPX_TRY(tril(GPX_F64, a.p, n, lda, nullptr));
    GPX_HIP(hipMemcpy(info, inf.p, sizeof(int), hipMemcpyDeviceToHost));
    GPX_TRY(check_internal_info(*info));
    GPX_HIP(hipMemcpy2D(L, (size_t)n * 8, a.p, (size_t)lda * 8, (size_t)n * 8, (size_t)n, hipMemcpyDeviceToHost));
    return GPX_OK;
}

int gpx_cho_solve(const double *L, int64_t n, double *b)
{
    GPX_TRY(ensure_device());
    gpx::StreamTurn turn__(nullptr);                           // (this thread's scratch buffers: one stream at a time, gpx_common.h)
    GPX_ARG(n >= 0, "n < 0");
    if (n == 0) return GPX_OK;
    GPX_ARG(L && b, "NULL pointer");
    const int64_t ldl = round_up(n, 16);
    DevBuf l, v0, v1;
    GPX_TRY(l.alloc((size_t)n * ldl * 8));
    GPX_TRY(v0.alloc((size_t)n * 8));
    GPX_TRY(v1.alloc((size_t)n * 8));
    GPX_HIP(hipMemcpy2D(l.p, (size_t)ldl * 8, L, (size_t)n * 8, (size_t)n * 8, (size_t)n, hipMemcpyHostToDevice));
    GPX_HIP(hipMemcpy(v0.p, b, (size_t)n * 8, hipMemcpyHostToDevice));
    GPX_TRY(trsv_lower(GPX_F64, l.p, n, ldl, v0.p, v1.p, 0, nullptr));
    GPX_TRY(trsv_lower(GPX_F64, l.p, n, ldl, v1.p, v0.p, 1, nullptr));
    GPX_HIP(hipMemcpy(b, v0.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return GPX_OK;
}

int gpx_gp_c_log_lh(const double *y, const double *L, const double *Kiy, int64_t n, double *log_lh)
{
    GPX_TRY(ensure_device());
    GPX_ARG(n >= 0 && log_lh, "bad arguments");
    GPX_ARG(n == 0 || (y && L && Kiy), "NULL pointer");
    double h[2] = {0.0, 0.0};
    if (n > 0) {
        // only the diagonal of L is needed: gather it on the host side of the copy
        DevBuf dg, a, b, sc;
        GPX_TRY(dg.alloc((size_t)n * 8));
        GPX_TRY(a.alloc((size_t)n * 8));
        GPX_TRY(b.alloc((size_t)n * 8));
        GPX_TRY(sc.alloc(2 * sizeof(double)));
        GPX_HIP(hipMemcpy2D(dg.p, 8, L, (size_t)(n + 1) * 8, 8, (size_t)n, hipMemcpyHostToDevice));
        GPX_HIP(hipMemcpy(a.p, y, (size_t)n * 8, hipMemcpyHostToDevice));
        GPX_HIP(hipMemcpy(b.p, Kiy, (size_t)n * 8, hipMemcpyHostToDevice));
        GPX_TRY(logdet_chol(GPX_F64, dg.p, n, 0, (double *)sc.p, nullptr));   // stride = ldl + 1 = 1
        GPX_TRY(dot(GPX_F64, a.p, b.p, n, (double *)sc.p + 1, nullptr));
        GPX_HIP(hipMemcpy(h, sc.p, sizeof(h), hipMemcpyDeviceToHost));
    }
    const double logdet = h[0];
    if (!(logdet >= GPX_MIN_LOG)) { *log_lh = -INFINITY; return GPX_OK; }        // gp_c.pyx:22-23
    *log_lh = -0.5 * h[1] + -0.5 * logdet + -0.5 * (double)n * log(2 * M_PI);     // gp_c.pyx:26-29
    return GPX_OK;
}

}  // extern "C"
