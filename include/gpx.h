/*
 * gpx.h -- C ABI of libgpx.so: the MI355X (gfx950) GP regression core.
 *
 * This is the drop-in boundary for the GP fit/predict hot path of
 * jhamrick/gaussian_processes.  Every entry point names the reference interface
 * it replaces (paths relative to the reference tree):
 *
 *   gp/ext/gaussian_c.pyx   K, jacobian, hessian, dK_*, d2K_*      (Cython -> C)
 *   gp/ext/periodic_c.pyx   K, jacobian, hessian, dK_*, d2K_*
 *   gp/ext/gp_c.pyx         log_lh
 *   gp/gp.py:294            scipy.linalg.cholesky(Kxx, lower=True)     (LAPACK dpotrf)
 *   gp/gp.py:332-334        scipy.linalg.cho_solve((L, True), y)       (LAPACK dpotrs)
 *   gp/gp.py:311-312        inv(L).T @ inv(L)
 *   gp/gp.py:597, 622-625   posterior mean / covariance
 *
 * Conventions
 *   - plain C: pointers, sizes, scalars.  No torch / numpy types.
 *   - all matrices are ROW-MAJOR (numpy C order, as the reference's
 *     np.ndarray[float64, mode='c']); `ld` = elements between consecutive rows.
 *   - every function returns an int status: GPX_OK (0) or a negative GPX_ERR_*;
 *     gpx_last_error() gives the message (thread-local).  Factorisations report
 *     LAPACK-style `info` (> 0: that leading minor is not positive definite)
 *     through an out-parameter, not through the status.  A HOST-side info is
 *     never negative: an internal failure of the factorisation is the status
 *     GPX_ERR_INTERNAL.
 *   - check_finite: the reference factors with scipy's check_finite=True
 *     (gp/gp.py:294, 332-334) -- NaN / inf in the kernel matrix or in y raise
 *     ValueError("array must not contain infs or NaNs").  The handle does the
 *     same with one O(n d) device reduction over x and y per set_data and a host
 *     check of the kernel constants: gpx_gp_fit returns GPX_ERR_ARG with that
 *     text when x, a kernel parameter or s is not finite; every getter that
 *     involves y (alpha, log_lh, mean, the derivative stack) when y is not.
 *   - "device" entry points (gpx_d_*) take DEVICE pointers and a hipStream_t
 *     passed as void* (NULL = the null stream); they enqueue work and return.
 *     Device matrices must have ld % 16 == 0 and 16-byte aligned bases.
 *     Threading: the library keeps its scratch blocks, the look-ahead side
 *     stream and its event pool PER HOST THREAD.  Work one host thread issues on
 *     two streams therefore takes turns: gpx_d_potrf, gpx_d_potrf_panel,
 *     gpx_d_trsv_lower*, gpx_d_trsm_right_lt, gpx_d_mean* and every gpx_gp_* call
 *     on another stream than the thread's previous call first wait (on the device,
 *     not the host) for that call's work (round 4; before that it was the caller's
 *     duty, and gpx_gp_fit(gp, NULL) on several handles broke it).  Drive streams
 *     that should overlap from different host threads.  Nothing is ordered while a
 *     stream is being captured.  Since round 5 the multi-GPU handle's entries
 *     (gpx_mg_*, on the handle's update stream) and the host-array helpers gpx_cholesky /
 *     gpx_cho_solve (null stream) take their turn in the same way, and "another stream"
 *     is decided by the stream AND an epoch the library bumps whenever it destroys one
 *     (a new stream at a destroyed one's address is a different stream); streams the
 *     caller destroys with HIP directly are the caller's to order.  Handles
 *     (gpx_gp_*) remember the device they were created on and make it current
 *     for the duration of every call (the caller's current device is restored);
 *     they synchronise their stream before they return -- the one exception is
 *     gpx_gp_fit(gp, NULL), which only enqueues -- and different handles are
 *     safe to use concurrently from different host threads.
 *     The first factorisation a host thread issues creates that thread's look-ahead
 *     stream and scratch blocks: tens of ms once (measured with round 2's streams: 8 x n = 8192
 *     lock-step, 29 ms from a long-lived thread, 58 ms from a thread started per call)
 *     -- keep worker threads alive across calls.
 *   - "host" entry points (gpx_gaussian_c_*, gpx_periodic_c_*, gpx_gp_c_*,
 *     gpx_cholesky, gpx_cho_solve ...) take HOST pointers with the reference's
 *     exact argument meaning, run on the current device and return when the
 *     result is in the caller's buffer.
 *   - no CPU fallback exists: without a usable GPU every compute entry point
 *     returns GPX_ERR_NO_DEVICE.
 */
#ifndef GPX_H
#define GPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPX_VERSION 100

/* status codes */
#define GPX_OK              0
#define GPX_ERR_ARG        -1   /* bad argument (maps to ValueError)              */
#define GPX_ERR_HIP        -2   /* HIP runtime failure (maps to RuntimeError)     */
#define GPX_ERR_NO_DEVICE  -3   /* no usable GPU                                  */
#define GPX_ERR_NOMEM      -4   /* device allocation failed                       */
#define GPX_ERR_UNSUPPORTED -5  /* combination not implemented                    */
#define GPX_ERR_INTERNAL   -6   /* the factorisation itself failed (a hand-off between workgroups of the resident
                                   panel kernel timed out: device info < 0).  Maps to RuntimeError -- it is never
                                   reported as "not positive definite" (info > 0) nor as log_lh = -inf        */

/* dtype */
#define GPX_F64 0
#define GPX_F32 1

/* kernel families (gp/kernels/gaussian.py, gp/kernels/periodic.py) */
#define GPX_KERNEL_GAUSSIAN 0   /* params = (h, w)    */
#define GPX_KERNEL_PERIODIC 1   /* params = (h, w, p) */

/* members of a kernel family: the function and its parameter derivatives.
 * Gaussian: gaussian_c.pyx:18-164.  Periodic: periodic_c.pyx:18-235. */
#define GPX_K          0
#define GPX_DK_DH      1
#define GPX_DK_DW      2
#define GPX_DK_DP      3   /* periodic only */
#define GPX_D2K_DHDH   4
#define GPX_D2K_DHDW   5   /* == d2K_dwdh */
#define GPX_D2K_DHDP   6   /* periodic only; == d2K_dpdh */
#define GPX_D2K_DWDW   7
#define GPX_D2K_DWDP   8   /* periodic only; == d2K_dpdw */
#define GPX_D2K_DPDP   9   /* periodic only */

/* triangle selector for symmetric builds */
#define GPX_FULL  0
#define GPX_LOWER 1   /* only tiles touching the lower triangle are written */

/* MIN = log(2^-1018): gp/gp.py:17, gaussian_c.pyx:15, gp_c.pyx:14 */
#define GPX_MIN_LOG (-705.6238298100243)

/* ---------------------------------------------------------------- runtime -- */
int         gpx_version(void);
const char *gpx_last_error(void);
int gpx_device_count(int *count);
int gpx_set_device(int device);
int gpx_get_device(int *device);
/* name: caller buffer; cus / clock_mhz / hbm_bytes may be NULL */
int gpx_device_info(int device, char *name, size_t name_len, int *cus,
                    int *clock_mhz, uint64_t *hbm_bytes);

int gpx_malloc(void **dptr, size_t bytes);
int gpx_free(void *dptr);
/* free / total HBM of the current device, bytes (hipMemGetInfo) */
int gpx_mem_info(size_t *free_bytes, size_t *total_bytes);
int gpx_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
int gpx_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
int gpx_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream);
/* strided copies: `rows` rows of `row_bytes`, pitches in bytes */
int gpx_memcpy2d_h2d(void *dst, size_t dpitch, const void *src, size_t spitch,
                     size_t row_bytes, size_t rows, void *stream);
int gpx_memcpy2d_d2h(void *dst, size_t dpitch, const void *src, size_t spitch,
                     size_t row_bytes, size_t rows, void *stream);
int gpx_memset(void *dst, int value, size_t bytes, void *stream);

int gpx_stream_create(void **stream);
int gpx_stream_destroy(void *stream);
int gpx_stream_sync(void *stream);
int gpx_device_sync(void);
int gpx_event_create(void **event);
int gpx_event_destroy(void *event);
int gpx_event_record(void *event, void *stream);
int gpx_event_sync(void *event);
int gpx_event_elapsed_ms(void *start, void *stop, float *ms);
int gpx_stream_wait_event(void *stream, void *event);

/* Live per-kernel-class timing.  When enabled every launch of the hot-path
 * kernels is bracketed by HIP events on its own stream; gpx_prof_read sums, per
 * class, the launches, their elapsed milliseconds and their ALGORITHMIC work
 * (flops for GPX_PROF_GEMM, bytes for the others).  bench.py derives
 * roofline.achieved from these. */
#define GPX_PROF_KMAT       0   /* bytes written                                  */
#define GPX_PROF_GEMM       1   /* gemm_nt_fast_kernel<T,128,1>: trailing updates (gpx_d_syrk_bc) on 128 x 128 tiles; flops: 2*K per updated element */
#define GPX_PROF_POTRF_DIAG 2   /* flops jb^3/3                                   */
#define GPX_PROF_TRSM_ROWS  3   /* flops rows*jb^2                                */
#define GPX_PROF_TRSV       4   /* bytes of L read                                */
#define GPX_PROF_MEAN       5   /* kernel evaluations m*n                         */
#define GPX_PROF_REDUCE     6   /* bytes read                                     */
#define GPX_PROF_GEMM_SKINNY  7 /* gemm_nt_fast_kernel<T,64> (panel products); flops */
#define GPX_PROF_GEMM_GENERIC 8 /* gemm_nt_kernel<T> (unaligned / K-tail shapes); flops */
#define GPX_PROF_GEMM_PANEL   9 /* gemm_nt_fast_kernel<T,128,0>: panel / covariance products; flops */
#define GPX_PROF_GEMM_N64    10 /* gemm_nt_fast_kernel<T,64,1>: trailing updates of few tiles on 128 x 64 tiles; flops */
int gpx_prof_enable(int on);    /* also clears the registry */
int gpx_prof_read(int cls, double *launches, double *total_ms, double *total_work);

/* Route counters: how often each of the alternative host-side routes was taken since the last reset (counted at
 * the point of decision, always on).  Every behaviour switch of the library is an environment variable; each entry point
 * takes one snapshot of them for the calling thread (csrc/gpx_tune.h); a test that forces a route
 * asserts it here. */
#define GPX_ROUTE_TRSV_OPS        0   /* single-rhs solve: operator form (one launch per 512-block step)      */
#define GPX_ROUTE_TRSV_STEPS      1   /* single-rhs solve: two launches per 512-block                         */
#define GPX_ROUTE_PANEL_RES       2   /* panel: the resident one-launch kernel                                */
#define GPX_ROUTE_PANEL_CHAIN     3   /* panel: 64-wide leaf + row substitution launches                      */
#define GPX_ROUTE_FIT_RIDE        4   /* fit: y rode along in the factorisation (forward solve folded in)     */
#define GPX_ROUTE_FIT_TWO_SOLVES  5   /* fit: forward and backward solve after the factorisation              */
#define GPX_ROUTE_GEMM_FAST       6   /* product on the LDS-DMA MFMA kernel                                   */
#define GPX_ROUTE_GEMM_GENERIC    7   /* product on the generic (unaligned / K-tail) kernel                   */
#define GPX_ROUTE_SYRK_EXACT      8   /* trailing update with the exact tile enumeration                      */
#define GPX_ROUTE_SYRK_PATCH      9   /* trailing update on the 1024 x 1024 patch grid                        */
#define GPX_ROUTE_MG_BCAST_ONE   10   /* multi-GPU panel broadcast: one collective per row chunk              */
#define GPX_ROUTE_MG_BCAST_SAG   11   /* multi-GPU panel broadcast: scatter + all-gather (point to point)     */
#define GPX_ROUTE_FIT_OPS_AHEAD  12   /* gpx_gp_fit: block operators of the solves built beside the factorisation */
#define GPX_ROUTE_TRSM_OPS       13   /* X L^-T (posterior covariance, inverse): in-block step as one product with inv(L_kk) */
#define GPX_ROUTE_POTRF_PAIR     14   /* a factorisation that entered the pair phase: far trailing updates of depth K = 2048, one per two panels */
int gpx_debug_route_count(int route, int64_t *count);
/* roctx ranges pushed so far (GPX_ROCTX=1: every gpx_gp_* call and every launch class below it is a nested host range for
 * `rocprofv3 --marker-trace`; libroctx64.so is loaded on first use; 0 while the switch is off) */
int gpx_debug_roctx_ranges(int64_t *count);
int gpx_debug_route_reset(void);
/* Switch snapshots taken so far: every entry point re-reads the GPX_* environment variables ONCE for the calling thread
 * (one pass over `environ`; csrc/gpx_tune.h is the table of all of them); nothing below an entry point reads the
 * environment -- libgpx.so does not import getenv. */
int gpx_debug_tune_refreshes(int64_t *count);
/* The asm-scheduled MFMA leaf of the panel kernel (csrc/gpx_leaf.h) carries wait states measured on gfx950; before a process
 * first uses it on a device it is checked there against the compiler-scheduled leaf (full-mantissa 256 x 256 panel, alone and
 * beside a product that loads every matrix pipe).  *state: 0 not run yet, 1 passed, 2 failed -- the library then uses the
 * compiler-scheduled leaf and says so on stderr.  run_now != 0: run it now if it has not run. */
int gpx_debug_leaf_selfcheck(int run_now, int *state);

/* ------------------------------------------------- device-level hot path -- */

/* Kernel-matrix build: out[i, j] = member(x1[i, :], x2[j, :]) (+ diag_add if i == j).
 * Replaces gaussian_c.K / periodic_c.K (and the d*K members) plus the
 * `K += eye(n) * s**2` of gp/gp.py:265 (diag_add = s*s, applied where i == j).
 * x1: (n, d), x2: (m, d) row-major, densely packed; out: (n, m) with ld.
 * params: HOST array (h, w[, p]) as doubles.  tri = GPX_LOWER skips tiles that
 * lie strictly above the diagonal (only meaningful for x1 == x2).
 * d > 1 (an extension: the reference is 1-D) uses r2 = sum_k (x1[i,k]-x2[j,k])^2;
 * periodic members other than GPX_K require d == 1. */
int gpx_d_kmat(int dtype, int kernel, int member, const void *x1, int64_t n,
               const void *x2, int64_t m, int d, const double *params,
               double diag_add, int tri, void *out, int64_t ld, void *stream);

/* Fused posterior mean  out[i] = sum_j K(xo[i], x[j]) * alpha[j]   (gp/gp.py:597
 * without materialising Kxox).  xo: (m, d), x: (n, d), alpha: (n,), out: (m,). */
int gpx_d_mean(int dtype, int kernel, const void *xo, int64_t m, const void *x,
               int64_t n, int d, const double *params, const void *alpha,
               void *out, void *stream);

/* The same with any member of the kernel family in place of K (the Jacobian / Hessian members of
 * gaussian_c.pyx:51-164, periodic_c.pyx:53-235): out = member(xo, x) @ alpha, never materialised.
 * Periodic members other than GPX_K need d == 1. */
int gpx_d_mean_member(int dtype, int kernel, int member, const void *xo, int64_t m, const void *x,
                      int64_t n, int d, const double *params, const void *alpha, void *out, void *stream);

/* C (M x N, ldc) += alpha * A (M x K, lda) * B (N x K, ldb)^T -- the MFMA work-horse.
 * tri = GPX_LOWER: only tiles with some row >= col are touched and elements
 * with col > row are left unchanged (SYRK form; requires M == N geometry with
 * row/col origins equal).  row0/col0 shift the triangle test:
 * element (i, j) is "lower" when row0 + i >= col0 + j. */
int gpx_d_gemm_nt(int dtype, int64_t M, int64_t N, int64_t K, double alpha,
                  const void *A, int64_t lda, const void *B, int64_t ldb, void *C,
                  int64_t ldc, int tri, int64_t row0, int64_t col0, void *stream);

/* Blocked right-looking Cholesky, lower, in place (replaces scipy.linalg.cholesky,
 * gp/gp.py:294 -> LAPACK dpotrf).  Only the lower triangle of A is read; the
 * strict upper triangle is left untouched (use gpx_d_tril to zero it).
 * info_dev: DEVICE int; 0 on success, j (1-based) if the j-th leading minor is
 * not positive definite (pivot <= 0 or NaN), as LAPACK reports it; negative: an
 * internal failure (the host entry points turn that into GPX_ERR_INTERNAL).
 * A pure enqueue at every size (round 4: the host-paced panel launches that made
 * the call block for n <= 16384 are confined to the handle's gpx_gp_fit, which
 * says so; nothing here waits on the host, so the call is legal under stream
 * capture); the result is complete in stream order. */
int gpx_d_potrf(int dtype, void *A, int64_t n, int64_t lda, int *info_dev,
                void *stream);

/* The two building blocks of gpx_d_potrf, exported for multi-GPU drivers (1-D
 * block-cyclic block columns).
 * gpx_d_potrf_panel factors rows [r0, n) x columns [c0, c0 + kb) of A in place; the
 * kb x kb diagonal block sits at (r0, c0) (c0 == r0 on one GPU; a local column
 * offset on a distributed matrix).  info_dev as in gpx_d_potrf (index r0-based
 * global: r0 + j + 1), never cleared here.
 * gpx_d_syrk_bc applies a factored panel Pb (row i = global row k0 + i, kb columns,
 * ldp) to this rank's block columns: local columns [cl0, cl1) of Cloc (n rows,
 * ldc), rows [row_begin, n); local block jl is global block jl * P + rank of width
 * nb;  C[g, c] -= sum_k Pb[g - k0, k] * Pb[gcol(c) - k0, k]  where g >= gcol(c). */
int gpx_d_potrf_panel(int dtype, void *A, int64_t lda, int64_t n, int64_t r0, int64_t c0,
                      int64_t kb, int *info_dev, void *stream);
int gpx_d_syrk_bc(int dtype, int64_t n, int64_t row_begin, void *Cloc, int64_t ldc,
                  int64_t cl0, int64_t cl1, const void *Pb, int64_t ldp, int64_t k0,
                  int64_t kb, int64_t nb, int P, int rank, void *stream);

/* Zero the strict upper triangle (scipy's cholesky returns a clean L). */
int gpx_d_tril(int dtype, void *A, int64_t n, int64_t lda, void *stream);

/* x <- L^-1 b (transpose = 0, forward) or x <- L^-T b (transpose = 1, backward),
 * single right-hand side; b is used as scratch and destroyed, x != b.
 * Forward then backward = cho_solve((L, True), b), gp/gp.py:332-334 (dpotrs). */
int gpx_d_trsv_lower(int dtype, const void *L, int64_t n, int64_t ldl, void *b,
                     void *x, int transpose, void *stream);

/* Distributed-solve building blocks (block columns live on different GPUs).
 * gpx_d_trsv_lower_cols: forward substitution through a trapezoid -- an
 * ncols x ncols lower triangle on top of n - ncols further rows (one factored
 * block column, rows from its diagonal down): x[0:ncols] <- solution,
 * b[ncols:n] -= L[ncols:n, 0:ncols] x.  b is scratch, x != b.
 * gpx_d_panel_gemv_t: y[c] -= sum_r Lp[r, c] * x[r] for c < ncols, r < rows (the
 * sub-diagonal part of a block column, transposed); work: cdiv(rows, 256) * ncols
 * doubles of DEVICE scratch.  Deterministic (no atomics). */
int gpx_d_trsv_lower_cols(int dtype, const void *L, int64_t n, int64_t ldl, int64_t ncols,
                          void *b, void *x, void *stream);
int gpx_d_panel_gemv_t(int dtype, const void *Lp, int64_t ldl, int64_t rows, int64_t ncols,
                       const void *x, void *y, void *work, void *stream);

/* X (m x n, ldx) <- X * L^-T  : rows of X are right-hand sides; i.e. solves
 * L * x_row^T = b_row^T for every row.  With X = Kxox this yields V^T where
 * V = L^-1 Kxxo, the factor of the posterior covariance (gp/gp.py:622-625). */
int gpx_d_trsm_right_lt(int dtype, const void *L, int64_t n, int64_t ldl, void *X,
                        int64_t m, int64_t ldx, void *stream);

/* out_dev[0] = 2 * sum_i log L[i,i]  (f64 accumulation for both dtypes).
 * Replaces the LU-based np.linalg.slogdet(K) of gp_c.pyx:21 given L. */
int gpx_d_logdet_chol(int dtype, const void *L, int64_t n, int64_t ldl,
                      double *out_dev, void *stream);

/* out_dev[0] = sum_i a[i] * b[i]  (f64 accumulation).  np.dot(y, Kiy), gp_c.pyx:26 */
int gpx_d_dot(int dtype, const void *a, const void *b, int64_t n, double *out_dev,
              void *stream);

/* ------------------------------------------------ fitted-GP device handle -- */
/* One handle = one GP resident in HBM: x, y, the kernel matrix / its factor
 * (in place), alpha = K^-1 y, logdet, y^T alpha.  Mirrors the memoised
 * properties of gp.GP (gp/gp.py:242-396). */
typedef struct gpx_gp gpx_gp_t;

int gpx_gp_create(gpx_gp_t **gp, int dtype, int kernel, int64_t n, int d);
int gpx_gp_destroy(gpx_gp_t *gp);
/* x: (n, d) HOST float64; y: (n,) HOST float64 (converted to dtype on upload) */
int gpx_gp_set_data(gpx_gp_t *gp, const double *x, const double *y);
/* same, from DEVICE buffers (on the handle's device) already in the handle's dtype.  The
 * copies run on the handle's own non-blocking stream: whatever produced x_dev / y_dev must
 * have completed (synchronise the producing stream first).  Returns after the copies are
 * done, so the sources may be freed or overwritten at once. */
int gpx_gp_set_data_device(gpx_gp_t *gp, const void *x_dev, const void *y_dev);
/* params = (h, w[, p]); s = noise standard deviation (gp/gp.py:190-197) */
int gpx_gp_set_params(gpx_gp_t *gp, const double *params, double s);
/* Plugin kernels (any gp.kernels.Kernel subclass without a native id):
 * the host evaluates its own K(x, x) + s^2 I (gp/gp.py:263-266) and hands the
 * full (n, n) HOST float64 matrix over; gpx_gp_fit then skips the kernel build. */
int gpx_gp_set_K(gpx_gp_t *gp, const double *Kxx, int64_t ld);
/* kernel build (lower) -> potrf -> alpha -> logdet, y^T alpha.  info != NULL: *info (HOST)
 * is filled and the call returns after the fit has completed.  info == NULL: the call returns without waiting for
 * the end of the fit (the next gpx_gp_* getter synchronises); for n <= 16384 it paces its panel launches on the
 * device's progress, so it returns when most of the factorisation has run, not at once. */
int gpx_gp_fit(gpx_gp_t *gp, int *info);
/* log marginal likelihood with the reference's conventions (gp/gp.py:360-367,
 * gp_c.pyx:17-31): -inf when the factorisation failed or logdet < MIN. */
int gpx_gp_log_lh(gpx_gp_t *gp, double *log_lh);
int gpx_gp_logdet(gpx_gp_t *gp, double *logdet);
/* potrf info of the last fit (0 ok, j > 0: j-th leading minor not positive definite) */
int gpx_gp_info(gpx_gp_t *gp, int *info);
/* posterior mean at xo (m, d) HOST float64 -> out (m,) HOST float64 */
int gpx_gp_mean(gpx_gp_t *gp, const double *xo, int64_t m, double *out);
/* posterior covariance at xo -> out (m, m) HOST float64, via V = L^-1 Kxxo */
int gpx_gp_cov(gpx_gp_t *gp, const double *xo, int64_t m, double *out);
/* plugin-kernel forms: the caller supplies Kxox (m, n) [and Kxoxo (m, m)] as HOST
 * float64; mean = Kxox alpha, cov = Kxoxo - (Kxox L^-T)(Kxox L^-T)^T on the device */
int gpx_gp_mean_from_K(gpx_gp_t *gp, const double *Kxox, int64_t m, double *out);
int gpx_gp_cov_from_K(gpx_gp_t *gp, const double *Kxox, const double *Kxoxo, int64_t m,
                      double *out);
/* copy-outs to HOST float64: Kxx is rebuilt (full, + s^2 I); L has zero upper */
int gpx_gp_get_Kxx(gpx_gp_t *gp, double *out, int64_t ld);
int gpx_gp_get_Lxx(gpx_gp_t *gp, double *out, int64_t ld);
int gpx_gp_get_alpha(gpx_gp_t *gp, double *out);
/* K^-1 = L^-T L^-1 (gp/gp.py:311-312) -> out (n, n) HOST float64 */
int gpx_gp_get_inv_Kxx(gpx_gp_t *gp, double *out, int64_t ld);
/* d log_lh / d(theta): out[n_params + 1] HOST float64, order (kernel params..., s).  RW06 eq. 5.9
 * (gp/gp.py:398-433, gp_c.pyx:34-49) computed on the device: K^-1 by TRSM + SYRK on the MFMA
 * kernel, then one fused pass against kernel derivatives evaluated on the fly.  All NaN when the
 * factorisation failed (gp/gp.py:424-428).  Periodic kernel: d == 1 only. */
int gpx_gp_dloglh_dtheta(gpx_gp_t *gp, double *out);
/* dlh / d(theta) (gp/gp.py:435-465, gp_c.pyx:52-67) and d2lh / d(theta)^2 (gp/gp.py:467-502,
 * gp_c.pyx:70-111), both HOST float64: dlh[n_params + 1], d2lh[(n_params + 1)^2] row-major, parameter
 * order (kernel params..., s); either pointer may be NULL.  Device resident: K^-1, the products
 * K^-1 dK_i (one MFMA GEMM per kernel parameter) and every trace / quadratic form stay in HBM, the
 * kernel derivatives inside traces and quadratic forms are evaluated on the fly; only the scalars
 * return.  All NaN when the factorisation failed (gp/gp.py:458-462,493-497).  Native kernels only
 * (periodic: d == 1).  d2loglh (extension, may be NULL): the Hessian of the LOG marginal likelihood,
 * d2lh / lh - (dlh / lh)(dlh / lh)^T, from the same pass; it stays finite where lh underflows to 0
 * (log_lh < MIN, i.e. any n beyond a few hundred) and the reference's lh-scaled Hessian is all zeros. */
int gpx_gp_dlh_d2lh(gpx_gp_t *gp, double *dlh, double *d2lh, double *d2loglh);
/* d mean(xo) / d(theta) -> out (n_params + 1, m) HOST float64 (gp/gp.py:627-662, gp_c.pyx:114-131):
 * dK_i(xo, x) alpha - K(xo, x) K^-1 dK_i alpha with fused on-the-fly mat-vecs and two triangular
 * solves per parameter; no n x n matrix is formed. */
int gpx_gp_dm_dtheta(gpx_gp_t *gp, const double *xo, int64_t m, double *out);
/* Batched ML-II step (BASELINE config 5; the reference's inner step "set params -> read log_lh",
 * gp/gp.py:216-223,337-367, for a table of restarts on the handle's data set).
 * thetas: HOST (B, n_params + 1) row-major, rows (kernel params..., s); log_lh: HOST B doubles;
 * info: HOST B ints or NULL (potrf info per row; -1 for a row with invalid parameters).
 * A row whose parameters the reference rejects with ValueError (kernel parameter < EPS, s < 0,
 * non-finite) yields NaN; a non-positive-definite row or logdet < MIN yields -inf.
 * The kernel matrices of the rows are held in HBM together and factored in lock-step (every
 * launch covers all of them); chunked by free memory (cap: environment GPX_BATCH_MAX).
 * Leaves the handle's own fitted state untouched. */
int gpx_gp_fit_batch(gpx_gp_t *gp, const double *thetas, int64_t B, double *log_lh, int *info);
/* The same step WITH the gradient of every row (gp/gp.py:398-433 `dloglh_dtheta`, gp_c.pyx:34-49 -- the quantity the
 * reference's removed `fit_MLII` optimised, CHANGELOG.md:19): the lock-step factorisation above, then per row
 * K^-1 = L^-T L^-1 from the row's own factor and the fused trace / quadratic-form pass of gpx_gp_dloglh_dtheta.
 * dloglh: HOST (B, n_params + 1) row-major, order (kernel params..., s); all NaN for a row that is not positive
 * definite (gp/gp.py:424-428) or has invalid parameters.  logdet_yta (may be NULL): HOST (B, 2) = (log det K,
 * y^T K^-1 y), NaN where the factorisation failed: from them a caller forms the log marginal likelihood WITHOUT the
 * reference's logdet < MIN clamp (gp_c.pyx:22-29), which returns -inf for any well-conditioned n beyond a few
 * thousand and leaves an optimiser nothing to follow (an extension; `log_lh` itself keeps the clamp). */
int gpx_gp_fit_batch_grad(gpx_gp_t *gp, const double *thetas, int64_t B, double *log_lh, double *dloglh,
                          double *logdet_yta, int *info);
/* Checkpoint of a fitted handle (the reference persists by pickling its memoised host arrays,
 * gp/gp.py:78-92; a 32 GiB factor cannot go that way).  File: header (dtype, kernel, n, d, params, s,
 * logdet, y^T alpha, info), x, y, alpha as float64, then the LOWER trapezoid of L in row blocks
 * (rows [r0, r1) x columns [0, r1), packed, handle dtype).  Streamed through two pinned staging
 * buffers of GPX_IO_BLOCK_BYTES (default 64 MiB) each: host memory use does not grow with n.
 * gpx_gp_load creates a NEW handle on the current device, fitted, without recomputing anything. */
int gpx_gp_save(gpx_gp_t *gp, const char *path);
int gpx_gp_load(gpx_gp_t **gp, const char *path);
/* what a handle holds (any pointer may be NULL); x: (n, d), y: (n,) HOST float64 */
int gpx_gp_describe(gpx_gp_t *gp, int *dtype, int *kernel, int64_t *n, int *d, double *params3, double *s);
int gpx_gp_get_xy(gpx_gp_t *gp, double *x, double *y);
/* timing of the last fit, milliseconds per stage (HIP events on the handle's
 * stream): [0] kernel build [1] potrf [2] solve [3] logdet+dot [4] total.
 * Up to n = 16384 the forward substitution L t = y is done inside [1] (y is carried
 * as one more row of the matrix through the factorisation) and [2] is the backward
 * solve alone. */
int gpx_gp_last_timing(gpx_gp_t *gp, float *ms5);
/* raw device views for tests / multi-GPU drivers (do not free); A has n + 1 rows of lda
 * elements (row n is the handle's work row, see gpx_gp_last_timing) */
int gpx_gp_device_ptrs(gpx_gp_t *gp, void **A, int64_t *lda, void **x, void **y,
                       void **alpha, void **stream);

/* ------------------------------------------------------------- multi-GPU -- */
/* One GP spread over the GPUs of a node, one PROCESS per GPU (north-star; SURVEY 8e -- the reference has
 * nothing distributed).  1-D block-cyclic block columns of width nb (global block column j on rank
 * j % world); per panel the owner factors, packs and broadcasts it (root = owner, row-chunked so that
 * the next owner's column update hides under the transfer), every rank updates its own block columns;
 * one-panel look-ahead on a high-priority side stream.  The whole schedule -- HIP streams, events and
 * the collectives -- runs in C.
 * Communicator: RCCL over xGMI (librccl is dlopen'ed at first use; libgpx has no link-time dependency on
 * it).  Rank 0 calls gpx_mg_unique_id and distributes the GPX_MG_ID_BYTES out of band (MPI, a file,
 * torch.distributed over gloo ...); every rank then calls gpx_mg_create on its own device
 * (gpx_set_device first).  gpx_mg_create_cb runs the same schedule over host-supplied collectives on
 * DEVICE pointers (tests: several ranks on one GPU over gloo, which RCCL cannot do). */
#define GPX_MG_ID_BYTES 128
typedef struct gpx_mg gpx_mg_t;
typedef int (*gpx_mg_bcast_fn)(void *user, void *dev_ptr, size_t bytes, int root, void *stream);
/* dtype: GPX_F64 / GPX_F32 / 2 (int32); op: 0 sum, 1 max; in place on dev_ptr; 0 = success */
typedef int (*gpx_mg_allreduce_fn)(void *user, void *dev_ptr, size_t count, int dtype, int op, void *stream);
/* can RCCL be used here at all?  dlopen + symbol lookup only (no bootstrap thread, no socket): the pre-flight a
 * launcher runs on every rank before anybody enters ncclCommInitRank */
int gpx_mg_probe(void);
int gpx_mg_unique_id(void *id128);
/* Two-step creation, so that a launcher can AGREE on the outcome of the local step (device memory: the local
 * matrix and two panel buffers; streams) over its own control plane before any rank enters ncclCommInitRank -- a
 * rank that fails to allocate would otherwise leave the others blocked in there for ever:
 *   gpx_mg_create_local (no communicator yet)  ->  all ranks exchange ok / not ok  ->  gpx_mg_connect.
 * gpx_mg_create does both in one call (single process / tests). */
int gpx_mg_create_local(gpx_mg_t **mg, int dtype, int kernel, int64_t n, int d, int64_t nb, int world, int rank);
int gpx_mg_connect(gpx_mg_t *mg, const void *id128);
int gpx_mg_create(gpx_mg_t **mg, int dtype, int kernel, int64_t n, int d, int64_t nb, int world, int rank,
                  const void *id128);
/* what the communicator itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice); nranks = 0 for
 * the callback back-end.  bcast_sag: the panel-broadcast algorithm in force.  Any pointer may be NULL. */
int gpx_mg_comm_info(gpx_mg_t *mg, int *nranks, int *rank, int *device, int *bcast_sag);
/* panel broadcast algorithm: 0 one collective per row chunk (ncclBroadcast), 1 scatter + all-gather by grouped
 * ncclSend / ncclRecv (every xGMI link of the root carries 1 / world of the payload per phase).  Also selected by
 * the environment (GPX_MG_BCAST=sag).  Every rank must set the same mode before the next gpx_mg_fit. */
int gpx_mg_set_bcast(gpx_mg_t *mg, int sag);
/* test hook: the next gpx_mg_fit of this rank behaves as if its factorisation had left `value` in the device
 * info word (value < 0: an internal failure, which every rank must then report as GPX_ERR_INTERNAL) */
int gpx_debug_mg_inject_info(gpx_mg_t *mg, int value);
/* test hook, host arithmetic only (no GPU needed): how panel j of an n x n problem with block width nb on `world`
 * ranks travels -- per row chunk (each its own broadcast + event) six values: first row, end row (relative to the
 * panel's first row; the panel has n + 1 - j nb rows, the rider row included), elements in the chunk, elements per
 * scatter / all-gather piece, elements in the last piece, 1 if the scatter + all-gather form is eligible.
 * Returns the number of chunks (>= 1) or a negative status. */
int gpx_debug_mg_plan(int64_t n, int64_t nb, int world, int chunks, int64_t j, int64_t *out, int cap);
int gpx_mg_create_cb(gpx_mg_t **mg, int dtype, int kernel, int64_t n, int d, int64_t nb, int world, int rank,
                     gpx_mg_bcast_fn bcast, gpx_mg_allreduce_fn allreduce, void *user);
int gpx_mg_destroy(gpx_mg_t *mg);
/* x: (n, d), y: (n,) HOST float64, the same on every rank.  NaN / infinite entries: GPX_ERR_ARG with the
 * check_finite message (as gpx_gp_fit; gpx_mg_fit checks the kernel constants the same way) */
int gpx_mg_set_data(gpx_mg_t *mg, const double *x, const double *y);
/* kernel build (owned block columns) -> distributed Cholesky -> alpha (replicated) -> log_lh with the
 * reference's conventions (gp/gp.py:360-367, gp_c.pyx:17-31); *info = first failing leading minor over
 * all ranks.  Collective: every rank calls it with the same arguments.  Synchronous. */
int gpx_mg_fit(gpx_mg_t *mg, const double *params, double s, double *log_lh, int *info);
/* posterior mean at xo (m, d) HOST float64 -> out (m,) on every rank (each evaluates a slice) */
int gpx_mg_mean(gpx_mg_t *mg, const double *params, const double *xo, int64_t m, double *out);
int gpx_mg_get_alpha(gpx_mg_t *mg, double *out);
int gpx_mg_scalars(gpx_mg_t *mg, double *logdet, double *yta, int *info);
/* this rank's times of the last fit, ms (HIP events): [0] kernel build [1] factorisation [2] solves
 * [3] reductions, and the chain inside [1]: [4] panels it factored [5] packs [6] broadcasts (incl.
 * waiting for the root) [7] trailing / column updates */
int gpx_mg_timing(gpx_mg_t *mg, double *ms8);
/* The same with the classes added in round 5 (count <= 13 values): [8] EXPOSED chain time -- how long this rank's update
 * stream sat idle in front of a panel (chunk) that had not arrived yet, summed over the fit; [9] the longest single such
 * wait; [10] the number of waits longer than 20 us; [11] rehearsal only: the modelled transfer time the fit enqueued;
 * [12] rehearsal only: the delays that stood for the remote owners' chains.
 * A hidden chain shows [8] ~ 0 whatever chain_panel / chain_bcast (kernel durations on the other streams) say. */
int gpx_mg_timing_ex(gpx_mg_t *mg, double *ms, int count);
/* ms[j] for j < count: what the owner's chain of panel j cost this rank in the last fit (factor + pack + the last chunk of
 * its column update), 0 for panels it does not own. */
int gpx_mg_chain_by_panel(gpx_mg_t *mg, double *ms, int64_t count);
/* raw device view of this rank's local matrix, (n + 1) x ld, block column j of the rank at column (j / world) nb (tests, the
 * rehearsal's check; do not free) */
int gpx_mg_device_ptrs(gpx_mg_t *mg, void **A, int64_t *ld);
/* Move the RCCL communicator of `from` (same world, rank and device; e.g. a handle of another block-column width) into `mg`,
 * which was made with gpx_mg_create_local and has none yet: ONE ncclCommInitRank per process however many layouts a run
 * tries.  `from` keeps working only as far as it needs no collective (world 1) and may be destroyed. */
int gpx_mg_adopt_comm(gpx_mg_t *mg, gpx_mg_t *from);
/* on != 0: the owner of the next panel factors it before it starts its own share of the trailing update (the panel is the
 * serial chain of the run; default: on for world >= 2, GPX_MG_OWNER_FIRST=0 / 1 overrides). */
int gpx_mg_set_owner_first(gpx_mg_t *mg, int on);
/* on != 0: also time how long the update stream waits in front of panels (chunks) that have not arrived -- values [8] .. [10]
 * of gpx_mg_timing_ex; costs two timing-enabled event records per wait on the stream that bounds the step, so it is OFF by
 * default (bench.py and the rehearsal turn it on; with it off those three values read 0). */
int gpx_mg_set_wait_timing(gpx_mg_t *mg, int on);
/* the schedule parameters IN EFFECT for the next fit (any pointer may be NULL): owner-first, row chunks per panel
 * broadcast, scatter + all-gather broadcast form, wait timing */
int gpx_mg_schedule_info(gpx_mg_t *mg, int *owner_first, int *chunks, int *sag, int *wait_timing);
/* Row chunks per panel broadcast for the following fits (1 .. 16; collective: the same on every rank). */
int gpx_mg_set_chunks(gpx_mg_t *mg, int chunks);
/* REHEARSAL: rank `rank` of a `world`-rank run in ONE process on one GPU, without a communicator.  The rank builds, factors,
 * packs and updates exactly its own block columns; a panel it does not own is copied (device to device) out of L_dev -- a
 * resident factor of the same matrix, (n + 1) x ldl with the right-hand side's row (gpx_gp_fit with the riding solve) --
 * behind a delay that stands for its owner's chain (this rank's own measurement for its nearest panel in the fit before);
 * every broadcast is followed by a delay that MODELS the transfer: bytes / link_GBps for one ring, 2 bytes / (world
 * link_GBps) for scatter + all-gather, + latency_us per collective.  alpha_dev: the solution (for the back substitution's
 * remote blocks).  Scalars of a rehearsal fit are this rank's share only.  Measured compute, modelled transfer. */
int gpx_mg_create_rehearsal(gpx_mg_t **out, int dtype, int kernel, int64_t n, int d, int64_t nb, int world, int rank,
                            const void *L_dev, int64_t ldl, const void *alpha_dev, double link_GBps, double latency_us);

/* ------------------------------------------ host-level drop-in entry points -- */
/* gaussian_c.K(out, x1, x2, h, w) -- gaussian_c.pyx:18 ; and the derivative
 * members through `member`.  out: (n, m) C-contiguous float64. */
int gpx_gaussian_c(int member, double *out, const double *x1, int64_t n,
                   const double *x2, int64_t m, double h, double w);
/* gaussian_c.jacobian(out[2,n,m], ...) :39 ; gaussian_c.hessian(out[2,2,n,m], ...) :44 */
int gpx_gaussian_c_jacobian(double *out, const double *x1, int64_t n,
                            const double *x2, int64_t m, double h, double w);
int gpx_gaussian_c_hessian(double *out, const double *x1, int64_t n,
                           const double *x2, int64_t m, double h, double w);
/* periodic_c.K(out, x1, x2, h, w, p) -- periodic_c.pyx:18 ; members as above */
int gpx_periodic_c(int member, double *out, const double *x1, int64_t n,
                   const double *x2, int64_t m, double h, double w, double p);
int gpx_periodic_c_jacobian(double *out, const double *x1, int64_t n,
                            const double *x2, int64_t m, double h, double w, double p);
int gpx_periodic_c_hessian(double *out, const double *x1, int64_t n,
                           const double *x2, int64_t m, double h, double w, double p);
/* (n, d) generalisation of the two above (BASELINE configs use d = 8/16/32).  OUT-OF-CORE: the matrix
 * is built on the device in row panels of GPX_KMAT_PANEL_BYTES (default 256 MiB) and streamed into
 * `out` with double buffering, so (n, m) may exceed HBM -- `out` can be a memory-mapped file. */
int gpx_kmat_host(int kernel, int member, double *out, const double *x1, int64_t n,
                  const double *x2, int64_t m, int d, const double *params,
                  double diag_add);

/* scipy.linalg.cholesky(A, lower=True) -- gp/gp.py:294.  A, L: (n, n) row-major
 * HOST float64 (may alias).  *info as LAPACK dpotrf. */
int gpx_cholesky(double *L, const double *A, int64_t n, int *info);
/* scipy.linalg.cho_solve((L, True), b) -- gp/gp.py:332-334.  b: (n,) in/out. */
int gpx_cho_solve(const double *L, int64_t n, double *b);
/* gp_c.log_lh(y, K, Kiy) -- gp_c.pyx:17-31.  Given L (not K): the logdet comes
 * from diag(L) instead of the reference's second LU factorisation of K. */
int gpx_gp_c_log_lh(const double *y, const double *L, const double *Kiy, int64_t n,
                    double *log_lh);
/* The derivative glue of gp/ext/gp_c.pyx for PLUGIN kernels, with the reference's own
 * arguments: every array HOST float64, C order; n = number of training points, np =
 * number of KERNEL parameters (outputs have np + 1 entries: the noise s is last);
 * Ki (n, n) = inv(Kxx), Kj (np, n, n) the kernel Jacobian, Kh (np, np, n, n) its
 * Hessian, Kiy (n,) = Ki y.  Every matrix crosses PCIe once; quadratic forms run as
 * matrix-vector work, traces of products as reductions, only the np products
 * dK_i Ki of the second-derivative trace terms as GEMMs (csrc/gpx_deriv.hip, "Glue").
 *   gpx_gp_c_dloglh_dtheta -- gp_c.pyx:34-49     dloglh (np + 1)
 *   gpx_gp_c_dlh_dtheta    -- gp_c.pyx:52-67     dlh    (np + 1)
 *   gpx_gp_c_d2lh_dtheta2  -- gp_c.pyx:70-111    d2lh   (np + 1, np + 1); dlh is an INPUT as there
 *   gpx_gp_c_dm_dtheta     -- gp_c.pyx:114-131   dm     (np + 1, m); Kjxo (np, m, n), Kxox (m, n) */
int gpx_gp_c_dloglh_dtheta(const double *y, const double *Ki, const double *Kj, const double *Kiy,
                           double s, int64_t n, int np, double *dloglh);
int gpx_gp_c_dlh_dtheta(const double *y, const double *Ki, const double *Kj, const double *Kiy,
                        double s, double lh, int64_t n, int np, double *dlh);
int gpx_gp_c_d2lh_dtheta2(const double *y, const double *Ki, const double *Kj, const double *Kh,
                          const double *Kiy, double s, double lh, const double *dlh, int64_t n,
                          int np, double *d2lh);
int gpx_gp_c_dm_dtheta(const double *y, const double *Ki, const double *Kj, const double *Kjxo,
                       const double *Kxox, double s, int64_t n, int np, int64_t m, double *dm);

/* C (M, N) = A (M, K) * B (N, K)^T, all HOST float64 C-contiguous: the np.dot
 * calls of the derivative glue (gp/ext/gp_c.pyx:43-48,61-66,89-110,127-131) on
 * the fp64 matrix cores. */
int gpx_gemm_nt_host(double *C, const double *A, const double *B, int64_t M, int64_t N,
                     int64_t K);

#ifdef __cplusplus
}
#endif
#endif /* GPX_H */
