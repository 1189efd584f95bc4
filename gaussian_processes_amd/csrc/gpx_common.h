// gpx_common.h -- shared internals of libgpx.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "../../include/gpx.h"
#include "gpx_tune.h"

namespace gpx {

void set_error(const char *fmt, ...);
int  hip_fail(hipError_t e, const char *what, const char *file, int line);
int  ensure_device();   // GPX_OK when a GPU is usable

#define GPX_HIP(call)                                                        \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess) return gpx::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

#define GPX_TRY(call)                                                        \
    do {                                                                     \
        int rc__ = (call);                                                   \
        if (rc__ != GPX_OK) return rc__;                                     \
    } while (0)

#define GPX_ARG(cond, msg)                                                   \
    do {                                                                     \
        if (!(cond)) { gpx::set_error("%s: %s", __func__, msg); return GPX_ERR_ARG; } \
    } while (0)

#define GPX_LAUNCH_CHECK()                                                   \
    do {                                                                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) return gpx::hip_fail(e__, "kernel launch", __FILE__, __LINE__); \
    } while (0)

// Environment switches: ONE table, gpx_tune.h -- every extern "C" entry takes one snapshot of the
// GPX_* variables for the calling thread (tune_refresh), everything below reads tune().field.  Nothing in the library
// calls getenv.

// Route counters (gpx_debug_route_count): which of the alternative routes a call took, counted on the host at the
// point of decision.  Tests that force a route through an environment switch assert it here.
enum Route { RT_TRSV_OPS = 0, RT_TRSV_STEPS = 1, RT_PANEL_RES = 2, RT_PANEL_CHAIN = 3, RT_FIT_RIDE = 4,
             RT_FIT_TWO_SOLVES = 5, RT_GEMM_FAST = 6, RT_GEMM_GENERIC = 7, RT_SYRK_EXACT = 8, RT_SYRK_PATCH = 9,
             RT_MG_BCAST_ONE = 10, RT_MG_BCAST_SAG = 11, RT_FIT_OPS_AHEAD = 12, RT_TRSM_OPS = 13, RT_POTRF_PAIR = 14, RT_COUNT = 15 };
void route_hit(int route);

// LAPACK-style info of a factorisation as the host sees it: > 0 "not positive definite" (the caller's business),
// < 0 an INTERNAL failure of the factorisation (a hand-off inside the resident panel kernel timed out: -7) --
// never to be mistaken for a property of the matrix.  Returns GPX_ERR_INTERNAL for the latter.
int check_internal_info(int info);

static inline hipStream_t S(void *s) { return (hipStream_t)s; }
static inline size_t esize(int dtype) { return dtype == GPX_F64 ? 8 : 4; }
static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- live per-kernel-class timing (HIP events around each launch; off by default) ----
enum ProfClass { PC_KMAT = 0, PC_GEMM = 1, PC_POTRF_DIAG = 2, PC_TRSM_ROWS = 3, PC_TRSV = 4,
                 PC_MEAN = 5, PC_REDUCE = 6, PC_GEMM_SKINNY = 7, PC_GEMM_GENERIC = 8, PC_GEMM_PANEL = 9, PC_GEMM_N64 = 10,
                 PC_COUNT = 11 };
extern bool g_prof_on;
// the registry is shared by all host threads (mutex inside); a scope ends its OWN record
int  prof_begin(int cls, double work, hipStream_t st);    // record index, -1 when nothing was recorded
void prof_end(int rec, hipStream_t st);
// roctx ranges (GPX_ROCTX=1; libroctx64.so by dlopen): every gpx_gp_* call and every launch class below it is a nested host
// range, so `rocprofv3 --marker-trace --kernel-trace` shows which stage of a fit a kernel belongs to.  Off: one field of the snapshot per scope.
bool roctx_push(const char *name);          // false: ranges are off (or the library is not there): nothing to pop
void roctx_pop();
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char *name) : on(roctx_push(name)) {}
    ~RoctxRange() { if (on) roctx_pop(); }
};
const char *prof_class_name(int cls);

// launches the calling thread makes while this is > 0 are not recorded (the asm leaf's self-check: its nested launches
// are no part of anybody's fit)
extern thread_local int g_prof_mute;
struct ProfScope {
    hipStream_t st; int rec; RoctxRange range;
    ProfScope(int cls, double work, hipStream_t s) : st(s), rec(g_prof_on && g_prof_mute == 0 ? prof_begin(cls, work, s) : -1), range(prof_class_name(cls)) {}
    ~ProfScope() { if (rec >= 0) prof_end(rec, st); }
};

// hipFuncAttributeMaxDynamicSharedMemorySize, once per (kernel, device); thread-safe
int set_max_lds(const void *fn, int bytes);

// Makes `device` current for the calling host thread and restores the previous one on scope
// exit (HIP's current device is per host thread; handles remember the device they live on).
struct DeviceGuard {
    int prev = -1; bool switched = false; int rc = GPX_OK;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (device >= 0 && device != prev) {
            hipError_t e = hipSetDevice(device);
            if (e != hipSuccess) rc = hip_fail(e, "hipSetDevice", __FILE__, __LINE__);
            else switched = true;
        }
    }
    ~DeviceGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
};

// Everything a host thread enqueues through this library shares that thread's grow-only scratch buffers (block inverses and
// operators of the solves, the panels' hand-off blocks, reduction scratch ...).  Calls on ONE stream are ordered by the stream;
// a call on ANOTHER stream first waits for the work of the call before it: handles fitted asynchronously back to back from
// one thread (each on its own stream) would otherwise race on those buffers (seen once as a wrong log_lh in the four-handle
// test of tests/test_gpu_configs.py, when faster panels changed the overlap).  One event record per API call; nothing is
// recorded or waited for while a stream is being captured.
void stream_epoch_bump();          // call before destroying any stream (see TurnState, gpx_runtime.hip)
struct StreamTurn {
    hipStream_t st;
    explicit StreamTurn(hipStream_t s);
    ~StreamTurn();
};

// parameters of a kernel-matrix member, precomputed on the host in f64
struct KParams {
    double c[6];     // member-specific constants (see gpx_kmat.hip)
    double diag_add;
    int kernel;
    int member;
};
int make_kparams(int kernel, int member, const double *params, double diag_add, KParams *out);

// Batched launches: `count` matrices of identical shape; element strides between consecutive
// matrices for the A / B / C operands of a product (one stride for everything else).
// Optional second batch dimension (count2 > 1; launch grid z): matrix (i, i2) sits at i * s + i2 * t.
struct Batch {
    int count; int64_t sA, sB, sC;
    int count2 = 1; int64_t tA = 0, tB = 0, tC = 0;
};

// internal (non-ABI) helpers shared between translation units
// C = beta * C + alpha * A * B^T with beta = 1 (default) or 0 (beta0 != 0: C is not read)
int gemm_nt(int dtype, int64_t M, int64_t N, int64_t K, const void *A, int64_t lda, const void *B,
            int64_t ldb, void *C, int64_t ldc, double alpha, int tri, int64_t row0, int64_t col0,
            hipStream_t st, int beta0 = 0, int ktri = 0, const Batch *bt = nullptr,
            int wide_tiles = 0);   // wide_tiles: 128-wide tiles whatever the tile count (N <= 128: ONE tile per row block,
                                   // which is what makes C == A legal -- a tile reads its rows of A before it stores them)
// (ktri != 0: A == B is upper triangular in (row, k) and M == N == K -- the k-loop of tile row i skips k < i)
// X[r, 0:jb] <- X[r, 0:jb] * Ljj^-T for rows r in [0, rows); Ljj = jb x jb lower block
int trsm_rows(int dtype, void *X, int64_t ldx, int64_t rows, const void *Ljj, int64_t ldl, int jb,
              hipStream_t st, const Batch *bt = nullptr);   // bt: sA = stride of X, sB = stride of Ljj
int syrk_bc(int dtype, int64_t n, int64_t row_begin, void *Cloc, int64_t ldc, int64_t cl0, int64_t cl1,
            const void *Pb, int64_t ldp, int64_t k0, int64_t kb, int64_t nb, int P, int rank,
            hipStream_t st, const int *abort_flag = nullptr, const Batch *bt = nullptr);
// factor rows [r0, n) x columns [c0, c0 + kb) of A whose diagonal block sits at (r0, c0)
// done: an event recorded behind the panel's last launch
int potrf_panel(int dtype, void *A, int64_t lda, int64_t n, int64_t r0, int64_t c0, int64_t kb,
                int *info_dev, hipStream_t st, const Batch *bt = nullptr, int64_t kpre = 0,   // kpre: see potrf_panel_res
                hipEvent_t done = nullptr);
int potrf(int dtype, void *A, int64_t n, int64_t lda, int *info_dev, hipStream_t st, const Batch *bt = nullptr,
          int64_t xrows = 0,       // xrows: extra rows below the matrix that ride along (A has n + xrows rows)
          bool may_block = false); // may_block: the caller allows the host to pace the panel launches (hipEventSynchronize
                                   // inside the call); false: a pure enqueue (gpx_d_potrf, anything under stream capture)
// the same panel in ONE launch (gpx_panel.hip): kb a multiple of 64, at most panel_res_max()
int potrf_panel_res(int dtype, void *A, int64_t lda, int64_t n, int64_t r0, int64_t c0, int64_t kb, int *info_dev,
                    hipStream_t st, const Batch *bt = nullptr, int64_t kpre = 0, hipEvent_t done = nullptr);
int64_t panel_res_max();
// set by potrf()'s loop for the one panel launch that will find the chip idle; reading it clears it (gpx_potrf.hip)
bool potrf_take_idle_chip_hint();
// this host thread's look-ahead stream of the blocked factorisation on the current device (nullptr before the first
// one): it lives as long as the thread, so an event may be recorded on it at any time
hipStream_t potrf_side_stream();
// would K(x, x) + s^2 I hold only finite numbers for finite x?  (gpx_gp.hip; the check_finite of the reference's cho_factor)
bool kernel_values_finite(int kernel, const double *p, double s, int dtype);   // in the handle's arithmetic
bool panel_res_fold(int64_t rows, int64_t kpre, int64_t kb, size_t es, int64_t lda, const void *base);
// Per-factor block operators of the single-right-hand-side solves (gpx_solve.hip, "operator form"): owned by
// whoever owns the factor; `valid` must be cleared whenever the factor changes.  nullptr: built per call.
struct TrsvOps {
    void *buf = nullptr; size_t bytes = 0;
    bool valid = false;          // all blocks of the CURRENT factor have their operators
    int64_t built = 0;           // leading 512-blocks of the current factor that have them (trsv_ops_build_upto)
    void invalidate() { valid = false; built = 0; }   // a new factor: every owner calls this, never `valid = false` alone
};
// Build the operators of an n x n factor (n a multiple of 512) ahead of time on `st`; a later trsv_lower with these ops
// takes the operator route whatever n is (the distributed solve prepares each diagonal block right after its panel).
int trsv_ops_build(int dtype, const void *L, int64_t n, int64_t ldl, TrsvOps *ops, hipStream_t st);
// The same in instalments, while the factorisation is still running: the operators of the 512-blocks [ops->built, kend) --
// which need nothing but block columns < kend of L -- on `st`; ops->valid once kend reaches n / 512.  n % 512 == 0 only
// (trsv_ops_ahead_ok); the first call of a factor passes ops->built == 0.
bool trsv_ops_ahead_ok(int dtype, const void *L, int64_t n, int64_t ldl);
size_t trsv_ops_bytes(int dtype, int64_t n);
int trsv_ops_build_upto(int dtype, const void *L, int64_t n, int64_t ldl, TrsvOps *ops, int64_t kend, hipStream_t st);
// potrf() progress hook of the calling host thread (null: none): called with the number of leading columns that are final
// once `panel_done` has fired; single matrices with more than one outer block only.
struct PotrfHook { int (*fn)(void *user, int64_t cols_done, hipEvent_t panel_done); void *user; };
void potrf_set_hook(const PotrfHook *hook);
int trsv_lower(int dtype, const void *L, int64_t n, int64_t ldl, void *b, void *x, int transpose,
               hipStream_t st, const Batch *bt = nullptr,   // bt: sA = stride of L, sB = stride of b / x
               TrsvOps *ops = nullptr);
int trsm_right_lt(int dtype, const void *L, int64_t n, int64_t ldl, void *X, int64_t m, int64_t ldx,
                  hipStream_t st, int x_upper = 0, TrsvOps *ops = nullptr);   // ops: this factor's block operators (completed here if need be): in-block substitution = one product with inv(L_kk)
int trsm_right_lt_batch(int dtype, const void *L, int64_t sL, int64_t n, int64_t ldl, void *X, int64_t sX, int64_t m, int64_t ldx,
                        hipStream_t st, int x_upper, const void *ops_base, int64_t sO, int count);   // `count` systems in lock-step (operator route)
int logdet_chol(int dtype, const void *L, int64_t n, int64_t ldl, double *out_dev, hipStream_t st, int count = 1,
                int64_t sL = 0, int64_t so = 0);
int dot(int dtype, const void *a, const void *b, int64_t n, double *out_dev, hipStream_t st, int count = 1,
        int64_t sa = 0, int64_t sb = 0, int64_t so = 0);
int tril(int dtype, void *A, int64_t n, int64_t lda, hipStream_t st);
int dloglh_reduce(int dtype, int kernel, const void *x, int64_t n, int d, const double *params,
                  const void *alpha, const void *W, int64_t ldw, double *partial_dev, double *out4,
                  hipStream_t st);
int kmat(int dtype, int kernel, int member, const void *x1, int64_t n, const void *x2, int64_t m,
         int d, const double *params, double diag_add, int tri, void *out, int64_t ld, hipStream_t st);

}  // namespace gpx
