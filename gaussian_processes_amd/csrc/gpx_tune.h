// gpx_tune.h -- every behaviour switch of libgpx in ONE table.
//
// The switches are environment variables (GPX_*).  Until round 4 each was a getenv at its point of use -- a dozen per
// panel launch on the factorisation's critical path, 66 sites in all.  Now the calling thread takes ONE snapshot per API
// call: every extern "C" entry refreshes it (tune_refresh: one pass over `environ` that only looks at entries starting
// with "GPX_" -- no getenv, ~0.2 us with none set), and everything below reads fields of that snapshot.  A test or a
// tuning script that sets a variable between two API calls still gets the route it asked for (the route counters
// assert it); nothing changes route in the middle of a fit.
//
// Thresholds that depend on the arithmetic live here too, keyed by dtype (GPX_F64 = 0, GPX_F32 = 1): X2 entries.
#pragma once
#include <stdint.h>

namespace gpx {

//   X(field, "ENV", default)            int64 value; unset or empty -> default
//   XF(field, "ENV")                    flag: true when set and non-empty
//   XS(field, "ENV", default)           int64 value + field##_set (was the variable given at all?)
//   X2(field, "ENV", dflt_f64, dflt_f32) int64 value per dtype: field[dtype]; the variable overrides both
#define GPX_TUNE_LIST(X, XF, XS, X2)                                                                                   \
    /* ---- MFMA GEMM (gpx_gemm.hip) ---- */                                                                           \
    XF(gemm_no_fast, "GPX_GEMM_NO_FAST")                                                                               \
    X(gemm_ablate, "GPX_GEMM_ABLATE", 0)                                                                               \
    X(gemm_pad_lds, "GPX_GEMM_PAD_LDS", 0)                                                                             \
    X(gemm_bn64_tiles, "GPX_GEMM_BN64_TILES", 256)                                                                     \
    X(gemm_exact, "GPX_GEMM_EXACT", 1)                                                                                 \
    X(gemm_fine_tiles, "GPX_GEMM_FINE_TILES", 16384)                                                                   \
    X2(syrk_bn64_tiles, "GPX_SYRK_BN64_TILES", 2600, 4000)                                                             \
    /* ---- fit / batch / io (gpx_gp.hip, gpx_io.hip) ---- */                                                          \
    X(fit_ride_max, "GPX_FIT_RIDE_MAX", 16384)                                                                         \
    X(fit_ops_ahead, "GPX_FIT_OPS_AHEAD", 1)                                                                           \
    X(fit_ops_ahead_min, "GPX_FIT_OPS_AHEAD_MIN", 8192)                                                                \
    X(fit_ops_tail, "GPX_FIT_OPS_TAIL", 4)                                                                             \
    XS(batch_max, "GPX_BATCH_MAX", 1)                                                                                  \
    X(io_block_bytes, "GPX_IO_BLOCK_BYTES", (int64_t)64 << 20)                                                         \
    X(kmat_panel_bytes, "GPX_KMAT_PANEL_BYTES", (int64_t)256 << 20)                                                    \
    /* ---- multi-GPU schedule (gpx_mg.hip) ---- */                                                                    \
    XF(force_collectives, "GPX_FORCE_COLLECTIVES")                                                                     \
    X(mg_bcast_chunks, "GPX_MG_BCAST_CHUNKS", 4)                                                                       \
    XF(mg_no_timing, "GPX_MG_NO_TIMING")                                                                               \
    XS(mg_owner_first, "GPX_MG_OWNER_FIRST", 0)                                                                        \
    /* ---- resident panel kernel (gpx_panel.hip) ---- */                                                              \
    XF(trace, "GPX_TRACE")                                                                                             \
    X(potrf_res, "GPX_POTRF_RES", 256)                                                                                 \
    X(res_strict, "GPX_RES_STRICT", 1)                                                                                 \
    X(two_part_rows, "GPX_POTRF_TWO_PART_ROWS", 16384)                                                                 \
    X(two_part_batch, "GPX_POTRF_TWO_PART_BATCH", 98304)                                                               \
    X(leaf_mfma_f32_rows, "GPX_LEAF_MFMA_F32_ROWS", 16384)                                                             \
    X(panel_excl_rows, "GPX_PANEL_EXCL_ROWS", 5120)                                                                    \
    XS(leaf, "GPX_LEAF", 0)                                                                                            \
    X2(leaf4_rows, "GPX_LEAF4_ROWS", 8192, 5120)                                                                       \
    X2(panel_pad_lds, "GPX_PANEL_PAD_LDS", 8 * 1024, 40 * 1024)                                                        \
    X(fold_rows, "GPX_POTRF_FOLD_ROWS", 16384)                                                                         \
    X(fold_k, "GPX_POTRF_FOLD_K", 256)                                                                                 \
    /* ---- blocked factorisation (gpx_potrf.hip) ---- */                                                              \
    XS(potrf_nb, "GPX_POTRF_NB", 0)                                                                                    \
    XF(potrf_trsm_rows, "GPX_POTRF_TRSM_ROWS")                                                                         \
    X(potrf_inv_max, "GPX_POTRF_INV_MAX", 16384)                                                                       \
    XF(no_lookahead, "GPX_POTRF_NO_LOOKAHEAD")                                                                         \
    X(taper, "GPX_POTRF_TAPER", 1)                                                                                     \
    X2(pair_rows, "GPX_POTRF_PAIR_ROWS", 0, 0)                                                                         \
    X(host_paced, "GPX_POTRF_HOST_PACED", 16384)                                                                       \
    X(gate_rows, "GPX_POTRF_GATE_ROWS", 16384)                                                                         \
    /* ---- solves (gpx_solve.hip) ---- */                                                                             \
    X(trsv_ops, "GPX_TRSV_OPS", 1)                                                                                     \
    X(trsv_ops_min, "GPX_TRSV_OPS_MIN", 10240)                                                                         \
    X(trsm_ops, "GPX_TRSM_OPS", 1)                                                                                     \
    /* ---- diagnostics (gpx_runtime.hip) ---- */                                                                      \
    XF(roctx, "GPX_ROCTX")

struct Tune {
#define GPX_T_X(f, e, d) int64_t f = (d);
#define GPX_T_XF(f, e) bool f = false;
#define GPX_T_XS(f, e, d) int64_t f = (d); bool f##_set = false;
#define GPX_T_X2(f, e, d0, d1) int64_t f[2] = {(d0), (d1)};
    GPX_TUNE_LIST(GPX_T_X, GPX_T_XF, GPX_T_XS, GPX_T_X2)
#undef GPX_T_X
#undef GPX_T_XF
#undef GPX_T_XS
#undef GPX_T_X2
    // strings and lists
    long long potrf_widths[3] = {1, 8192, 12288};      // GPX_POTRF_WIDTHS = "rows128,rows256,rows512": outer-block taper
    bool mg_bcast_set = false, mg_bcast_sag = false;   // GPX_MG_BCAST ("sag": scatter + all-gather)
    char rccl_lib[256] = {0};                          // GPX_RCCL_LIB
};

// the calling host thread's snapshot (defaults until its first refresh)
const Tune &tune();
// re-read the environment into the calling thread's snapshot: every extern "C" entry does this once (ensure_device /
// GP_ENTER / MG_ENTER); nothing below an entry reads the environment
void tune_refresh();
// how often the calling process has looked a GPX_* variable up since start (tests: a fit adds refreshes, never lookups
// per launch) -- gpx_debug_tune_refreshes
int64_t tune_refresh_count();

}  // namespace gpx
