// gpx_gp_internal.h -- the fitted-GP handle, shared by gpx_gp.hip and gpx_deriv.hip (not part of the ABI).
#pragma once
#include "gpx_common.h"

struct gpx_gp {
    int device;        // the HIP device the handle lives on; every entry point makes it current
    int dtype, kernel, d, nparams;
    int64_t n, lda;
    void *x, *y, *A, *alpha, *t0, *t1;
    double *scal;      // device: [0] logdet [1] y^T alpha [2] spare ; int info at scal + 3
    hipStream_t st;
    hipEvent_t ev[6];
    double params[3];
    double s;
    bool have_data, have_params, fitted, have_K;
    bool x_finite, y_finite;   // scipy's check_finite=True (gp/gp.py:294, 332-334): one O(n d) device reduction per set_data
    float ms[5];
    // fit_batch workspace (grow-only, freed with the handle): the matrices of one chunk + their vectors
    void *bw; size_t bw_bytes; int64_t bw_cap;
    // block operators of the triangular solves (built once per factor, reused by every later solve)
    gpx::TrsvOps ops;
    // fit_batch_grad: X = L^-T and W = K^-1 of one matrix at a time + the reduction's partial sums (grow-only), and
    // the block operators of the row being differentiated
    void *gw; size_t gw_bytes; int64_t gw_cap;   // gw_cap: rows per lock-step gradient group the block holds
    gpx::TrsvOps bops;
    hipStream_t st_ops;   // lazily created: where gpx_gp_fit builds `ops` while the factorisation is still running
    hipEvent_t ev_ops;
};

namespace gpx {

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes)
    {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e != hipSuccess) { p = nullptr; return hip_fail(e, "hipMalloc", __FILE__, __LINE__); }
        return GPX_OK;
    }
};

}  // namespace gpx

namespace gpx {
// x_finite / y_finite of the handle from its device arrays (one O(n d) reduction; synchronous)
int gp_scan_finite(gpx_gp *g);
}

// every gpx_gp_* entry: the handle's device becomes current for the duration of the call, and the handle's stream takes
// its turn among the streams this host thread drives (StreamTurn, gpx_common.h); with GPX_ROCTX=1 the call is a roctx range
#define GP_ENTER(g)                                                          \
    GPX_ARG((g) != nullptr, "gp is NULL");                                   \
    gpx::tune_refresh();                                                     \
    gpx::DeviceGuard guard__((g)->device);                                   \
    if (guard__.rc != GPX_OK) return guard__.rc;                             \
    gpx::StreamTurn turn__((g)->st);                                         \
    gpx::RoctxRange api_range__(__func__)

