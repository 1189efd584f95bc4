// gpx_io.hip -- persistence of a fitted GP and the out-of-core kernel-matrix build (gfx950).
//
// The reference persists a GP by pickling its memoised host arrays (gp/gp.py:78-92: `_memoized` holds
// Kxx, Lxx, ... as numpy arrays).  Here the factor lives in HBM (32 GiB at n = 65536): pickling it
// would need a full host copy.  Two streaming paths replace that:
//   * gpx_gp_save / gpx_gp_load: a checkpoint file of the fitted handle -- header, x, y, alpha and the
//     LOWER trapezoid of L in row blocks, moved through two pinned staging buffers (device -> host copy
//     of block k+1 overlaps the file write of block k); host memory use is two blocks, whatever n is.
//   * gpx_kmat_host (the host-pointer entry behind gaussian_c.K / periodic_c.K and their derivative
//     members): the matrix is built in row panels on the device and streamed to the caller's buffer
//     through the same double buffering, so an (n, m) result larger than HBM -- e.g. a numpy.memmap --
//     is produced with two panels of device memory.
#include "gpx_gp_internal.h"
#include <stdio.h>
#include <string>
#include <unistd.h>
#include <fcntl.h>
#include <errno.h>
#include <time.h>
#include <atomic>
#include <stdlib.h>
#include <sys/stat.h>
#include <cmath>
#include <vector>

extern "C" int gpx_gp_create(gpx_gp_t **out, int dtype, int kernel, int64_t n, int d);
extern "C" int gpx_gp_destroy(gpx_gp_t *g);

namespace gpx {

struct FactorHeader {
    char magic[8];                 // "GPXFACT1"
    int32_t version, dtype, kernel, d, nparams, info;
    int64_t n, block_rows;
    double params[3], s, logdet, yta;
};

struct Pinned {
    void *p = nullptr;
    ~Pinned() { if (p) (void)hipHostFree(p); }
    int alloc(size_t bytes)
    {
        hipError_t e = hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; return hip_fail(e, "hipHostMalloc", __FILE__, __LINE__); }
        return GPX_OK;
    }
};

struct File {
    FILE *f = nullptr;
    ~File() { if (f) fclose(f); }
};

// two events for the double-buffered staging, released on every way out
struct EventPair {
    hipEvent_t e[2] = {nullptr, nullptr};
    ~EventPair() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
    int create()
    {
        GPX_HIP(hipEventCreateWithFlags(&e[0], hipEventDisableTiming));
        GPX_HIP(hipEventCreateWithFlags(&e[1], hipEventDisableTiming));
        return GPX_OK;
    }
};

// bytes of a checkpoint with this header: header, x, y, alpha as float64, then the lower trapezoid in row blocks
static bool expected_file_bytes(const FactorHeader &hd, size_t es, unsigned long long *out)
{
    const unsigned long long n = (unsigned long long)hd.n, d = (unsigned long long)hd.d, R = (unsigned long long)hd.block_rows;
    unsigned long long total = sizeof(FactorHeader) + 8ull * (n * d + 2 * n);
    for (unsigned long long r0 = 0; r0 < n; r0 += R) {
        const unsigned long long r1 = std::min(n, r0 + R);
        total += (r1 - r0) * r1 * es;
    }
    *out = total;
    return true;
}

static int64_t io_block_rows(int64_t n, size_t es)
{
    const size_t target = (size_t)std::max<int64_t>(1, tune().io_block_bytes);
    int64_t r = (int64_t)(target / ((size_t)std::max<int64_t>(n, 1) * es));
    return std::max<int64_t>(1, std::min<int64_t>(r, n));
}

template <typename T>
__global__ void vec_to_f64(const T *__restrict__ src, double *__restrict__ dst, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (double)src[i];
}
template <typename T>
__global__ void vec_from_f64(const double *__restrict__ src, T *__restrict__ dst, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (T)src[i];
}

static int vec_d2h_f64(gpx_gp *g, const void *dev, int64_t count, std::vector<double> &out)
{
    out.resize((size_t)count);
    DevBuf tmp;
    GPX_TRY(tmp.alloc((size_t)count * 8));
    const unsigned nb = (unsigned)cdiv(count, 256);
    if (g->dtype == GPX_F64) hipLaunchKernelGGL((vec_to_f64<double>), dim3(nb), dim3(256), 0, g->st, (const double *)dev, (double *)tmp.p, count);
    else hipLaunchKernelGGL((vec_to_f64<float>), dim3(nb), dim3(256), 0, g->st, (const float *)dev, (double *)tmp.p, count);
    GPX_LAUNCH_CHECK();
    GPX_HIP(hipMemcpyAsync(out.data(), tmp.p, (size_t)count * 8, hipMemcpyDeviceToHost, g->st));
    GPX_HIP(hipStreamSynchronize(g->st));
    return GPX_OK;
}

static int vec_h2d_f64(gpx_gp *g, void *dev, int64_t count, const std::vector<double> &in)
{
    DevBuf tmp;
    GPX_TRY(tmp.alloc((size_t)count * 8));
    GPX_HIP(hipMemcpyAsync(tmp.p, in.data(), (size_t)count * 8, hipMemcpyHostToDevice, g->st));
    const unsigned nb = (unsigned)cdiv(count, 256);
    if (g->dtype == GPX_F64) hipLaunchKernelGGL((vec_from_f64<double>), dim3(nb), dim3(256), 0, g->st, (const double *)tmp.p, (double *)dev, count);
    else hipLaunchKernelGGL((vec_from_f64<float>), dim3(nb), dim3(256), 0, g->st, (const double *)tmp.p, (float *)dev, count);
    GPX_LAUNCH_CHECK();
    GPX_HIP(hipStreamSynchronize(g->st));
    return GPX_OK;
}

}  // namespace gpx

using namespace gpx;

extern "C" {

int gpx_gp_save(gpx_gp_t *g, const char *path)
{
    GP_ENTER(g);
    GPX_ARG(path && g->fitted, "gp is not fitted / path is NULL");
    const int64_t n = g->n, lda = g->lda;
    const size_t es = esize(g->dtype);
    double h4[4];
    GPX_HIP(hipMemcpyAsync(h4, g->scal, sizeof(h4), hipMemcpyDeviceToHost, g->st));
    GPX_HIP(hipStreamSynchronize(g->st));
    FactorHeader hd;
    memset(&hd, 0, sizeof(hd));
    memcpy(hd.magic, "GPXFACT1", 8);
    hd.version = 1; hd.dtype = g->dtype; hd.kernel = g->kernel; hd.d = g->d;
    hd.nparams = g->have_params ? g->nparams : 0;          // 0: fitted from an uploaded matrix (plugin kernel)
    memcpy(&hd.info, &h4[3], sizeof(int));
    hd.n = n; hd.block_rows = io_block_rows(n, es);
    for (int i = 0; i < 3; ++i) hd.params[i] = g->have_params ? g->params[i] : 0.0;
    hd.s = g->s; hd.logdet = h4[0]; hd.yta = h4[1];
    // written under a temporary name and renamed when complete: a failure never leaves a truncated checkpoint
    // under the final name
    // (a UNIQUE temporary in the target's directory so that two saves to the same path, e.g. every rank of a multi-rank
    //  job checkpointing, never write or unlink each other's file; the last rename wins, each is complete.  Created with
    //  open(O_CREAT | O_EXCL, 0666) under a random name: the KERNEL applies the caller's umask, as fopen would -- round 5
    //  read the umask with umask(0) / umask(old) to fix up mkstemp's 0600, which is process-wide and not thread-safe)
    std::string tmp_path;
    int tfd = -1;
    {
        static std::atomic<unsigned> seq{0};
        struct timespec ts;
        (void)clock_gettime(CLOCK_MONOTONIC, &ts);
        unsigned long long r = (unsigned long long)ts.tv_nsec ^ ((unsigned long long)getpid() << 20) ^ (unsigned long long)(uintptr_t)g;
        for (int attempt = 0; attempt < 64 && tfd < 0; ++attempt) {
            r = r * 6364136223846793005ull + 1442695040888963407ull + seq.fetch_add(1);
            char suffix[24];
            snprintf(suffix, sizeof(suffix), ".%012llx.tmp", r >> 16);
            tmp_path = std::string(path) + suffix;
            tfd = open(tmp_path.c_str(), O_CREAT | O_EXCL | O_WRONLY | O_CLOEXEC, 0666);
            if (tfd < 0 && errno != EEXIST) break;
        }
    }
    if (tfd < 0) { set_error("gpx_gp_save: cannot create a temporary file beside %s", path); return GPX_ERR_ARG; }
    struct Unlink { const std::string &p; bool armed = true; ~Unlink() { if (armed) (void)unlink(p.c_str()); } } cleanup{tmp_path};
    File fp;
    fp.f = fdopen(tfd, "wb");
    if (!fp.f) { (void)close(tfd); set_error("gpx_gp_save: cannot open %s for writing", tmp_path.c_str()); return GPX_ERR_ARG; }
    bool ok = fwrite(&hd, sizeof(hd), 1, fp.f) == 1;
    std::vector<double> v;
    GPX_TRY(vec_d2h_f64(g, g->x, n * g->d, v)); ok = ok && fwrite(v.data(), 8, v.size(), fp.f) == v.size();
    GPX_TRY(vec_d2h_f64(g, g->y, n, v));        ok = ok && fwrite(v.data(), 8, v.size(), fp.f) == v.size();
    GPX_TRY(vec_d2h_f64(g, g->alpha, n, v));    ok = ok && fwrite(v.data(), 8, v.size(), fp.f) == v.size();
    // the lower trapezoid of L, row block [r0, r1): columns [0, r1), packed row-major, two staging buffers
    const int64_t R = hd.block_rows;
    Pinned stage[2];
    EventPair evp;
    GPX_TRY(stage[0].alloc((size_t)R * n * es));
    GPX_TRY(stage[1].alloc((size_t)R * n * es));
    GPX_TRY(evp.create());
    hipEvent_t *ev = evp.e;
    const int64_t nblk = cdiv(n, R);
    auto issue = [&](int64_t b) -> int {
        const int64_t r0 = b * R, r1 = std::min(n, r0 + R);
        GPX_HIP(hipMemcpy2DAsync(stage[b & 1].p, (size_t)r1 * es, (const char *)g->A + (size_t)r0 * lda * es,
                                 (size_t)lda * es, (size_t)r1 * es, (size_t)(r1 - r0), hipMemcpyDeviceToHost, g->st));
        GPX_HIP(hipEventRecord(ev[b & 1], g->st));
        return GPX_OK;
    };
    int rc = issue(0);
    for (int64_t b = 0; b < nblk && rc == GPX_OK && ok; ++b) {
        if (b + 1 < nblk) rc = issue(b + 1);                        // in flight while block b is written
        if (hipEventSynchronize(ev[b & 1]) != hipSuccess) { rc = GPX_ERR_HIP; break; }
        const int64_t r0 = b * R, r1 = std::min(n, r0 + R);
        char *buf = (char *)stage[b & 1].p;
        for (int64_t r = r0; r < r1; ++r)                           // strict upper part of the diagonal block: zeros
            memset(buf + ((size_t)(r - r0) * r1 + (size_t)r + 1) * es, 0, (size_t)(r1 - r - 1) * es);
        ok = fwrite(buf, es, (size_t)(r1 - r0) * r1, fp.f) == (size_t)(r1 - r0) * r1;
    }
    (void)hipStreamSynchronize(g->st);
    if (rc != GPX_OK) return rc;
    if (!ok || fflush(fp.f) != 0 || fsync(fileno(fp.f)) != 0) { set_error("gpx_gp_save: short write to %s", tmp_path.c_str()); return GPX_ERR_ARG; }
    {
        FILE *f = fp.f; fp.f = nullptr;
        if (fclose(f) != 0) { set_error("gpx_gp_save: closing %s failed", tmp_path.c_str()); return GPX_ERR_ARG; }
    }
    if (rename(tmp_path.c_str(), path) != 0) { set_error("gpx_gp_save: cannot rename %s to %s", tmp_path.c_str(), path); return GPX_ERR_ARG; }
    cleanup.armed = false;
    // the rename itself is only durable once the directory entry is: fsync the directory (best effort)
    {
        std::string dir(path);
        const size_t slash = dir.find_last_of('/');
        dir = slash == std::string::npos ? std::string(".") : (slash == 0 ? std::string("/") : dir.substr(0, slash));
        const int dfd = open(dir.c_str(), O_RDONLY | O_DIRECTORY);
        if (dfd >= 0) { (void)fsync(dfd); (void)close(dfd); }
    }
    return GPX_OK;
}

int gpx_gp_load(gpx_gp_t **out, const char *path)
{
    GPX_TRY(ensure_device());
    GPX_ARG(out && path, "NULL argument");
    *out = nullptr;
    File fp;
    fp.f = fopen(path, "rb");
    if (!fp.f) { set_error("gpx_gp_load: cannot open %s", path); return GPX_ERR_ARG; }
    FactorHeader hd;
    if (fread(&hd, sizeof(hd), 1, fp.f) != 1 || memcmp(hd.magic, "GPXFACT1", 8) != 0 || hd.version != 1) {
        set_error("gpx_gp_load: %s is not a gpx factor checkpoint", path);
        return GPX_ERR_ARG;
    }
    // the header is not trusted: every field is checked, and the sizes it implies against the size of the file,
    // before anything is allocated from it
    const bool dtype_ok = hd.dtype == GPX_F64 || hd.dtype == GPX_F32;
    const bool kernel_ok = hd.kernel == GPX_KERNEL_GAUSSIAN || hd.kernel == GPX_KERNEL_PERIODIC;
    const int np_expected = hd.kernel == GPX_KERNEL_PERIODIC ? 3 : 2;
    bool fields_ok = dtype_ok && kernel_ok && hd.n >= 1 && hd.n <= ((int64_t)1 << 24) && hd.d >= 1 && hd.d <= 65536 &&
                     hd.block_rows >= 1 && hd.block_rows <= hd.n && (hd.nparams == 0 || hd.nparams == np_expected) &&
                     hd.info >= 0 && std::isfinite(hd.s) && hd.s >= 0;
    for (int i = 0; i < hd.nparams && fields_ok; ++i) fields_ok = std::isfinite(hd.params[i]);
    if (!fields_ok) { set_error("gpx_gp_load: %s has a corrupt header", path); return GPX_ERR_ARG; }
    {
        struct stat sb;
        unsigned long long want = 0;
        expected_file_bytes(hd, hd.dtype == GPX_F64 ? 8 : 4, &want);
        if (fstat(fileno(fp.f), &sb) != 0 || (unsigned long long)sb.st_size != want) {
            set_error("gpx_gp_load: %s is truncated or its header is corrupt (%llu bytes expected)", path, want);
            return GPX_ERR_ARG;
        }
    }
    gpx_gp_t *g = nullptr;
    GPX_TRY(gpx_gp_create(&g, hd.dtype, hd.kernel, hd.n, hd.d));
    struct Guard { gpx_gp_t *g; ~Guard() { if (g) gpx_gp_destroy(g); } } guard{g};
    const int64_t n = g->n, lda = g->lda;
    const size_t es = esize(g->dtype);
    std::vector<double> v;
    auto rd = [&](void *dev, int64_t count) -> int {
        v.resize((size_t)count);
        if (fread(v.data(), 8, v.size(), fp.f) != v.size()) { set_error("gpx_gp_load: %s is truncated", path); return GPX_ERR_ARG; }
        return vec_h2d_f64(g, dev, count, v);
    };
    GPX_TRY(rd(g->x, n * g->d));
    GPX_TRY(rd(g->y, n));
    GPX_TRY(rd(g->alpha, n));
    const int64_t R = hd.block_rows;
    Pinned stage[2];
    EventPair evp;
    GPX_TRY(stage[0].alloc((size_t)R * n * es));
    GPX_TRY(stage[1].alloc((size_t)R * n * es));
    GPX_TRY(evp.create());
    hipEvent_t *ev = evp.e;
    const int64_t nblk = cdiv(n, R);
    int rc = GPX_OK;
    bool used[2] = {false, false};
    for (int64_t b = 0; b < nblk && rc == GPX_OK; ++b) {
        const int64_t r0 = b * R, r1 = std::min(n, r0 + R);
        if (used[b & 1] && hipEventSynchronize(ev[b & 1]) != hipSuccess) { rc = GPX_ERR_HIP; break; }   // its upload is done
        if (fread(stage[b & 1].p, es, (size_t)(r1 - r0) * r1, fp.f) != (size_t)(r1 - r0) * r1) {
            set_error("gpx_gp_load: %s is truncated", path);
            rc = GPX_ERR_ARG;
            break;
        }
        if (hipMemcpy2DAsync((char *)g->A + (size_t)r0 * lda * es, (size_t)lda * es, stage[b & 1].p, (size_t)r1 * es,
                             (size_t)r1 * es, (size_t)(r1 - r0), hipMemcpyHostToDevice, g->st) != hipSuccess ||
            hipEventRecord(ev[b & 1], g->st) != hipSuccess) { rc = GPX_ERR_HIP; break; }
        used[b & 1] = true;
    }
    (void)hipStreamSynchronize(g->st);
    if (rc != GPX_OK) return rc;
    GPX_TRY(gp_scan_finite(g));
    double h4[4] = {hd.logdet, hd.yta, 0.0, 0.0};
    memcpy(&h4[3], &hd.info, sizeof(int));
    GPX_HIP(hipMemcpy(g->scal, h4, sizeof(h4), hipMemcpyHostToDevice));
    for (int i = 0; i < 3; ++i) g->params[i] = hd.params[i];
    g->s = hd.s;
    g->have_data = true; g->have_params = hd.nparams > 0; g->fitted = true; g->have_K = false;
    g->ops.invalidate();
    guard.g = nullptr;
    *out = g;
    return GPX_OK;
}

int gpx_gp_describe(gpx_gp_t *g, int *dtype, int *kernel, int64_t *n, int *d, double *params3, double *s)
{
    GP_ENTER(g);
    if (dtype) *dtype = g->dtype;
    if (kernel) *kernel = g->kernel;
    if (n) *n = g->n;
    if (d) *d = g->d;
    if (params3) for (int i = 0; i < 3; ++i) params3[i] = g->params[i];
    if (s) *s = g->s;
    return GPX_OK;
}

int gpx_gp_get_xy(gpx_gp_t *g, double *x, double *y)
{
    GP_ENTER(g);
    GPX_ARG(g->have_data, "no data in the handle");
    std::vector<double> v;
    if (x) { GPX_TRY(vec_d2h_f64(g, g->x, g->n * g->d, v)); memcpy(x, v.data(), v.size() * 8); }
    if (y) { GPX_TRY(vec_d2h_f64(g, g->y, g->n, v)); memcpy(y, v.data(), v.size() * 8); }
    return GPX_OK;
}

// ---- out-of-core kernel-matrix build: row panels, double buffered -------------------------------
int gpx_kmat_host(int kernel, int member, double *out, const double *x1, int64_t n, const double *x2,
                  int64_t m, int d, const double *params, double diag_add)
{
    GPX_TRY(ensure_device());
    GPX_ARG(n >= 0 && m >= 0 && d >= 1, "bad dimensions");
    if (n == 0 || m == 0) return GPX_OK;
    GPX_ARG(out && x1 && x2 && params, "NULL pointer");
    const int64_t ld = round_up(m, 16);
    // panel height: GPX_KMAT_PANEL_BYTES of device memory per panel (default 256 MiB), a multiple of 64 rows
    const size_t target = (size_t)std::max<int64_t>(1, tune().kmat_panel_bytes);
    int64_t R = (int64_t)(target / ((size_t)ld * 8));
    R = std::max<int64_t>(64, R / 64 * 64);
    R = std::min<int64_t>(R, round_up(n, 64));
    DevBuf a, b, o[2];
    GPX_TRY(a.alloc((size_t)n * d * 8));
    GPX_HIP(hipMemcpy(a.p, x1, (size_t)n * d * 8, hipMemcpyHostToDevice));
    const void *bp = a.p;
    if (!(x2 == x1 && m == n)) {
        GPX_TRY(b.alloc((size_t)m * d * 8));
        GPX_HIP(hipMemcpy(b.p, x2, (size_t)m * d * 8, hipMemcpyHostToDevice));
        bp = b.p;
    }
    const int64_t npan = cdiv(n, R);
    GPX_TRY(o[0].alloc((size_t)R * ld * 8));
    if (npan > 1) GPX_TRY(o[1].alloc((size_t)R * ld * 8));
    hipStream_t st[2] = {nullptr, nullptr};
    GPX_HIP(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking));
    if (npan > 1) GPX_HIP(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
    int rc = GPX_OK;
    // panel p is built and copied out on stream p & 1: the build of panel p + 1 (other stream, other buffer)
    // overlaps the copy of panel p; a buffer is reused only after its own stream has drained
    // (the copy into pageable host memory blocks the calling thread, so the NEXT panel's build is
    //  enqueued first; diag_add belongs to global (i, i) and is applied on the host at the end)
    auto build = [&](int64_t p) -> int {
        const int64_t r0 = p * R, rows = std::min(R, n - r0);
        return gpx_d_kmat(GPX_F64, kernel, member, (const double *)a.p + r0 * d, rows, bp, m, d, params, 0.0, GPX_FULL,
                          o[p & 1].p, ld, (void *)st[p & 1]);
    };
    rc = build(0);
    for (int64_t p = 0; p < npan && rc == GPX_OK; ++p) {
        const int64_t r0 = p * R, rows = std::min(R, n - r0);
        if (p + 1 < npan) { rc = build(p + 1); if (rc != GPX_OK) break; }   // buffer (p+1)&1: its copy (panel p-1) is done
        if (hipMemcpy2DAsync(out + (size_t)r0 * m, (size_t)m * 8, o[p & 1].p, (size_t)ld * 8, (size_t)m * 8, (size_t)rows,
                             hipMemcpyDeviceToHost, st[p & 1]) != hipSuccess ||
            hipStreamSynchronize(st[p & 1]) != hipSuccess) { rc = GPX_ERR_HIP; break; }
    }
    stream_epoch_bump();
    for (int i = 0; i < 2; ++i) if (st[i]) { (void)hipStreamSynchronize(st[i]); (void)hipStreamDestroy(st[i]); }
    if (rc == GPX_ERR_HIP) { set_error("gpx_kmat_host: HIP failure while streaming panels"); return rc; }
    if (rc != GPX_OK) return rc;
    if (diag_add != 0.0) {
        const int64_t k = std::min(n, m);
        for (int64_t i = 0; i < k; ++i) out[(size_t)i * m + i] += diag_add;
    }
    return GPX_OK;
}

}  // extern "C"
