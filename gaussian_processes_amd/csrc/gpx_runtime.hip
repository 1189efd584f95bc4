// gpx_runtime.hip -- device / memory / stream / event plumbing of the C ABI.
#include "gpx_common.h"
#include <stdarg.h>
#include <stddef.h>
#include <dlfcn.h>
#include <atomic>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

namespace gpx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    if (e == hipErrorOutOfMemory) return GPX_ERR_NOMEM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return GPX_ERR_NO_DEVICE;
    return GPX_ERR_HIP;
}

int ensure_device()
{
    tune_refresh();                                            // (every gpx_d_* / host entry comes through here: gpx_tune.h)
    static thread_local int ok = 0;
    if (ok) return GPX_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no usable HIP device (hipGetDeviceCount -> %d, count %d): libgpx has no CPU fallback",
                  (int)e, n);
        return GPX_ERR_NO_DEVICE;
    }
    ok = 1;
    return GPX_OK;
}


}  // namespace gpx

// ---- switches: one snapshot per API call (gpx_tune.h) -------------------------------------------------------------
extern "C" char **environ;
namespace gpx {
static thread_local Tune g_tune;
static std::atomic<long long> g_tune_refreshes{0};
const Tune &tune() { return g_tune; }
int64_t tune_refresh_count() { return (int64_t)g_tune_refreshes.load(std::memory_order_relaxed); }

namespace {
enum TuneKind { TK_I64, TK_FLAG, TK_I64_SET, TK_I64_2 };
struct TuneEntry { const char *name; size_t len; int kind; size_t off, off2; };
#define GPX_E_X(f, e, d) {e, sizeof(e) - 1, TK_I64, offsetof(Tune, f), 0},
#define GPX_E_XF(f, e) {e, sizeof(e) - 1, TK_FLAG, offsetof(Tune, f), 0},
#define GPX_E_XS(f, e, d) {e, sizeof(e) - 1, TK_I64_SET, offsetof(Tune, f), offsetof(Tune, f##_set)},
#define GPX_E_X2(f, e, d0, d1) {e, sizeof(e) - 1, TK_I64_2, offsetof(Tune, f), 0},
const TuneEntry g_tune_table[] = { GPX_TUNE_LIST(GPX_E_X, GPX_E_XF, GPX_E_XS, GPX_E_X2) };
#undef GPX_E_X
#undef GPX_E_XF
#undef GPX_E_XS
#undef GPX_E_X2
}  // namespace

void tune_refresh()
{
    Tune t;                                                    // defaults
    for (char **e = environ; e && *e; ++e) {
        const char *kv = *e;
        if (kv[0] != 'G' || kv[1] != 'P' || kv[2] != 'X' || kv[3] != '_') continue;
        const char *eq = strchr(kv, '=');
        if (!eq || eq[1] == 0) continue;                       // (empty: as if unset -- the convention of env_i64 / env_set)
        const size_t len = (size_t)(eq - kv);
        const char *val = eq + 1;
        if (len == 16 && memcmp(kv, "GPX_POTRF_WIDTHS", 16) == 0) {
            long long a, b, c;
            if (sscanf(val, "%lld,%lld,%lld", &a, &b, &c) == 3) { t.potrf_widths[0] = a; t.potrf_widths[1] = b; t.potrf_widths[2] = c; }
            continue;
        }
        if (len == 12 && memcmp(kv, "GPX_MG_BCAST", 12) == 0) { t.mg_bcast_set = true; t.mg_bcast_sag = strcmp(val, "sag") == 0; continue; }
        if (len == 12 && memcmp(kv, "GPX_RCCL_LIB", 12) == 0) { snprintf(t.rccl_lib, sizeof(t.rccl_lib), "%s", val); continue; }
        for (const TuneEntry &en : g_tune_table) {
            if (en.len != len || memcmp(en.name, kv, len) != 0) continue;
            char *base = (char *)&t;
            switch (en.kind) {
            case TK_FLAG: *(bool *)(base + en.off) = true; break;
            case TK_I64: *(int64_t *)(base + en.off) = (int64_t)atoll(val); break;
            case TK_I64_SET: *(int64_t *)(base + en.off) = (int64_t)atoll(val); *(bool *)(base + en.off2) = true; break;
            case TK_I64_2: ((int64_t *)(base + en.off))[0] = ((int64_t *)(base + en.off))[1] = (int64_t)atoll(val); break;
            }
            break;
        }
    }
    g_tune = t;
    g_tune_refreshes.fetch_add(1, std::memory_order_relaxed);
}
}  // namespace gpx

extern "C" int gpx_debug_tune_refreshes(int64_t *count)
{
    if (!count) return GPX_ERR_ARG;
    *count = gpx::tune_refresh_count();
    return GPX_OK;
}

namespace gpx {
// ---- profiling registry -----------------------------------------------------
// One registry for all host threads (mlii drives one handle per thread): every access is
// under g_prof_mu, a scope ends the record it began (by index), and the registry is capped so a
// forgotten gpx_prof_enable(1) cannot grow without bound.
bool g_prof_on = false;
thread_local int g_prof_mute = 0;
struct ProfRec { int cls; double work; hipEvent_t a, b; };
static std::vector<ProfRec> g_prof;
static std::mutex g_prof_mu;
static const size_t PROF_CAP = 1u << 20;

int prof_begin(int cls, double work, hipStream_t st)
{
    ProfRec r;
    r.cls = cls; r.work = work; r.a = nullptr; r.b = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (g_prof.size() >= PROF_CAP) return -1;
    }
    if (hipEventCreate(&r.a) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (hipEventCreate(&r.b) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(r.a); return -1; }
    (void)hipEventRecord(r.a, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}

void prof_end(int rec, hipStream_t st)
{
    hipEvent_t b = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (rec >= 0 && (size_t)rec < g_prof.size()) b = g_prof[rec].b;   // cleared meanwhile: nothing to end
    }
    if (b) (void)hipEventRecord(b, st);
}

static void prof_clear()
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : g_prof) {
        if (r.a) (void)hipEventDestroy(r.a);
        if (r.b) (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
}

// ---- route counters ----------------------------------------------------------
static std::atomic<long long> g_route[RT_COUNT];
void route_hit(int route) { if (route >= 0 && route < RT_COUNT) g_route[route].fetch_add(1, std::memory_order_relaxed); }

// ---- per-(kernel, device) launch attributes -------------------------------------
static std::mutex g_attr_mu;
static std::set<std::pair<const void *, int>> g_attr_done;

int set_max_lds(const void *fn, int bytes)
{
    int dev = 0;
    GPX_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_attr_mu);
    const auto key = std::make_pair(fn, dev);
    if (g_attr_done.count(key)) return GPX_OK;
    GPX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    g_attr_done.insert(key);
    return GPX_OK;
}

}  // namespace gpx

using namespace gpx;

// ---- roctx ranges (gpx_common.h) ----
namespace gpx {
static std::atomic<long long> g_roctx_ranges{0};
struct RoctxLib { int (*push)(const char *) = nullptr; int (*pop)() = nullptr; bool tried = false; };
static RoctxLib g_roctx;
static std::mutex g_roctx_mu;
const char *prof_class_name(int cls)
{
    static const char *names[PC_COUNT] = {"gpx:kernel_matrix", "gpx:gemm_trailing_update", "gpx:potrf_panel", "gpx:trsm_rows", "gpx:trsv",
                                          "gpx:mean", "gpx:reduce", "gpx:gemm_skinny", "gpx:gemm_generic", "gpx:gemm_panel", "gpx:gemm_n64"};
    return (cls >= 0 && cls < PC_COUNT) ? names[cls] : "gpx:other";
}
bool roctx_push(const char *name)
{
    if (!tune().roctx) return false;
    {
        std::lock_guard<std::mutex> lk(g_roctx_mu);
        if (!g_roctx.tried) {
            g_roctx.tried = true;
            // (rocprofv3 intercepts the rocprofiler-sdk flavour; the roctracer one is the fall-back for older tools)
            const char *libs[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                                  "libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"};
            for (const char *l : libs) {
                void *h = dlopen(l, RTLD_NOW | RTLD_GLOBAL);
                if (!h) continue;
                g_roctx.push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
                g_roctx.pop = (int (*)())dlsym(h, "roctxRangePop");
                if (g_roctx.push && g_roctx.pop) break;
                g_roctx.push = nullptr; g_roctx.pop = nullptr;
            }
        }
    }
    if (!g_roctx.push) return false;
    (void)g_roctx.push(name);
    g_roctx_ranges.fetch_add(1, std::memory_order_relaxed);
    return true;
}
void roctx_pop() { if (g_roctx.pop) (void)g_roctx.pop(); }
}  // namespace gpx

extern "C" int gpx_debug_roctx_ranges(int64_t *count)
{
    if (!count) return GPX_ERR_ARG;
    *count = (int64_t)gpx::g_roctx_ranges.load(std::memory_order_relaxed);
    return GPX_OK;
}

// ---- StreamTurn (gpx_common.h): one turn at a time for the streams a host thread drives ----
namespace gpx {
// `last` is compared together with the stream EPOCH at which it was recorded: the library bumps the epoch whenever it
// destroys a stream (gpx_stream_destroy, a handle's own streams), so a new stream that happens to get a destroyed one's
// address is not mistaken for it.  (Streams the caller creates and destroys with HIP directly are the caller's to order.)
std::atomic<unsigned long long> g_stream_epoch{1};
void stream_epoch_bump() { g_stream_epoch.fetch_add(1, std::memory_order_relaxed); }
struct TurnState { hipEvent_t ev = nullptr; hipStream_t last = nullptr; bool have = false; unsigned long long epoch = 0; };
static thread_local TurnState g_turn[16];                      // per device

static TurnState *turn_state()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return (dev >= 0 && dev < 16) ? &g_turn[dev] : nullptr;
}
static bool capturing(hipStream_t st)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}
StreamTurn::StreamTurn(hipStream_t s) : st(s)
{
    TurnState *t = turn_state();
    if (!t || !t->have || (t->last == st && t->epoch == g_stream_epoch.load(std::memory_order_relaxed)) || capturing(st)) return;
    if (hipStreamWaitEvent(st, t->ev, 0) != hipSuccess) (void)hipGetLastError();
}
StreamTurn::~StreamTurn()
{
    TurnState *t = turn_state();
    if (!t || capturing(st)) return;
    if (!t->ev && hipEventCreateWithFlags(&t->ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); t->ev = nullptr; return; }
    if (hipEventRecord(t->ev, st) == hipSuccess) { t->last = st; t->have = true; t->epoch = g_stream_epoch.load(std::memory_order_relaxed); }
    else (void)hipGetLastError();
}
}  // namespace gpx

extern "C" {

int gpx_version(void) { return GPX_VERSION; }

int gpx_debug_route_count(int route, int64_t *count)
{
    GPX_ARG(route >= 0 && route < RT_COUNT && count, "bad route / NULL count");
    *count = (int64_t)g_route[route].load(std::memory_order_relaxed);
    return GPX_OK;
}

int gpx_debug_route_reset(void)
{
    for (int i = 0; i < RT_COUNT; ++i) g_route[i].store(0, std::memory_order_relaxed);
    return GPX_OK;
}

const char *gpx_last_error(void) { return g_err; }

int gpx_device_count(int *count)
{
    GPX_ARG(count, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return GPX_OK;
}

int gpx_set_device(int device)
{
    GPX_TRY(ensure_device());
    GPX_HIP(hipSetDevice(device));
    return GPX_OK;
}

int gpx_get_device(int *device)
{
    GPX_ARG(device, "device is NULL");
    GPX_TRY(ensure_device());
    GPX_HIP(hipGetDevice(device));
    return GPX_OK;
}

int gpx_device_info(int device, char *name, size_t name_len, int *cus, int *clock_mhz,
                    uint64_t *hbm_bytes)
{
    GPX_TRY(ensure_device());
    hipDeviceProp_t p;
    GPX_HIP(hipGetDeviceProperties(&p, device));
    if (name && name_len) {
        snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    if (cus) *cus = p.multiProcessorCount;
    if (clock_mhz) *clock_mhz = p.clockRate / 1000;
    if (hbm_bytes) *hbm_bytes = (uint64_t)p.totalGlobalMem;
    return GPX_OK;
}

int gpx_malloc(void **dptr, size_t bytes)
{
    GPX_ARG(dptr, "dptr is NULL");
    GPX_TRY(ensure_device());
    *dptr = nullptr;
    if (bytes == 0) bytes = 16;
    GPX_HIP(hipMalloc(dptr, bytes));
    return GPX_OK;
}

int gpx_free(void *dptr)
{
    if (!dptr) return GPX_OK;
    GPX_HIP(hipFree(dptr));
    return GPX_OK;
}

int gpx_mem_info(size_t *free_bytes, size_t *total_bytes)
{
    GPX_TRY(ensure_device());
    size_t f = 0, t = 0;
    GPX_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return GPX_OK;
}

int gpx_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
    GPX_TRY(ensure_device());
    if (bytes == 0) return GPX_OK;
    GPX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, S(stream)));
    GPX_HIP(hipStreamSynchronize(S(stream)));
    return GPX_OK;
}

int gpx_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
    GPX_TRY(ensure_device());
    if (bytes == 0) return GPX_OK;
    GPX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, S(stream)));
    GPX_HIP(hipStreamSynchronize(S(stream)));
    return GPX_OK;
}

int gpx_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream)
{
    GPX_TRY(ensure_device());
    if (bytes == 0) return GPX_OK;
    GPX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, S(stream)));
    return GPX_OK;
}

int gpx_memcpy2d_h2d(void *dst, size_t dpitch, const void *src, size_t spitch,
                     size_t row_bytes, size_t rows, void *stream)
{
    GPX_TRY(ensure_device());
    if (row_bytes == 0 || rows == 0) return GPX_OK;
    GPX_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows, hipMemcpyHostToDevice, S(stream)));
    GPX_HIP(hipStreamSynchronize(S(stream)));
    return GPX_OK;
}

int gpx_memcpy2d_d2h(void *dst, size_t dpitch, const void *src, size_t spitch,
                     size_t row_bytes, size_t rows, void *stream)
{
    GPX_TRY(ensure_device());
    if (row_bytes == 0 || rows == 0) return GPX_OK;
    GPX_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows, hipMemcpyDeviceToHost, S(stream)));
    GPX_HIP(hipStreamSynchronize(S(stream)));
    return GPX_OK;
}

int gpx_memset(void *dst, int value, size_t bytes, void *stream)
{
    GPX_TRY(ensure_device());
    if (bytes == 0) return GPX_OK;
    GPX_HIP(hipMemsetAsync(dst, value, bytes, S(stream)));
    return GPX_OK;
}

int gpx_stream_create(void **stream)
{
    GPX_ARG(stream, "stream is NULL");
    GPX_TRY(ensure_device());
    hipStream_t s;
    GPX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return GPX_OK;
}

int gpx_stream_destroy(void *stream)
{
    if (!stream) return GPX_OK;
    gpx::stream_epoch_bump();
    GPX_HIP(hipStreamDestroy(S(stream)));
    return GPX_OK;
}

int gpx_stream_sync(void *stream)
{
    GPX_TRY(ensure_device());
    GPX_HIP(hipStreamSynchronize(S(stream)));
    return GPX_OK;
}

int gpx_device_sync(void)
{
    GPX_TRY(ensure_device());
    GPX_HIP(hipDeviceSynchronize());
    return GPX_OK;
}

int gpx_event_create(void **event)
{
    GPX_ARG(event, "event is NULL");
    GPX_TRY(ensure_device());
    hipEvent_t e;
    GPX_HIP(hipEventCreate(&e));
    *event = (void *)e;
    return GPX_OK;
}

int gpx_event_destroy(void *event)
{
    if (!event) return GPX_OK;
    GPX_HIP(hipEventDestroy((hipEvent_t)event));
    return GPX_OK;
}

int gpx_event_record(void *event, void *stream)
{
    GPX_HIP(hipEventRecord((hipEvent_t)event, S(stream)));
    return GPX_OK;
}

int gpx_event_sync(void *event)
{
    GPX_HIP(hipEventSynchronize((hipEvent_t)event));
    return GPX_OK;
}

int gpx_event_elapsed_ms(void *start, void *stop, float *ms)
{
    GPX_ARG(ms, "ms is NULL");
    GPX_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return GPX_OK;
}

int gpx_prof_enable(int on)
{
    if (g_prof_on || on) GPX_TRY(ensure_device());
    if (g_prof_on) (void)hipDeviceSynchronize();
    prof_clear();
    g_prof_on = on != 0;
    return GPX_OK;
}

int gpx_prof_read(int cls, double *launches, double *total_ms, double *total_work)
{
    GPX_ARG(cls >= 0 && cls < PC_COUNT, "unknown profile class");
    GPX_TRY(ensure_device());
    GPX_HIP(hipDeviceSynchronize());
    double n = 0, ms = 0, w = 0;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : g_prof) {
        if (r.cls != cls || !r.a || !r.b) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) { (void)hipGetLastError(); continue; }
        n += 1; ms += t; w += r.work;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (total_work) *total_work = w;
    return GPX_OK;
}

int gpx_stream_wait_event(void *stream, void *event)
{
    GPX_HIP(hipStreamWaitEvent(S(stream), (hipEvent_t)event, 0));
    return GPX_OK;
}

}  // extern "C"
