// gpx_mg.hip -- multi-GPU GP fit / predict behind the C ABI: one process per GPU, the factorisation's
// critical path (panel -> pack -> broadcast -> column update) driven from C on HIP streams with RCCL
// collectives over xGMI (the reference has nothing distributed).
//
// Layout: 1-D block-cyclic block columns of width nb -- global block column j lives on rank j % P as local
// block j / P (row-major local matrix: (n + 1) rows x ceil(nblk / P) * nb columns; row n carries y, see below).
// Per panel k (right-looking, one-panel look-ahead, two panel buffers):
//   owner(k)    factors its block column below the diagonal (potrf_panel) on the panel stream Q,
//               packs it into a contiguous (n + 1 - k0) x nb buffer (pack kernel),
//   all ranks   receive it by a broadcast rooted at the owner, issued in ROW CHUNKS on the broadcast stream B
//               (one collective per chunk, or scatter + all-gather by grouped send / recv): the owner of
//               panel k+1 applies chunk c to its block column k+1 (main stream S) as soon as chunk c has
//               landed, so that its column update hides under the tail of the transfer,
//   all ranks   update their remaining block columns with the whole panel (one launch, syrk_bc) on S
//               while Q already factors and B broadcasts panel k+1.
// Solve: y rides along as row n of every block column (forward substitution happens inside the factorisation);
// the backward substitution is right-looking over the block columns, one nb-vector broadcast per block.
// logdet and the posterior mean are local sums plus one all-reduce.
//
// Communicator back-ends: (1) RCCL, loaded with dlopen at first use -- libgpx.so itself has no link-time
// dependency on it, single-GPU users never load it; ranks are joined through an ncclUniqueId that the
// host distributes out of band (gpx_mg_unique_id on rank 0).  (2) host callbacks on DEVICE pointers
// (gpx_mg_create_cb): the same C schedule over any transport -- the tests run it with several ranks on
// ONE GPU over gloo, which RCCL cannot do.
#include "gpx_common.h"
#include <dlfcn.h>
#include <climits>
#include <cmath>
#include <vector>

// The handful of RCCL declarations this file needs, restated from the public nccl.h ABI (stable since NCCL 2.x):
// libgpx.so builds on a ROCm install without the RCCL headers, as it already runs without the library.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                 // ncclSuccess = 0
typedef int ncclDataType_t;               // ncclInt32 = 2, ncclFloat32 = 7, ncclFloat64 = 8
typedef int ncclRedOp_t;                  // ncclSum = 0, ncclMax = 2
}
static constexpr ncclResult_t ncclSuccess = 0;
static constexpr ncclDataType_t ncclInt32 = 2, ncclFloat32 = 7, ncclFloat64 = 8;
static constexpr ncclRedOp_t ncclSum = 0, ncclMax = 2;
static_assert(sizeof(ncclUniqueId) == GPX_MG_ID_BYTES, "ncclUniqueId is 128 bytes");

extern "C" int gpx_d_mean(int dtype, int kernel, const void *xo, int64_t m, const void *x, int64_t n, int d,
                          const double *params, const void *alpha, void *out, void *stream);
namespace gpx {

// ---- RCCL through dlopen ---------------------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;

static int rccl_load()
{
    if (g_rccl.lib) return GPX_OK;
    const char *names[] = {tune().rccl_lib[0] ? tune().rccl_lib : nullptr, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) {
        if (!nm || !*nm) continue;
        h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) { set_error("cannot load RCCL (librccl.so.1): %s", dlerror()); return GPX_ERR_UNSUPPORTED; }
#define GPX_SYM(field, name)                                                                  \
    *(void **)(&g_rccl.field) = dlsym(h, name);                                               \
    if (!g_rccl.field) { set_error("RCCL symbol %s not found", name); dlclose(h); return GPX_ERR_UNSUPPORTED; }
    GPX_SYM(GetUniqueId, "ncclGetUniqueId");
    GPX_SYM(CommInitRank, "ncclCommInitRank");
    GPX_SYM(CommDestroy, "ncclCommDestroy");
    GPX_SYM(Broadcast, "ncclBroadcast");
    GPX_SYM(AllReduce, "ncclAllReduce");
    GPX_SYM(Send, "ncclSend");
    GPX_SYM(Recv, "ncclRecv");
    GPX_SYM(GroupStart, "ncclGroupStart");
    GPX_SYM(GroupEnd, "ncclGroupEnd");
    GPX_SYM(CommCount, "ncclCommCount");
    GPX_SYM(CommUserRank, "ncclCommUserRank");
    GPX_SYM(CommCuDevice, "ncclCommCuDevice");
    GPX_SYM(GetErrorString, "ncclGetErrorString");
#undef GPX_SYM
    g_rccl.lib = h;
    return GPX_OK;
}

#define GPX_NCCL(call)                                                                              \
    do {                                                                                            \
        ncclResult_t r__ = (call);                                                                  \
        if (r__ != ncclSuccess) {                                                                   \
            set_error("RCCL error %d (%s) in %s", (int)r__, g_rccl.GetErrorString(r__), #call);     \
            return GPX_ERR_HIP;                                                                     \
        }                                                                                           \
    } while (0)

// contiguous copy of a factored block column: rows x kb of A (ld) -> buf (ld nb), 16 bytes per lane
template <typename T>
__global__ __launch_bounds__(256) void pack_panel_kernel(const T *__restrict__ A, int64_t lda, T *__restrict__ buf,
                                                         int64_t nb, int64_t rows, int kb)
{
    constexpr int V = 16 / sizeof(T);
    const int chunks = (kb + V - 1) / V;                  // 16-byte chunks per row (kb % 16 == 0 except ragged last block)
    const int64_t total = rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / chunks;
        const int c = (int)(i - r * chunks) * V;
        if (c + V <= kb) {
            struct alignas(16) Q { T e[V]; };
            *reinterpret_cast<Q *>(buf + r * nb + c) = *reinterpret_cast<const Q *>(A + r * lda + c);
        } else {
            for (int e = c; e < kb; ++e) buf[r * nb + e] = A[r * lda + e];
        }
    }
}

// y[c] -= sum_{r < rows} A[r, c] * x[r]  for c < ncols: a block of `rows` (<= 1024) rows of this rank's local matrix
// against one finished block of alpha -- the right-looking step of the distributed back substitution.  64 columns per
// workgroup (one per lane: coalesced rows), the rows split over the four waves, partial sums combined in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void rowblock_gemv_t_kernel(const T *__restrict__ A, int64_t ld, int rows, int64_t ncols,
                                                              const T *__restrict__ x, T *__restrict__ y)
{
    __shared__ T sx[1024];
    __shared__ double part[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < rows; i += 256) sx[i] = x[i];
    __syncthreads();
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    const int per = (rows + 3) / 4, ra = wave * per, rb = min(rows, ra + per);
    double acc = 0.0;
    if (c < ncols) {
        const T *col = A + c;
        int r = ra;
        for (; r + 8 <= rb; r += 8) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = col[(int64_t)(r + u) * ld];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fma((double)v[u], (double)sx[r + u], acc);
        }
        for (; r < rb; ++r) acc = fma((double)col[(int64_t)r * ld], (double)sx[r], acc);
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < ncols) y[c] = (T)((double)y[c] - (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]));
}

template <typename T>
__global__ void axpy_slot_kernel(double *__restrict__ acc, const double *__restrict__ v)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) acc[0] += v[0];
}

__global__ void info_key_kernel(const int *__restrict__ info, int *__restrict__ key)
{
    // LAPACK info over all ranks = the SMALLEST positive one: max-reduce of (BIG - info) for info > 0.  A negative
    // info (an internal failure of a resident panel launch on some rank) outranks everything: INT_MAX.
    if (threadIdx.x == 0 && blockIdx.x == 0) key[0] = info[0] < 0 ? INT_MAX : (info[0] > 0 ? (1 << 30) - info[0] : 0);
}

}  // namespace gpx

using namespace gpx;

constexpr int GPX_MG_TIMING_N = 13;
typedef int (*gpx_mg_bcast_fn)(void *user, void *dev_ptr, size_t bytes, int root, void *stream);
typedef int (*gpx_mg_allreduce_fn)(void *user, void *dev_ptr, size_t count, int dtype, int op, void *stream);

struct gpx_mg {
    int device, dtype, kernel, d, world, rank;
    int64_t n, nr, nb, nblk, ncols_local, ld;    // nr = n + 1: y rides along as row n of the local matrix (the forward solve)
    size_t es;
    std::vector<int64_t> my_blocks;
    std::vector<gpx::TrsvOps> ops;                // per owned block column: inverse of its diagonal block (backward solve)
    // communicator
    ncclComm_t comm = nullptr;
    gpx_mg_bcast_fn cb_bcast = nullptr;
    gpx_mg_allreduce_fn cb_allreduce = nullptr;
    void *cb_user = nullptr;
    int bcast_chunks = 4;
    int bcast_sag = 0;                            // panel broadcast as scatter + all-gather (GPX_MG_BCAST=sag / gpx_mg_set_bcast)
    int owner_first = 0;                          // the owner of the next panel factors it BEFORE it starts its own trailing update
                                                  // (gpx_mg_set_owner_first / GPX_MG_OWNER_FIRST; see mg_factor)
    hipEvent_t last_panel_ev = nullptr;           // behind the last panel this rank factored (stream Q)
    void *sag_tmp = nullptr; size_t sag_tmp_bytes = 0;   // callback back-end only: where a rank drops pieces that are not its own
    int debug_info = 0;                           // gpx_debug_mg_inject_info: written into the device info word after the factorisation
    // device state
    void *A = nullptr, *pbuf[2] = {nullptr, nullptr}, *x = nullptr, *y = nullptr, *alpha = nullptr, *tmp = nullptr;
    double *scal = nullptr;                       // scal: [0] logdet block [1] y^T alpha [2] logdet acc [3] spare
    int *info = nullptr;                          // [0] info [1] reduction key
    hipStream_t S = nullptr, Q = nullptr, B = nullptr;   // main (updates, solves) / panel (factor, pack) / panel broadcasts
    hipStream_t O = nullptr;                      // lowest priority: the diagonal blocks' solve operators, off the panel chain
    std::vector<hipEvent_t> ev;                   // sync events (no timing), reused round-robin per fit
    size_t ev_next = 0;
    std::vector<hipEvent_t> tev;                  // timing events (pairs)
    std::vector<int> tcls;                        // class of each timing pair
    std::vector<int64_t> tpan;                    // the panel it belongs to (-1: none)
    std::vector<double> own_chain_ms;             // per owned panel j of the LAST fit: factor + pack + the last chunk's share of its column update
    size_t tev_next = 0;
    bool have_data = false, fitted = false;
    double logdet = 0, yta = 0;
    int info_host = 0;
    double ms[GPX_MG_TIMING_N] = {0};            // 0 .. 7: stages and chain (gpx_mg_timing); 8 .. 10: exposed waits (gpx_mg_timing_ex)
    bool timing = true;
    bool wait_timing = false;                     // T_WAIT pairs around the update stream's waits (gpx_mg_set_wait_timing): opt-in
    // REHEARSAL of one rank's share of a `world`-rank run in ONE process (gpx_mg_create_rehearsal): no communicator; a
    // panel this rank does not own is copied out of a resident factor of the same matrix (reh_L: n + 1 rows, the rider row
    // included), and every broadcast is followed by a delay that models the transfer at reh_GBps per xGMI link
    bool rehearse = false;
    const void *reh_L = nullptr; int64_t reh_ld = 0;
    const void *reh_alpha = nullptr;
    double reh_GBps = 100.0, reh_lat_us = 20.0;
    double reh_model_ms = 0;                      // modelled transfer time enqueued by the last fit (all panel broadcasts)
    double reh_chain_ms = 0;                      // modelled remote-owner chain time enqueued by the last fit

    int64_t owner(int64_t j) const { return j % world; }
    int64_t local_col(int64_t j) const { return (j / world) * nb; }
    int64_t k0(int64_t j) const { return j * nb; }
    int64_t kb(int64_t j) const { return std::min(nb, n - j * nb); }
    int64_t first_local_block_after(int64_t k) const
    {
        for (size_t jl = 0; jl < my_blocks.size(); ++jl) if (my_blocks[jl] > k) return (int64_t)jl;
        return -1;
    }
    char *Aat(int64_t r, int64_t c) const { return (char *)A + ((size_t)r * ld + c) * es; }
};

namespace gpx {

enum { T_BUILD = 0, T_FACTOR = 1, T_SOLVE = 2, T_REDUCE = 3, T_PANEL = 4, T_PACK = 5, T_BCAST = 6, T_UPDATE = 7,
       T_WAIT = 8 };   // T_WAIT: the update stream sat idle in front of a panel (chunk) that had not arrived: EXPOSED chain time

static int mg_event(gpx_mg *g, hipEvent_t *e)
{
    if (g->ev_next == g->ev.size()) {
        hipEvent_t x;
        GPX_HIP(hipEventCreateWithFlags(&x, hipEventDisableTiming));
        g->ev.push_back(x);
    }
    *e = g->ev[g->ev_next++];
    return GPX_OK;
}

// record a "now" on `to` after everything enqueued so far on `from`
static int mg_order(gpx_mg *g, hipStream_t from, hipStream_t to)
{
    hipEvent_t e;
    GPX_TRY(mg_event(g, &e));
    GPX_HIP(hipEventRecord(e, from));
    GPX_HIP(hipStreamWaitEvent(to, e, 0));
    return GPX_OK;
}

struct MgTimer {
    gpx_mg *g; hipStream_t st; size_t idx; bool on;
    // T_WAIT: two timing-enabled records around every hipStreamWaitEvent of the update stream -- 2 (chunks + 1) marker packets
    // per owned panel on the stream that bounds the step -- only when the caller asked for the exposed-wait figures
    MgTimer(gpx_mg *g_, int cls, hipStream_t s, int64_t panel = -1) : g(g_), st(s), idx(0), on(g_->timing && (cls != T_WAIT || g_->wait_timing))
    {
        if (!on) return;
        if (g->tev_next + 2 > g->tev.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { (void)hipGetLastError(); on = false; return; }
            g->tev.push_back(a); g->tev.push_back(b); g->tcls.push_back(cls); g->tpan.push_back(panel);
        }
        idx = g->tev_next; g->tev_next += 2;
        g->tcls[idx / 2] = cls; g->tpan[idx / 2] = panel;
        (void)hipEventRecord(g->tev[idx], st);
    }
    ~MgTimer() { if (on) (void)hipEventRecord(g->tev[idx + 1], st); }
};

static ncclDataType_t nccl_type(int dtype) { return dtype == GPX_F64 ? ncclFloat64 : ncclFloat32; }

// rehearsal: hold the stream for `ticks` of the 100 MHz real-time clock (one lane; a modelled transfer)
__global__ void mg_delay_kernel(unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
static int mg_delay(gpx_mg *g, double us, hipStream_t st)
{
    if (us <= 0) return GPX_OK;
    g->reh_model_ms += us * 1e-3;
    hipLaunchKernelGGL(mg_delay_kernel, dim3(1), dim3(1), 0, st, (unsigned long long)(us * 100.0));
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}
// the modelled duration of one panel-chunk broadcast of `bytes` (DESIGN section 5's assumptions, stated in the JSON the
// rehearsal writes): a ring moves the payload through one link per hop, pipelined -- bytes / rate; scatter + all-gather puts
// 1 / P of it on each of the root's links, twice -- 2 bytes / (P rate); one latency per collective call
static double mg_model_us(const gpx_mg *g, size_t bytes, bool sag)
{
    const double per_us = g->reh_GBps * 1e3;                      // bytes per microsecond
    return sag ? 2.0 * (double)bytes / ((double)g->world * per_us) + 2.0 * g->reh_lat_us : (double)bytes / per_us + g->reh_lat_us;
}

// broadcast `count` elements of the handle's dtype at dev_ptr from `root`, on stream st
static int mg_bcast(gpx_mg *g, void *dev_ptr, size_t count, int root, hipStream_t st)
{
    if (count == 0) return GPX_OK;
    if (g->rehearse) {
        // (only alpha blocks travel this way: a block this rank does not own comes out of the resident solution)
        if (root != g->rank && g->reh_alpha) {
            const size_t off = (size_t)((const char *)dev_ptr - (const char *)g->alpha);
            GPX_HIP(hipMemcpyAsync(dev_ptr, (const char *)g->reh_alpha + off, count * g->es, hipMemcpyDeviceToDevice, st));
        }
        return mg_delay(g, g->reh_lat_us, st);
    }
    if (g->cb_bcast) {
        if (g->world == 1) return GPX_OK;
        const int rc = g->cb_bcast(g->cb_user, dev_ptr, count * g->es, root, (void *)st);
        if (rc != 0) { set_error("broadcast callback failed (%d)", rc); return GPX_ERR_HIP; }
        return GPX_OK;
    }
    if (g->world == 1 && !tune().force_collectives) return GPX_OK;
    if (!g->comm) { set_error("this handle has no communicator (it was handed to another handle by gpx_mg_adopt_comm, or never connected)"); return GPX_ERR_ARG; }
    GPX_NCCL(g_rccl.Broadcast(dev_ptr, dev_ptr, count, nccl_type(g->dtype), root, g->comm, st));
    return GPX_OK;
}

// op 0: sum of `count` elements of dtype (GPX_F64 / GPX_F32 / 2 = int32); op 1: max
static int mg_allreduce(gpx_mg *g, void *dev_ptr, size_t count, int dtype, int op, hipStream_t st)
{
    if (count == 0) return GPX_OK;
    if (g->rehearse) return mg_delay(g, g->reh_lat_us, st);       // (local values only: the rehearsal's scalars are this rank's share)
    if (g->cb_allreduce) {
        if (g->world == 1) return GPX_OK;
        const int rc = g->cb_allreduce(g->cb_user, dev_ptr, count, dtype, op, (void *)st);
        if (rc != 0) { set_error("all-reduce callback failed (%d)", rc); return GPX_ERR_HIP; }
        return GPX_OK;
    }
    if (g->world == 1 && !tune().force_collectives) return GPX_OK;
    if (!g->comm) { set_error("this handle has no communicator (it was handed to another handle by gpx_mg_adopt_comm, or never connected)"); return GPX_ERR_ARG; }
    const ncclDataType_t t = dtype == GPX_F64 ? ncclFloat64 : (dtype == GPX_F32 ? ncclFloat32 : ncclInt32);
    GPX_NCCL(g_rccl.AllReduce(dev_ptr, dev_ptr, count, t, op == 0 ? ncclSum : ncclMax, g->comm, st));
    return GPX_OK;
}

// Panel broadcast.  Default: one collective (ncclBroadcast; RCCL picks its rings).  "sag" (GPX_MG_BCAST=sag or
// gpx_mg_set_bcast): two point-to-point phases -- the root scatters piece i to rank i, then every rank sends its
// piece straight to every other rank -- so that each of the root's xGMI links carries 1 / P of the payload per phase
// instead of one ring pushing all of it through link after link (xGMI is point to point, 7 links per GPU).
// The callback back-end (tests: several ranks on one GPU) emulates the two phases with the broadcast callback: in
// phase 1 piece i is received into its place only by rank i (the others drop it into a scratch block), in phase 2
// piece i is re-broadcast by rank i, so a wrong piece map or a missing piece shows in the result.
// The arithmetic of the two splits a panel broadcast goes through, as pure functions (also behind gpx_debug_mg_plan, so
// that alignment and coverage can be checked at N = 65536, P = 8 without a GPU-sized run):
//   pieces of one broadcast (scatter + all-gather): P - 1 pieces of mg_piece() elements, the last one takes the rest
static size_t mg_piece(size_t count, int P) { return count / (size_t)P / 32 * 32; }   // 32 elements = 128 / 256 bytes
static bool mg_piece_sag_ok(size_t count, int P) { return P > 2 && mg_piece(count, P) >= 1024; }
//   row chunks of one panel (each its own broadcast + event): ends[c] = first row AFTER chunk c, relative to the panel's
//   first row; the first chunk covers at least the next block column's diagonal rows (its B-operand rows)
static void mg_chunk_plan(int64_t rows, int64_t nb, int nch_req, std::vector<int64_t> *ends)
{
    ends->clear();
    int nch = std::max(1, nch_req);
    const int64_t min_rows = std::min(rows, std::max<int64_t>(2 * nb, 1024));
    if (rows < 4 * min_rows) nch = 1;
    int64_t done = 0;
    for (int c = 0; c < nch; ++c) {
        int64_t end = (c + 1 == nch) ? rows : std::max(min_rows, (rows * (c + 1) / nch) / 128 * 128);
        end = std::min(end, rows);
        if (end <= done) continue;
        ends->push_back(end);
        done = end;
    }
}

static int mg_pack_from(gpx_mg *g, const void *src, int64_t ld, int64_t rows, int64_t kb, void *buf, hipStream_t st);
// (j, row0, rows: which rows of which panel dev_ptr holds -- the rehearsal needs to know what it stands in for)
static int mg_bcast_panel(gpx_mg *g, void *dev_ptr, size_t count, int root, hipStream_t st, int64_t j = -1, int64_t row0 = 0, int64_t rows = 0)
{
    const int P = g->world, me = g->rank;
    const size_t piece = mg_piece(count, P);                      // 32-element (128- / 256-byte) aligned pieces; the last one takes the rest
    if (g->rehearse) {
        const bool sag_m = g->bcast_sag && mg_piece_sag_ok(count, P);
        route_hit(sag_m ? RT_MG_BCAST_SAG : RT_MG_BCAST_ONE);
        if (root != me) {
            if (j < 0 || !g->reh_L) { set_error("rehearsal: a panel broadcast without its context"); return GPX_ERR_ARG; }
            const int64_t k0 = g->k0(j), kb = g->kb(j);
            GPX_TRY(mg_pack_from(g, (const char *)g->reh_L + ((size_t)(k0 + row0) * g->reh_ld + k0) * g->es, g->reh_ld, rows, kb, dev_ptr, st));
        }
        return mg_delay(g, mg_model_us(g, count * g->es, sag_m), st);
    }
    const bool sag = g->bcast_sag && mg_piece_sag_ok(count, P) && (g->cb_bcast || g->comm);
    if (!sag) { route_hit(RT_MG_BCAST_ONE); return mg_bcast(g, dev_ptr, count, root, st); }
    route_hit(RT_MG_BCAST_SAG);
    auto off = [&](int i) { return (size_t)i * piece; };
    auto len = [&](int i) { return i + 1 == P ? count - off(i) : piece; };
    auto at = [&](int i) { return (char *)dev_ptr + off(i) * g->es; };
    if (g->cb_bcast) {
        const size_t need = (count - off(P - 1)) * g->es;
        if (g->sag_tmp_bytes < need) {
            GPX_HIP(hipStreamSynchronize(st));
            if (g->sag_tmp) (void)hipFree(g->sag_tmp);
            g->sag_tmp = nullptr; g->sag_tmp_bytes = 0;
            GPX_HIP(hipMalloc(&g->sag_tmp, need));
            g->sag_tmp_bytes = need;
        }
        for (int i = 0; i < P; ++i) {                             // phase 1: root -> rank i
            if (i == root) continue;
            void *dst = (me == root || me == i) ? (void *)at(i) : g->sag_tmp;
            const int rc = g->cb_bcast(g->cb_user, dst, len(i) * g->es, root, (void *)st);
            if (rc != 0) { set_error("broadcast callback failed (%d)", rc); return GPX_ERR_HIP; }
        }
        for (int i = 0; i < P; ++i) {                             // phase 2: rank i -> everybody
            const int rc = g->cb_bcast(g->cb_user, at(i), len(i) * g->es, i, (void *)st);
            if (rc != 0) { set_error("broadcast callback failed (%d)", rc); return GPX_ERR_HIP; }
        }
        return GPX_OK;
    }
    const ncclDataType_t t = nccl_type(g->dtype);
    GPX_NCCL(g_rccl.GroupStart());
    if (me == root) {
        for (int i = 0; i < P; ++i)
            if (i != root) GPX_NCCL(g_rccl.Send(at(i), len(i), t, i, g->comm, st));
    } else {
        GPX_NCCL(g_rccl.Recv(at(me), len(me), t, root, g->comm, st));
    }
    GPX_NCCL(g_rccl.GroupEnd());
    GPX_NCCL(g_rccl.GroupStart());
    for (int j = 0; j < P; ++j) {
        if (j == me) continue;
        if (j != root) GPX_NCCL(g_rccl.Send(at(me), len(me), t, j, g->comm, st));      // (the root already holds every piece)
        if (me != root) GPX_NCCL(g_rccl.Recv(at(j), len(j), t, j, g->comm, st));
    }
    GPX_NCCL(g_rccl.GroupEnd());
    return GPX_OK;
}

static int mg_pack_from(gpx_mg *g, const void *src, int64_t ld, int64_t rows, int64_t kb, void *buf, hipStream_t st)
{
    if (rows <= 0) return GPX_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(rows * cdiv(kb, 2), 256), 4096);
    if (g->dtype == GPX_F64)
        hipLaunchKernelGGL((pack_panel_kernel<double>), dim3(blocks), dim3(256), 0, st, (const double *)src, ld,
                           (double *)buf, g->nb, rows, (int)kb);
    else
        hipLaunchKernelGGL((pack_panel_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float *)src, ld,
                           (float *)buf, g->nb, rows, (int)kb);
    GPX_LAUNCH_CHECK();
    return GPX_OK;
}
static int mg_pack(gpx_mg *g, int64_t r0, int64_t cl, int64_t rows, int64_t kb, void *buf, hipStream_t st)
{
    return mg_pack_from(g, g->Aat(r0, cl), g->ld, rows, kb, buf, st);
}

// Factor (owner, stream Q), pack (owner, Q) and broadcast (everybody, stream B) block column j into buf.  The
// broadcasts have a stream of their own: all ranks issue them in panel order on B, and Q -- the panel chain -- never
// queues a factorisation behind a transfer it does not depend on.  buf_free: the last update that read `buf`
// (pack and receive wait for it; the factorisation itself, in place in A, does not).  chunk_ev: when non-null the
// broadcast goes out in row chunks and an event is recorded after each (chunk_end[c] = first row AFTER chunk c,
// relative to the panel's first row).
static int mg_factor_and_bcast(gpx_mg *g, int64_t j, void *buf, hipEvent_t buf_free, std::vector<hipEvent_t> *chunk_ev,
                               std::vector<int64_t> *chunk_end)
{
    const int64_t r0 = g->k0(j), kb = g->kb(j), rows = g->nr - r0;   // (the rider row n travels with every panel)
    hipStream_t Q = g->Q, B = g->B;
    if (g->owner(j) == g->rank) {
        const int64_t cl = g->local_col(j);
        { MgTimer t(g, T_PANEL, Q, j); GPX_TRY(potrf_panel(g->dtype, g->A, g->ld, g->nr, r0, cl, kb, g->info, Q)); }
        GPX_TRY(mg_event(g, &g->last_panel_ev));
        GPX_HIP(hipEventRecord(g->last_panel_ev, Q));
        if (buf_free) GPX_HIP(hipStreamWaitEvent(Q, buf_free, 0));
        { MgTimer t(g, T_PACK, Q, j); GPX_TRY(mg_pack(g, r0, cl, rows, kb, buf, Q)); }
        GPX_TRY(mg_order(g, Q, B));
        // the inverse of this diagonal block for the backward solve: ~11 small launches, on a low-priority stream of
        // their own ordered after the pack (round 4; they used to sit on Q, where with world <= 2 the owner's next panel
        // queued right behind them; their buffers come from mg_alloc, nothing is allocated inside this loop)
        GPX_TRY(mg_order(g, Q, g->O));
        GPX_TRY(trsv_ops_build(g->dtype, g->Aat(r0, cl), kb, g->ld, &g->ops[(size_t)(j / g->world)], g->O));
    } else {
        if (buf_free) GPX_HIP(hipStreamWaitEvent(B, buf_free, 0));
        // rehearsal: the remote owner's chain behind the arrival of the panel before -- taken to be what THIS rank measured
        // for the nearest panel it owns in the fit before (symmetric ranks; nothing in the first fit)
        if (g->rehearse && !g->own_chain_ms.empty()) {
            double best = 0; int64_t dist = -1;
            for (int64_t o : g->my_blocks) {
                const int64_t dd = o > j ? o - j : j - o;
                if ((size_t)o < g->own_chain_ms.size() && g->own_chain_ms[(size_t)o] > 0 && (dist < 0 || dd < dist)) { dist = dd; best = g->own_chain_ms[(size_t)o]; }
            }
            const double before = g->reh_model_ms;
            GPX_TRY(mg_delay(g, best * 1e3, B));
            g->reh_chain_ms += g->reh_model_ms - before;          // (kept apart from the transfers)
            g->reh_model_ms = before;
        }
    }
    // row chunks (mg_chunk_plan)
    std::vector<int64_t> ends;
    mg_chunk_plan(rows, g->nb, (chunk_ev && g->world > 1) ? g->bcast_chunks : 1, &ends);
    MgTimer t(g, T_BCAST, B);
    int64_t done = 0;
    for (const int64_t end : ends) {
        GPX_TRY(mg_bcast_panel(g, (char *)buf + (size_t)done * g->nb * g->es, (size_t)(end - done) * g->nb, (int)g->owner(j), B, j, done,
                               end - done));
        done = end;
        if (chunk_ev) {
            hipEvent_t e;
            GPX_TRY(mg_event(g, &e));
            GPX_HIP(hipEventRecord(e, B));
            chunk_ev->push_back(e);
            chunk_end->push_back(end);
        }
    }
    return GPX_OK;
}

static int mg_build(gpx_mg *g, const double *params, double s)
{
    MgTimer t(g, T_BUILD, g->S);
    for (int64_t j : g->my_blocks) {
        const int64_t r0 = g->k0(j), kb = g->kb(j), cl = g->local_col(j);
        // A[r0:n, cl:cl+kb] <- K(x[r0:n], x[r0:r0+kb]) + s^2 on the block's diagonal; lower tiles only
        GPX_TRY(kmat(g->dtype, g->kernel, GPX_K, (const char *)g->x + (size_t)r0 * g->d * g->es, g->n - r0,
                     (const char *)g->x + (size_t)r0 * g->d * g->es, kb, g->d, params, s * s, GPX_LOWER, g->Aat(r0, cl),
                     g->ld, g->S));
        // row n of the block column: y of its columns.  It takes part in every panel and update as one more ROW, which
        // is forward substitution block by block: when the factor is done row n holds L^-1 y (gpx_gp_fit does the same
        // on one GPU up to n = 16384; here it replaces a chain of nblk all-reduces)
        GPX_HIP(hipMemcpyAsync(g->Aat(g->n, cl), (const char *)g->y + (size_t)r0 * g->es, (size_t)kb * g->es,
                               hipMemcpyDeviceToDevice, g->S));
    }
    return GPX_OK;
}

static int mg_factor(gpx_mg *g)
{
    hipStream_t S = g->S, Q = g->Q;
    MgTimer tf(g, T_FACTOR, S);
    GPX_TRY(mg_order(g, S, Q));                                   // the kernel build is done
    std::vector<hipEvent_t> cev; std::vector<int64_t> cend;
    GPX_TRY(mg_factor_and_bcast(g, 0, g->pbuf[0], nullptr, &cev, &cend));
    hipEvent_t readers_done[2] = {nullptr, nullptr};              // last update that read pbuf[i]
    for (int64_t k = 0; k < g->nblk; ++k) {
        const int64_t k0 = g->k0(k), kb = g->kb(k), r = k0 + kb;
        if (r >= g->n) {                                          // last panel: S must see it before the solve
            { MgTimer tw(g, T_WAIT, S); GPX_HIP(hipStreamWaitEvent(S, cev.back(), 0)); }
            break;
        }
        void *Pk = g->pbuf[k % 2];
        const int64_t nxt = k + 1;
        const bool own_next = g->owner(nxt) == g->rank;
        int64_t jl_first = g->first_local_block_after(k);
        if (own_next) {
            // block column k+1 first, chunk by chunk as the panel lands (rows of chunk c: [lo, hi) global)
            const int64_t cl = g->local_col(nxt);
            int64_t lo = r;
            for (size_t c = 0; c < cev.size(); ++c) {
                const int64_t hi = k0 + cend[c];
                { MgTimer tw(g, T_WAIT, S); GPX_HIP(hipStreamWaitEvent(S, cev[c], 0)); }
                if (hi > lo) {
                    MgTimer t(g, T_UPDATE, S, c + 1 == cev.size() ? nxt : -1);   // (the last chunk's update is what the panel waits for)
                    GPX_TRY(syrk_bc(g->dtype, hi, lo, g->A, g->ld, cl, cl + g->nb, Pk, g->nb, k0, kb, g->nb, g->world,
                                    g->rank, S));       // (hi reaches nr with the last chunk: the rider row is its last row)
                }
                lo = std::max(lo, hi);
            }
            GPX_TRY(mg_order(g, S, Q));
            jl_first = g->first_local_block_after(nxt);
        } else {
            { MgTimer tw(g, T_WAIT, S); GPX_HIP(hipStreamWaitEvent(S, cev.back(), 0)); }   // the whole panel k is here
        }
        std::vector<hipEvent_t> nev; std::vector<int64_t> nend;
        GPX_TRY(mg_factor_and_bcast(g, nxt, g->pbuf[nxt % 2], readers_done[nxt % 2], &nev, &nend));   // (update k-1 has to let go of that buffer)
        // OWNER FIRST (round 5): the panel is the serial chain of the whole run -- every rank waits for it -- while a
        // rank's own trailing update has slack whenever the chain is the bound (it owns one panel in `world`).  Beside that
        // update a 512-wide panel takes 1.2 - 1.5 ms, alone 0.55: with this switch the owner holds its update back until
        // its panel has left, and only then does its share of the update.  Measured in the rehearsal (N = 65536, one rank's
        // share, 100 GB/s links assumed; profiles/r05_mg_rehearsal_owner_first.jsonl): the owner's chain per panel 1.50 ->
        // 0.74 ms; P = 8: 0.285 -> 0.235 s (nb = 512), 0.290 -> 0.226 (nb = 1024); P = 4: 0.434 -> 0.407; P = 2: 0.762 -> 0.738
        if (own_next && g->owner_first && g->last_panel_ev) GPX_HIP(hipStreamWaitEvent(S, g->last_panel_ev, 0));
        if (jl_first >= 0) {
            MgTimer t(g, T_UPDATE, S);
            GPX_TRY(syrk_bc(g->dtype, g->nr, r, g->A, g->ld, jl_first * g->nb, g->ncols_local, Pk, g->nb, k0, kb, g->nb,
                            g->world, g->rank, S));
        }
        hipEvent_t e;
        GPX_TRY(mg_event(g, &e));
        GPX_HIP(hipEventRecord(e, S));
        readers_done[k % 2] = e;
        cev.swap(nev); cend.swap(nend);
    }
    GPX_TRY(mg_order(g, Q, S));
    GPX_TRY(mg_order(g, g->B, S));
    GPX_TRY(mg_order(g, g->O, S));                                // the solve reads the operators
    if (g->debug_info != 0) {                                     // test hook: as if a resident panel launch had failed
        GPX_HIP(hipMemcpyAsync(g->info, &g->debug_info, sizeof(int), hipMemcpyHostToDevice, S));
        GPX_HIP(hipStreamSynchronize(S));
        g->debug_info = 0;
    }
    return GPX_OK;
}

// alpha = K^-1 y.  The forward substitution L t = y rode along in the factorisation (row n of the local matrices holds
// this rank's blocks of t).  Backward, L^T alpha = t, right-looking over the block columns from the last: the owner of
// block j finishes alpha_j from its block of the running right-hand side -- one launch with the inverse of the diagonal
// block prepared after the panel (nb a multiple of 512; the step route otherwise) --, broadcasts it (nb numbers), and
// every rank subtracts  L[rows of block j, its columns left of block j]^T alpha_j  from its part of the right-hand side
// in one launch.  Per block: two small launches and one broadcast on the chain; round 2 walked the blocks twice with an
// all-reduce, a broadcast and five launches each (25 ms at N = 65536 with one rank; measured now: DESIGN section 5).
static int mg_solve(gpx_mg *g)
{
    hipStream_t S = g->S;
    MgTimer t(g, T_SOLVE, S);
    const int64_t n = g->n;
    char *rhs = g->Aat(n, 0);                                      // row n: this rank's blocks of L^-1 y, then of the running rhs
    for (int64_t j = g->nblk - 1; j >= 0; --j) {
        const int64_t r0 = g->k0(j), kb = g->kb(j);
        if (g->owner(j) == g->rank) {
            const int64_t cl = g->local_col(j);
            GPX_HIP(hipMemcpyAsync(g->tmp, rhs + (size_t)cl * g->es, (size_t)kb * g->es, hipMemcpyDeviceToDevice, S));
            GPX_TRY(trsv_lower(g->dtype, g->Aat(r0, cl), kb, g->ld, g->tmp, (char *)g->alpha + (size_t)r0 * g->es, 1, S, nullptr,
                               &g->ops[(size_t)(j / g->world)]));
        }
        GPX_TRY(mg_bcast(g, (char *)g->alpha + (size_t)r0 * g->es, (size_t)kb, (int)g->owner(j), S));
        int64_t before = 0;                                          // this rank's block columns left of block j
        for (int64_t b : g->my_blocks) if (b < j) ++before;
        const int64_t ncols = before * g->nb;
        if (ncols > 0) {
            const unsigned blocks = (unsigned)cdiv(ncols, 64);
            if (g->dtype == GPX_F64)
                hipLaunchKernelGGL((rowblock_gemv_t_kernel<double>), dim3(blocks), dim3(256), 0, S, (const double *)g->Aat(r0, 0), g->ld,
                                   (int)kb, ncols, (const double *)g->alpha + r0, (double *)rhs);
            else
                hipLaunchKernelGGL((rowblock_gemv_t_kernel<float>), dim3(blocks), dim3(256), 0, S, (const float *)g->Aat(r0, 0), g->ld,
                                   (int)kb, ncols, (const float *)g->alpha + r0, (float *)rhs);
            GPX_LAUNCH_CHECK();
        }
    }
    return GPX_OK;
}

static int mg_reduce(gpx_mg *g)
{
    hipStream_t S = g->S;
    {
        MgTimer t(g, T_REDUCE, S);
        GPX_HIP(hipMemsetAsync(g->scal + 2, 0, sizeof(double), S));
        for (int64_t j : g->my_blocks) {
            GPX_TRY(logdet_chol(g->dtype, g->Aat(g->k0(j), g->local_col(j)), g->kb(j), g->ld, g->scal + 0, S));
            hipLaunchKernelGGL((axpy_slot_kernel<double>), dim3(1), dim3(64), 0, S, g->scal + 2, g->scal + 0);
        }
        GPX_LAUNCH_CHECK();
        GPX_TRY(dot(g->dtype, g->y, g->alpha, g->n, g->scal + 1, S));
        GPX_TRY(mg_allreduce(g, g->scal + 2, 1, GPX_F64, 0, S));
        hipLaunchKernelGGL(info_key_kernel, dim3(1), dim3(64), 0, S, g->info, g->info + 1);
        GPX_LAUNCH_CHECK();
        GPX_TRY(mg_allreduce(g, g->info + 1, 1, 2, 1, S));
    }
    double h[4]; int hi[2];
    GPX_HIP(hipMemcpyAsync(h, g->scal, sizeof(h), hipMemcpyDeviceToHost, S));
    GPX_HIP(hipMemcpyAsync(hi, g->info, sizeof(hi), hipMemcpyDeviceToHost, S));
    GPX_HIP(hipStreamSynchronize(S));
    GPX_HIP(hipStreamSynchronize(g->Q));
    GPX_HIP(hipStreamSynchronize(g->B));
    GPX_HIP(hipStreamSynchronize(g->O));
    g->logdet = h[2]; g->yta = h[1];
    g->info_host = hi[1] == 0 ? 0 : (hi[1] == INT_MAX ? -7 : (1 << 30) - hi[1]);
    return GPX_OK;
}

static int mg_alloc(gpx_mg *g)
{
    const size_t es = g->es;
    const int64_t n = g->n;
#define MG_ALLOC(field, bytes) GPX_HIP(hipMalloc((void **)&g->field, (bytes) ? (bytes) : 16))
    MG_ALLOC(A, (size_t)g->nr * g->ld * es);
    MG_ALLOC(pbuf[0], (size_t)g->nr * g->nb * es);
    MG_ALLOC(pbuf[1], (size_t)g->nr * g->nb * es);
    MG_ALLOC(x, (size_t)n * g->d * es);
    MG_ALLOC(y, (size_t)n * es);
    MG_ALLOC(alpha, (size_t)n * es);
    MG_ALLOC(tmp, (size_t)g->nb * es);
    MG_ALLOC(scal, 4 * sizeof(double));
    MG_ALLOC(info, 4 * sizeof(int));
#undef MG_ALLOC
    GPX_HIP(hipMemset(g->info, 0, 4 * sizeof(int)));
    GPX_HIP(hipStreamCreateWithFlags(&g->S, hipStreamNonBlocking));
    int least = 0, greatest = 0;
    GPX_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    GPX_HIP(hipStreamCreateWithPriority(&g->Q, hipStreamNonBlocking, greatest));   // the panel chain: critical path
    GPX_HIP(hipStreamCreateWithPriority(&g->B, hipStreamNonBlocking, greatest));   // the panel broadcasts
    GPX_HIP(hipStreamCreateWithPriority(&g->O, hipStreamNonBlocking, least));      // the solve operators
    // the operator blocks of every owned diagonal block, here and not on first use inside the factorisation's loop
    // (~21 MB per 1024-wide fp64 block: 1.3 GB at N = 65536 on one rank -- part of this handle's HBM budget)
    for (size_t jl = 0; jl < g->my_blocks.size(); ++jl) {
        const int64_t kb = g->kb(g->my_blocks[jl]);
        if (kb < 512 || kb % 512 != 0) continue;                  // (the step route: trsv_ops_build declines these)
        const size_t need = trsv_ops_bytes(g->dtype, kb);
        GPX_HIP(hipMalloc(&g->ops[jl].buf, need));
        g->ops[jl].bytes = need;
    }
    return GPX_OK;
}

static int mg_new(gpx_mg **out, int dtype, int kernel, int64_t n, int d, int64_t nb, int world, int rank)
{
    GPX_TRY(ensure_device());
    GPX_ARG(out, "mg is NULL");
    *out = nullptr;
    GPX_ARG(dtype == GPX_F64 || dtype == GPX_F32, "dtype must be GPX_F64 or GPX_F32");
    GPX_ARG(kernel == GPX_KERNEL_GAUSSIAN || kernel == GPX_KERNEL_PERIODIC, "unknown kernel family");
    GPX_ARG(n >= 1 && d >= 1, "need n >= 1 and d >= 1");
    GPX_ARG(world >= 1 && rank >= 0 && rank < world, "bad world / rank");
    GPX_ARG(nb >= 64 && nb % 64 == 0 && 1024 % nb == 0, "nb must be 64 / 128 / 256 / 512 / 1024");
    gpx_mg *g = new gpx_mg();
    g->dtype = dtype; g->kernel = kernel; g->n = n; g->d = d; g->nb = nb; g->world = world; g->rank = rank;
    g->es = esize(dtype);
    if (hipGetDevice(&g->device) != hipSuccess) { (void)hipGetLastError(); g->device = 0; }
    g->nblk = cdiv(n, nb);
    for (int64_t j = rank; j < g->nblk; j += world) g->my_blocks.push_back(j);
    g->ncols_local = std::max<int64_t>(1, (int64_t)g->my_blocks.size()) * nb;
    g->ld = g->ncols_local;
    g->nr = n + 1;
    g->ops.resize(std::max<size_t>(1, g->my_blocks.size()));
    g->bcast_chunks = (int)std::max<int64_t>(1, std::min<int64_t>(16, tune().mg_bcast_chunks));
    g->timing = !tune().mg_no_timing;
    if (tune().mg_bcast_set) g->bcast_sag = tune().mg_bcast_sag;
    g->owner_first = tune().mg_owner_first_set ? (tune().mg_owner_first != 0) : (world >= 2);
    *out = g;
    return GPX_OK;
}

}  // namespace gpx

#define MG_ENTER(g)                                                          \
    GPX_ARG((g) != nullptr, "mg is NULL");                                   \
    gpx::tune_refresh();                                                     \
    gpx::DeviceGuard guard__((g)->device);                                   \
    if (guard__.rc != GPX_OK) return guard__.rc;                             \
    gpx::StreamTurn turn__((g)->S)      /* this thread's scratch buffers: one stream at a time (gpx_common.h) */

extern "C" {

int gpx_mg_unique_id(void *id128)
{
    GPX_ARG(id128, "id is NULL");
    GPX_TRY(ensure_device());
    GPX_TRY(rccl_load());
    ncclUniqueId id;
    GPX_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return GPX_OK;
}

int gpx_mg_destroy(gpx_mg_t *g)
{
    if (!g) return GPX_OK;
    gpx::DeviceGuard guard__(g->device);
    if (g->S) (void)hipStreamSynchronize(g->S);
    if (g->Q) (void)hipStreamSynchronize(g->Q);
    if (g->B) (void)hipStreamSynchronize(g->B);
    if (g->O) (void)hipStreamSynchronize(g->O);
    if (g->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g->comm);
    void *bufs[] = {g->A, g->pbuf[0], g->pbuf[1], g->x, g->y, g->alpha, g->tmp, g->scal, g->info, g->sag_tmp};
    for (void *b : bufs) if (b) (void)hipFree(b);
    for (gpx::TrsvOps &o : g->ops) if (o.buf) (void)hipFree(o.buf);
    for (hipEvent_t e : g->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : g->tev) (void)hipEventDestroy(e);
    stream_epoch_bump();
    if (g->S) (void)hipStreamDestroy(g->S);
    if (g->Q) (void)hipStreamDestroy(g->Q);
    if (g->B) (void)hipStreamDestroy(g->B);
    if (g->O) (void)hipStreamDestroy(g->O);
    delete g;
    return GPX_OK;
}

int gpx_mg_probe(void)
{
    GPX_TRY(ensure_device());
    return rccl_load();                                           // dlopen + dlsym only: no bootstrap thread, no socket
}

int gpx_mg_create_local(gpx_mg_t **out, int dtype, int kernel, int64_t n, int d, int64_t nb, int world, int rank)
{
    GPX_TRY(mg_new(out, dtype, kernel, n, d, nb, world, rank));
    gpx_mg *g = *out;
    int rc = rccl_load();
    if (rc == GPX_OK) rc = mg_alloc(g);
    if (rc != GPX_OK) { gpx_mg_destroy(g); *out = nullptr; }
    return rc;
}

int gpx_mg_connect(gpx_mg_t *g, const void *id128)
{
    MG_ENTER(g);
    GPX_ARG(id128, "the RCCL unique id is NULL");
    GPX_ARG(!g->comm && !g->cb_bcast, "the handle already has a communicator");
    GPX_TRY(rccl_load());
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&g->comm, g->world, id, g->rank);
    if (r != ncclSuccess) { g->comm = nullptr; set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); return GPX_ERR_HIP; }
    return GPX_OK;
}

int gpx_mg_create(gpx_mg_t **out, int dtype, int kernel, int64_t n, int d, int64_t nb, int world, int rank,
                  const void *id128)
{
    GPX_ARG(id128, "the RCCL unique id is NULL");
    GPX_TRY(gpx_mg_create_local(out, dtype, kernel, n, d, nb, world, rank));
    const int rc = gpx_mg_connect(*out, id128);
    if (rc != GPX_OK) { gpx_mg_destroy(*out); *out = nullptr; }
    return rc;
}

int gpx_mg_adopt_comm(gpx_mg_t *g, gpx_mg_t *from)
{
    MG_ENTER(g);
    GPX_ARG(from != nullptr && from != g, "no handle to adopt the communicator from");
    GPX_ARG(!g->comm && !g->cb_bcast && !g->rehearse, "the handle already has a communicator");
    GPX_ARG(from->comm != nullptr, "the other handle has no RCCL communicator");
    GPX_ARG(from->world == g->world && from->rank == g->rank && from->device == g->device, "world / rank / device differ");
    // everything the other handle queued on the communicator has to be done before this one issues collectives on it
    if (from->B) GPX_HIP(hipStreamSynchronize(from->B));
    if (from->S) GPX_HIP(hipStreamSynchronize(from->S));
    g->comm = from->comm;
    from->comm = nullptr;                                         // (its destroy no longer touches the communicator)
    return GPX_OK;
}

int gpx_mg_comm_info(gpx_mg_t *g, int *nranks, int *rank, int *device, int *bcast_sag)
{
    MG_ENTER(g);
    if (bcast_sag) *bcast_sag = g->bcast_sag;
    if (g->comm) {                                                // what RCCL itself says about the communicator
        int v = 0;
        if (nranks) { GPX_NCCL(g_rccl.CommCount(g->comm, &v)); *nranks = v; }
        if (rank) { GPX_NCCL(g_rccl.CommUserRank(g->comm, &v)); *rank = v; }
        if (device) { GPX_NCCL(g_rccl.CommCuDevice(g->comm, &v)); *device = v; }
        return GPX_OK;
    }
    if (nranks) *nranks = 0;                                      // no RCCL communicator (callback back-end)
    if (rank) *rank = g->rank;
    if (device) *device = g->device;
    return GPX_OK;
}

int gpx_mg_set_bcast(gpx_mg_t *g, int sag)
{
    MG_ENTER(g);
    g->bcast_sag = sag ? 1 : 0;                                   // collective: every rank sets the same mode before the next fit
    return GPX_OK;
}

int gpx_mg_set_owner_first(gpx_mg_t *g, int on)
{
    MG_ENTER(g);
    g->owner_first = on ? 1 : 0;                                  // (local scheduling only: ranks need not agree, but a run should)
    return GPX_OK;
}

int gpx_mg_set_wait_timing(gpx_mg_t *g, int on)
{
    MG_ENTER(g);
    g->wait_timing = on != 0;
    return GPX_OK;
}

int gpx_mg_schedule_info(gpx_mg_t *g, int *owner_first, int *chunks, int *sag, int *wait_timing)
{
    MG_ENTER(g);
    if (owner_first) *owner_first = g->owner_first ? 1 : 0;
    if (chunks) *chunks = (int)g->bcast_chunks;
    if (sag) *sag = g->bcast_sag ? 1 : 0;
    if (wait_timing) *wait_timing = g->wait_timing ? 1 : 0;
    return GPX_OK;
}

int gpx_mg_set_chunks(gpx_mg_t *g, int chunks)
{
    MG_ENTER(g);
    GPX_ARG(chunks >= 1 && chunks <= 16, "chunks must be 1 .. 16");
    g->bcast_chunks = chunks;                                     // collective: every rank sets the same number before the next fit
    return GPX_OK;
}

int gpx_mg_create_rehearsal(gpx_mg_t **out, int dtype, int kernel, int64_t n, int d, int64_t nb, int world, int rank,
                            const void *L_dev, int64_t ldl, const void *alpha_dev, double link_GBps, double latency_us)
{
    GPX_ARG(L_dev && ldl >= n && alpha_dev, "the resident factor (n + 1 rows) and solution are needed");
    GPX_ARG(link_GBps > 0 && latency_us >= 0, "bad transfer model");
    GPX_TRY(mg_new(out, dtype, kernel, n, d, nb, world, rank));
    gpx_mg *g = *out;
    g->rehearse = true;
    g->reh_L = L_dev; g->reh_ld = ldl; g->reh_alpha = alpha_dev;
    g->reh_GBps = link_GBps; g->reh_lat_us = latency_us;
    const int rc = mg_alloc(g);
    if (rc != GPX_OK) { gpx_mg_destroy(g); *out = nullptr; }
    return rc;
}

int gpx_debug_mg_inject_info(gpx_mg_t *g, int value)
{
    GPX_ARG(g != nullptr, "mg is NULL");
    g->debug_info = value;
    return GPX_OK;
}

int gpx_debug_mg_plan(int64_t n, int64_t nb, int world, int chunks, int64_t j, int64_t *out, int cap)
{
    // host arithmetic only (no device is touched): the row chunks of panel j and the pieces of each chunk's broadcast
    GPX_ARG(n >= 1 && nb >= 64 && world >= 1 && chunks >= 1 && j >= 0 && j * nb < n && out && cap >= 0, "bad arguments");
    const int64_t rows = (n + 1) - j * nb;                        // (the rider row travels with every panel)
    std::vector<int64_t> ends;
    mg_chunk_plan(rows, nb, world > 1 ? chunks : 1, &ends);
    int64_t done = 0;
    int k = 0;
    for (const int64_t end : ends) {
        if ((k + 1) * 6 > cap) { set_error("gpx_debug_mg_plan: out holds %d values, more are needed", cap); return GPX_ERR_ARG; }
        const size_t count = (size_t)(end - done) * (size_t)nb;
        const size_t piece = mg_piece(count, world);
        int64_t *o = out + (size_t)k * 6;
        o[0] = done; o[1] = end; o[2] = (int64_t)count; o[3] = (int64_t)piece;
        o[4] = (int64_t)(count - (size_t)(world - 1) * piece);    // the last piece
        o[5] = mg_piece_sag_ok(count, world) ? 1 : 0;
        done = end; ++k;
    }
    return k;                                                     // >= 0: number of chunks written
}

int gpx_mg_create_cb(gpx_mg_t **out, int dtype, int kernel, int64_t n, int d, int64_t nb, int world, int rank,
                     gpx_mg_bcast_fn bcast, gpx_mg_allreduce_fn allreduce, void *user)
{
    GPX_ARG(world == 1 || (bcast && allreduce), "callbacks are NULL");
    GPX_TRY(mg_new(out, dtype, kernel, n, d, nb, world, rank));
    gpx_mg *g = *out;
    g->cb_bcast = bcast ? bcast : [](void *, void *, size_t, int, void *) { return 0; };
    g->cb_allreduce = allreduce ? allreduce : [](void *, void *, size_t, int, int, void *) { return 0; };
    g->cb_user = user;
    const int rc = mg_alloc(g);
    if (rc != GPX_OK) { gpx_mg_destroy(g); *out = nullptr; }
    return rc;
}

int gpx_mg_set_data(gpx_mg_t *g, const double *x, const double *y)
{
    MG_ENTER(g);
    GPX_ARG(x && y, "NULL argument");
    const int64_t n = g->n;
    // the check_finite of the reference's cho_factor / cho_solve (gp/gp.py:294, 332-334), on the host copies that
    // every rank holds: all ranks refuse together, before any collective
    for (int64_t i = 0; i < n * g->d; ++i)
        if (!std::isfinite(x[i])) { set_error("array must not contain infs or NaNs (x)"); return GPX_ERR_ARG; }
    for (int64_t i = 0; i < n; ++i)
        if (!std::isfinite(y[i])) { set_error("array must not contain infs or NaNs (y)"); return GPX_ERR_ARG; }
    if (g->dtype == GPX_F64) {
        GPX_HIP(hipMemcpy(g->x, x, (size_t)n * g->d * 8, hipMemcpyHostToDevice));
        GPX_HIP(hipMemcpy(g->y, y, (size_t)n * 8, hipMemcpyHostToDevice));
    } else {
        std::vector<float> hx((size_t)n * g->d), hy((size_t)n);
        for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)x[i];
        for (size_t i = 0; i < hy.size(); ++i) hy[i] = (float)y[i];
        GPX_HIP(hipMemcpy(g->x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        GPX_HIP(hipMemcpy(g->y, hy.data(), hy.size() * 4, hipMemcpyHostToDevice));
    }
    g->have_data = true; g->fitted = false;
    return GPX_OK;
}

int gpx_mg_fit(gpx_mg_t *g, const double *params, double s, double *log_lh, int *info)
{
    MG_ENTER(g);
    GPX_ARG(g->have_data && params, "set_data must be called before fit");
    GPX_ARG(!(s < 0), "invalid value for s");
    if (!kernel_values_finite(g->kernel, params, s, g->dtype)) { set_error("array must not contain infs or NaNs"); return GPX_ERR_ARG; }
    g->fitted = false;
    g->ev_next = 0; g->tev_next = 0;
    GPX_HIP(hipMemsetAsync(g->info, 0, 4 * sizeof(int), g->S));
    GPX_TRY(mg_build(g, params, s));
    GPX_TRY(mg_factor(g));
    GPX_TRY(mg_solve(g));
    GPX_TRY(mg_reduce(g));
    GPX_TRY(check_internal_info(g->info_host));                   // every rank sees the same reduced word: all fail together
    g->fitted = true;
    // stage and chain times of this rank (HIP events, all streams drained by mg_reduce)
    for (double &v : g->ms) v = 0;
    std::vector<double> chain_now((size_t)g->nblk, 0.0);
    for (size_t i = 0; i + 1 < g->tev_next; i += 2) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g->tev[i], g->tev[i + 1]) != hipSuccess) { (void)hipGetLastError(); continue; }
        const int cls = g->tcls[i / 2];
        g->ms[cls] += t;
        if (g->tpan[i / 2] >= 0 && (cls == T_PANEL || cls == T_PACK || cls == T_UPDATE)) chain_now[(size_t)g->tpan[i / 2]] += t;
        if (cls == T_WAIT) {                                      // [9] the longest single wait, [10] waits of more than 20 us
            g->ms[9] = std::max(g->ms[9], (double)t);
            if (t > 0.02f) g->ms[10] += 1.0;
        }
    }
    g->own_chain_ms.swap(chain_now);
    g->ms[11] = g->reh_model_ms;                                  // (rehearsal: the modelled transfer time this fit enqueued)
    g->ms[12] = g->reh_chain_ms;                                  // (rehearsal: the remote owners' chains it stood in for)
    g->reh_model_ms = 0; g->reh_chain_ms = 0;
    if (info) *info = g->info_host;
    if (log_lh) {
        // gp/gp.py:362-365 and gp_c.pyx:22-29
        if (g->info_host != 0 || !(g->logdet >= GPX_MIN_LOG)) *log_lh = -INFINITY;
        else *log_lh = -0.5 * g->yta - 0.5 * g->logdet - 0.5 * (double)g->n * log(2 * M_PI);
    }
    return GPX_OK;
}

int gpx_mg_mean(gpx_mg_t *g, const double *params, const double *xo, int64_t m, double *out)
{
    MG_ENTER(g);
    GPX_ARG(g->fitted && params, "mg is not fitted");
    GPX_ARG(m >= 0 && (m == 0 || (xo && out)), "bad arguments");
    if (m == 0) return GPX_OK;
    const size_t es = g->es;
    void *dxo = nullptr, *dout = nullptr;
    GPX_HIP(hipMalloc(&dxo, (size_t)m * g->d * es));
    hipError_t e = hipMalloc(&dout, (size_t)m * es);
    if (e != hipSuccess) { (void)hipFree(dxo); return hip_fail(e, "hipMalloc", __FILE__, __LINE__); }
    int rc = GPX_OK;
    {
        if (g->dtype == GPX_F64) e = hipMemcpy(dxo, xo, (size_t)m * g->d * 8, hipMemcpyHostToDevice);
        else {
            std::vector<float> h((size_t)m * g->d);
            for (size_t i = 0; i < h.size(); ++i) h[i] = (float)xo[i];
            e = hipMemcpy(dxo, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy", __FILE__, __LINE__);
    }
    // every rank evaluates a slice of the test points, one all-reduce assembles the vector
    const int64_t per = cdiv(m, g->world);
    const int64_t m0 = std::min(m, g->rank * per), m1 = std::min(m, (g->rank + 1) * per);
    if (rc == GPX_OK && hipMemsetAsync(dout, 0, (size_t)m * es, g->S) != hipSuccess) rc = GPX_ERR_HIP;
    if (rc == GPX_OK && m1 > m0)
        rc = gpx_d_mean(g->dtype, g->kernel, (char *)dxo + (size_t)m0 * g->d * es, m1 - m0, g->x, g->n, g->d, params, g->alpha,
                        (char *)dout + (size_t)m0 * es, (void *)g->S);
    if (rc == GPX_OK) rc = mg_allreduce(g, dout, (size_t)m, g->dtype, 0, g->S);
    if (rc == GPX_OK) {
        if (g->dtype == GPX_F64) e = hipMemcpyAsync(out, dout, (size_t)m * 8, hipMemcpyDeviceToHost, g->S);
        std::vector<float> h;
        if (g->dtype != GPX_F64) { h.resize((size_t)m); e = hipMemcpyAsync(h.data(), dout, (size_t)m * 4, hipMemcpyDeviceToHost, g->S); }
        if (e == hipSuccess) e = hipStreamSynchronize(g->S);
        if (e != hipSuccess) rc = hip_fail(e, "mean copy-out", __FILE__, __LINE__);
        else if (g->dtype != GPX_F64) for (int64_t i = 0; i < m; ++i) out[i] = (double)h[(size_t)i];
    }
    (void)hipStreamSynchronize(g->S);
    (void)hipFree(dxo); (void)hipFree(dout);
    return rc;
}

int gpx_mg_get_alpha(gpx_mg_t *g, double *out)
{
    MG_ENTER(g);
    GPX_ARG(g->fitted && out, "bad arguments");
    if (g->dtype == GPX_F64) {
        GPX_HIP(hipMemcpy(out, g->alpha, (size_t)g->n * 8, hipMemcpyDeviceToHost));
    } else {
        std::vector<float> h((size_t)g->n);
        GPX_HIP(hipMemcpy(h.data(), g->alpha, h.size() * 4, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < g->n; ++i) out[i] = (double)h[(size_t)i];
    }
    return GPX_OK;
}

int gpx_mg_scalars(gpx_mg_t *g, double *logdet, double *yta, int *info)
{
    MG_ENTER(g);
    GPX_ARG(g->fitted, "mg is not fitted");
    if (logdet) *logdet = g->logdet;
    if (yta) *yta = g->yta;
    if (info) *info = g->info_host;
    return GPX_OK;
}

int gpx_mg_timing(gpx_mg_t *g, double *ms8)
{
    MG_ENTER(g);
    GPX_ARG(g->fitted && ms8, "bad arguments");
    for (int i = 0; i < 8; ++i) ms8[i] = g->ms[i];
    return GPX_OK;
}

int gpx_mg_device_ptrs(gpx_mg_t *g, void **A, int64_t *ld)
{
    MG_ENTER(g);
    if (A) *A = g->A;
    if (ld) *ld = g->ld;
    return GPX_OK;
}

int gpx_mg_chain_by_panel(gpx_mg_t *g, double *ms, int64_t count)
{
    MG_ENTER(g);
    GPX_ARG(g->fitted && ms && count >= 0, "bad arguments");
    for (int64_t j = 0; j < count; ++j) ms[j] = (size_t)j < g->own_chain_ms.size() ? g->own_chain_ms[(size_t)j] : 0.0;
    return GPX_OK;
}

int gpx_mg_timing_ex(gpx_mg_t *g, double *ms, int count)
{
    MG_ENTER(g);
    GPX_ARG(g->fitted && ms && count >= 0, "bad arguments");
    for (int i = 0; i < count; ++i) ms[i] = i < GPX_MG_TIMING_N ? g->ms[i] : 0.0;
    return GPX_OK;
}

}  // extern "C"
