// gpx_panel.hip -- the RESIDENT panel kernel: ONE launch factors a panel of up to 256 columns.
//
// The panel chain of the blocked factorisation (gpx_potrf.hip) used to be, per 64 panel columns, the leaf, the
// row substitution and one to three in-panel updates: 3 - 5 dependent launches of 12 - 30 us each, ~ 325 us per
// 256-column panel, which is what bounds every factorisation with n <= 16384 and the panel owner's critical path
// of the multi-GPU schedule.  Here the panel lives in REGISTERS for the whole launch:
//
//   workgroup w owns rows r0 + 64 w .. + 64 of the panel; wave v of it keeps its 16 rows x kb columns as MFMA
//   accumulator tiles (kb = 256 fp64: 64 doubles a lane), loaded once and stored once.
//   Step j (64 columns, right-looking inside the panel):
//     workgroup j   (the diagonal block, fully updated by the steps before) factors it and forms W = inv(L_jj)
//                   in place in the accumulator tiles, on the MFMA pipe (gpx_leaf.h: factor64_mfma for fp64,
//                   factor64_mfma_t<float> for fp32 panels of up to 16384 rows and the chain part of taller ones;
//                   the lean fp32 instantiation keeps the VALU sweep factor64_pipe) --, stores L_jj,
//                   publishes W and raises flag W_j; it is done.
//     before step 0 the launch applies the kpre columns immediately to its left (left-looking pre-update): what
//                   used to be one more dependent GEMM launch in front of every panel.
//     workgroup w>j waits for W_j (one lane polls, bounded), stages W through LDS, X = P_j W^T on the MFMA pipe
//                   (its rows of column block j: final, stored), keeps X in LDS as an operand and updates its
//                   blocks c > j:  P_c -= X X_c^T, where X_c -- the same step's rows of the diagonal block c --
//                   is its own X when c == w and otherwise published by workgroup c (flag X_cj).
//   The only serial chain is  W_j -> (workgroup j+1: X, own update, leaf) -> W_{j+1}: the leaf plus two 64^3
//   products and one flag hand-off per 64 columns.
//
// Hand-off between workgroups (possibly on different XCDs, whose L2s are not coherent with each other): payload
// and flags are written with agent-scope (sc1, write-through) stores and read with agent-scope loads; the writer
// drains its stores (s_waitcnt vmcnt(0)), the workgroup meets, one lane stores the flag; the reader polls the
// flag, then loads.  No L2-wide write-back or invalidate is issued -- the trailing update of the previous panel
// runs on the same L2s with gigabytes of dirty tiles.  Flags hold a per-launch serial, nothing is reset.
// Producers have the lowest workgroup ids of their matrix and workgroups are dispatched in id order, so a
// producer is resident (or finished) before any of its consumers exists; should that ever fail the spin is
// bounded and the panel reports info = -7 instead of hanging.
#include "gpx_common.h"
#include "gpx_leaf.h"

#include <algorithm>
#include <optional>
#include <mutex>
#include <vector>
#include <stdlib.h>

namespace gpx {

constexpr int RES_MAXSTEPS = 4;                   // panels of up to 256 columns
constexpr int RES_SLOTS = 10;                     // W_0..3, X_10, X_20, X_21, X_30, X_31, X_32
constexpr int RES_FLAGS = 16;                     // flag words per matrix (10 used)
constexpr int RES_SPIN = 1 << 22;                 // polls (~1 us each with the sleep): gives up after a few seconds

__device__ __forceinline__ int res_xslot(int c, int j) { return 4 + c * (c - 1) / 2 + j; }

template <typename T> struct ResPitch { static constexpr int v = IB + 16 / (int)sizeof(T); };

// agent-scope element accesses of the published blocks
template <typename T> __device__ __forceinline__ void pub_store(T *p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> __device__ __forceinline__ T pub_load(const T *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// all of this workgroup's published stores are on their way: drain them, meet, raise the flag.
// Ordering.  Every published element and every flag is an AGENT-scope atomic access (sc1: write-through stores, loads
// that do not hit in a non-coherent L2), so the hardware half of a release is exactly the drain below -- on gfx9 stores
// count in vmcnt, and the count only drops when the write is acknowledged at the coherence point -- followed by the
// workgroup barrier, and the hardware half of an acquire is "the flag load has returned before the payload loads
// issue" (the poll loop consumes the value; the barrier follows).  What a formal release / acquire pair adds on
// gfx942 / gfx950 is buffer_wbl2 sc1 / buffer_inv sc1 -- write back and invalidate this XCD's whole L2 for the sake of
// NON-atomic data, of which the hand-off has none; the first version of the kernel paid that next to a trailing
// update with gigabytes of dirty tiles.  `strict` issues the formal pair anyway and has been the DEFAULT since the end of round 4
// (GPX_RES_STRICT=0 is the relaxed form argued above): with hand-offs that now raise one flag per block it costs nothing that
// can be measured beside a trailing update (N = 65536: 1.3742 vs 1.3748 s; n = 8192: 5.91 vs 5.83 ms, i.e. none) and 1.5 % where
// the chain is all there is (n = 2048: 0.933 vs 0.95 ms) -- the price of not resting the library's results on an argument about
// what sc1 accesses do.  The soak test (tests/test_gpu_round4.py) runs both forms and compares them bit for bit.
__device__ __forceinline__ void res_raise(int *flag, int serial, int strict)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (strict) __hip_atomic_store(flag, serial, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(flag, serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// one lane polls; everybody learns the outcome through LDS (the caller's next barrier orders the payload loads)
__device__ __forceinline__ void res_wait(const int *flag, int serial, int *s_ok, int naps, int strict)
{
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != serial && spins < RES_SPIN) {
            __builtin_amdgcn_s_sleep(2);
            if (naps) __builtin_amdgcn_s_sleep(8);
            ++spins;
        }
        if (spins >= RES_SPIN) *s_ok = 0;
        if (strict) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("" ::: "memory");
    }
}

// 16-byte agent-scope (sc1) accesses: half as many requests as the 8-byte atomics for the same block.  The compiler
// does not count inline-asm loads: pub_wait() both waits and makes every later use depend on it.
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4v pub_load16(const void *p)
{
    uint4v v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void pub_store16(void *p, uint4v v)
{
    // (s_nop 1: gfx940+ needs two wait states between a store of more than 64 bits and a VALU write to its data
    //  registers; the compiler's hazard recogniser does not look inside inline asm)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
template <int N> __device__ __forceinline__ void pub_wait(uint4v (&v)[N])
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}

// stage the 16 x 16 tiles on and below the diagonal of a published W = inv(L) (row-major 64 x 64; fp64: the tiles
// above are never read by res_prod_w) into LDS: 10 of 16 tiles, 16-byte loads
template <typename T, int PT>
__device__ __forceinline__ void res_stage_w(const T *__restrict__ src, T (*dst)[PT])
{
    constexpr int E = 16 / (int)sizeof(T), UPR = 16 / E, UPT = 16 * UPR, TOTAL = 10 * UPT, PER = (TOTAL + 255) / 256;
    uint4v v[PER];
    int row[PER], col[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        int u = threadIdx.x + 256 * i;
        if (u >= TOTAL) u = TOTAL - 1;                          // (duplicates the last unit: harmless)
        const int tile = u / UPT, w = u % UPT;
        // tiles (ti, tj), tj <= ti, numbered row by row: 0:(0,0) 1:(1,0) 2:(1,1) 3:(2,0) ...
        const int ti = tile < 1 ? 0 : tile < 3 ? 1 : tile < 6 ? 2 : 3, tj = tile - ti * (ti + 1) / 2;
        row[i] = 16 * ti + w / UPR; col[i] = 16 * tj + (w % UPR) * E;
        v[i] = pub_load16(src + row[i] * IB + col[i]);
    }
    pub_wait<PER>(v);
#pragma unroll
    for (int i = 0; i < PER; ++i) *reinterpret_cast<uint4v *>(&dst[row[i]][col[i]]) = v[i];
}

// x[0..3] += sa[16 wave + li][:] . W^T with W lower triangular (fp64: one k-chunk = one tile, tile (jj, kc) is zero for kc > jj)
template <typename T, int PT>
__device__ __forceinline__ void res_prod_w(const T (*sa)[PT], const T (*sb)[PT], typename PM<T>::v4 (&acc)[4], int wave, int li, int lq)
{
    typedef PM<T> M;
    constexpr int EPK = M::EPK, SUB = M::SUB, NCH = IB / EPK;
    static_assert(EPK == 16, "one k-chunk per 16 x 16 tile");
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc) {
        T fa[SUB], fb[4][SUB];
        load_frag32<T>(&sa[16 * wave + li][kc * EPK + lq * SUB], fa);
#pragma unroll
        for (int jj = kc; jj < 4; ++jj) load_frag32<T>(&sb[16 * jj + li][kc * EPK + lq * SUB], fb[jj]);
#pragma unroll
        for (int ss = 0; ss < SUB; ++ss)
#pragma unroll
            for (int jj = kc; jj < 4; ++jj) acc[jj] = M::mfma(fa[ss], fb[jj][ss], acc[jj]);
    }
}

// stage a published 64 x 64 row-major block into LDS (pitch PT)
template <typename T, int PT>
__device__ __forceinline__ void res_stage(const T *__restrict__ src, T (*dst)[PT])
{
    constexpr int PER = IB * IB / 256;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int idx = threadIdx.x + 256 * i;
        dst[idx / IB][idx % IB] = pub_load(src + idx);
    }
}

// the same in two halves -- global -> registers, registers -> LDS -- so that the next block's loads can be in flight under
// the current block's products (the LV = 4 instantiation has the registers for it)
template <typename T> struct StageRegs {
    static constexpr int E = 16 / (int)sizeof(T), RPW = 256 / (IB / E), PER = IB / RPW;
    typedef uint4v Q;                                           // (a vector type, not a struct of words: stays in registers)
    Q q[PER];
};
template <typename T>
__device__ __forceinline__ void res_stage_rows_load(const T *__restrict__ src, int lda, int nvalid, StageRegs<T> &rg)
{
    typedef StageRegs<T> S;
    const int row0 = threadIdx.x / (IB / S::E), col = (threadIdx.x % (IB / S::E)) * S::E;
#pragma unroll
    for (int i = 0; i < S::PER; ++i) {
        const int row = row0 + S::RPW * i;
        const int rr = row < nvalid ? row : nvalid - 1;
        rg.q[i] = *reinterpret_cast<const typename S::Q *>(src + (unsigned)(rr * lda + col));
    }
}
template <typename T, int PT>
__device__ __forceinline__ void res_stage_rows_store(const StageRegs<T> &rg, T (*dst)[PT])
{
    typedef StageRegs<T> S;
    const int row0 = threadIdx.x / (IB / S::E), col = (threadIdx.x % (IB / S::E)) * S::E;
#pragma unroll
    for (int i = 0; i < S::PER; ++i) *reinterpret_cast<typename S::Q *>(&dst[row0 + S::RPW * i][col]) = rg.q[i];
}

// stage 64 rows x 64 columns of the matrix (rows lda apart, 16-byte aligned; rows >= nvalid repeat the last valid one)
// into LDS.  Offsets inside the block fit 32 bits whatever the matrix size.
template <typename T, int PT>
__device__ __forceinline__ void res_stage_rows(const T *__restrict__ src, int lda, int nvalid, T (*dst)[PT])
{
    constexpr int E = 16 / (int)sizeof(T), RPW = 256 / (IB / E), PER = IB / RPW;     // elements per 16 bytes; rows per pass
    struct alignas(16) Q { unsigned u[4]; };
    const int row0 = threadIdx.x / (IB / E), col = (threadIdx.x % (IB / E)) * E;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int row = row0 + RPW * i;
        const int rr = row < nvalid ? row : nvalid - 1;
        *reinterpret_cast<Q *>(&dst[row][col]) = *reinterpret_cast<const Q *>(src + (unsigned)(rr * lda + col));
    }
}

// acc[0..3] (+)= (+-) sa[16 wave + li][:] . sb[16 jj + li][:]^T over the 64 columns
template <typename T, int PT, bool NEG>
__device__ __forceinline__ void res_prod(const T (*sa)[PT], const T (*sb)[PT], typename PM<T>::v4 (&acc)[4], int wave, int li, int lq)
{
    typedef PM<T> M;
    constexpr int EPK = M::EPK, SUB = M::SUB, NCH = IB / EPK;
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc) {
        T fa[SUB], fb[4][SUB];
        load_frag32<T>(&sa[16 * wave + li][kc * EPK + lq * SUB], fa);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) load_frag32<T>(&sb[16 * jj + li][kc * EPK + lq * SUB], fb[jj]);
        if (NEG) {
#pragma unroll
            for (int ss = 0; ss < SUB; ++ss) fa[ss] = -fa[ss];
        }
#pragma unroll
        for (int ss = 0; ss < SUB; ++ss)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[jj] = M::mfma(fa[ss], fb[jj][ss], acc[jj]);
    }
}

// LEAF_MFMA: the diagonal block's leaf on the MFMA pipe (fp64: always; fp32: the short-panel instantiation -- the
// unrolled MFMA sweep costs the fp32 kernel 58 VGPRs, i.e. a third workgroup per CU, which tall panels miss more than
// they gain from the faster leaf)
template <typename T, bool LEAF_MFMA, int LV = 1>
__global__ __launch_bounds__(256, LV == 4 ? 1 : 2) void panel_res_kernel(T *__restrict__ A, int64_t lda, int64_t n, int64_t r0, int64_t c0,
                                                           int nsteps, int *__restrict__ info, T *__restrict__ pub,
                                                           int *__restrict__ flags, int serial, int64_t sM, int kpre,
                                                           unsigned long long *__restrict__ stamps, int w0, int strict)
{
    typedef PM<T> M;
    typedef typename M::v4 v4;
    constexpr int PT = ResPitch<T>::v;
    __shared__ __attribute__((aligned(16))) T sA[IB][PT];      // X of the step (A operand; B operand of the own-diagonal update)
    __shared__ __attribute__((aligned(16))) T sB[IB][PT];      // staged published block
    __shared__ int s_ok;
    A += (int64_t)blockIdx.y * sM;
    info += blockIdx.y;
    pub += (int64_t)blockIdx.y * (RES_SLOTS * IB * IB);
    flags += (int64_t)blockIdx.y * RES_FLAGS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int w = blockIdx.x + w0;                            // (w0 > 0: the rows of a panel launched in two parts, see panel_res_t)
    const bool future_diag = w < nsteps;
    // diagnostic (gpx_debug_panel_stamps): 8 words per workgroup: start, pre-update done, steps 0..3 done, end, hardware id
    auto stamp = [&](int slot) {
        if (stamps && tid == 0 && blockIdx.y == 0) stamps[(size_t)w * 16 + slot] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    if (stamps && tid == 0 && blockIdx.y == 0) stamps[(size_t)w * 16 + 7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    if (future_diag) __builtin_amdgcn_s_setprio(3);
    if (tid == 0) s_ok = (*info == 0);
    __syncthreads();
    if (!s_ok) {
        // an earlier panel failed: nobody computes, but nobody may be left waiting either
        if (future_diag && tid < RES_FLAGS) __hip_atomic_store(flags + tid, serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int64_t wr0 = r0 + (int64_t)w * IB + 16 * wave;      // this wave's 16 rows
    // this lane's four rows of the panel as 32-bit element offsets from the workgroup's first row (64 rows of at most 2^24
    // elements each; rows beyond the matrix repeat its last one on the way in and are not stored): a uniform base pointer
    // and four VGPRs instead of four 64-bit addresses held across the whole step loop (the LV = 5 instantiation spilled 11)
    T *__restrict__ const Aw = A + (r0 + (int64_t)w * IB) * lda + c0;
    unsigned roff[4];
    bool rin[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t gr = wr0 + M::row(lane, r);
        rin[r] = gr < n;
        roff[r] = (unsigned)(((rin[r] ? gr : n - 1) - (r0 + (int64_t)w * IB)) * lda) + li;
    }
    // ---- the panel rows into accumulator tiles: acc[c][jj] = rows x columns 64 c + 16 jj .. ----
    v4 acc[RES_MAXSTEPS][4];
#pragma unroll
    for (int c = 0; c < RES_MAXSTEPS; ++c) {
        const bool have = c < nsteps && (!future_diag || c <= w);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[c][jj][r] = have ? Aw[roff[r] + (unsigned)(IB * c + 16 * jj)] : (T)0;
    }
    // ---- left-looking pre-update with the kpre blocks of 64 columns immediately to the left of the panel (the
    // previous panel of a small factorisation, or the left half of a wider panel): what used to be one more
    // dependent GEMM launch in front of every panel.  P_c -= R R_c^T, R = my rows, R_c = the rows of diagonal block c.
    // Workgroup 0 needs block 0 only (its own rows are R_0): kpre short products before the first leaf can start;
    // the later diagonal blocks' longer pre-updates run under the leaves before them. ----
    if (kpre > 0) {
        const int nvalid = (int)(n - (r0 + (int64_t)w * IB) < IB ? n - (r0 + (int64_t)w * IB) : IB);
        const T *mine = A + (r0 + (int64_t)w * IB) * lda;      // my 64 rows, column 0
        // (LV == 4: the loads of my rows' NEXT 64 columns are issued before this block's products and land under them:
        //  workgroup 0's four short products then follow each other at the matrix pipe's pace -- they are the start of the chain)
        //  the rows of the diagonal blocks, the other operand, are fetched one block ahead in the same way: a workgroup's 4 x kpre
        //  products then run at the matrix pipe's pace too -- the last diagonal workgroup's pre-update, 50 us with every
        //  block's load latency in the open, is what the panel waited for once the leaf was fast)
        constexpr bool PREFETCH = LV == 4;
        StageRegs<T> nxt, nb;
        const int cmax = future_diag ? (w + 1 < nsteps ? w + 1 : nsteps) : nsteps;        // blocks c < cmax take part
        int lkc = 0, lc = -1;                                   // the B block (lkc, lc) that is in flight into nb
        auto advance = [&]() { do { if (++lc >= cmax) { lc = 0; ++lkc; } } while (lkc < kpre && lc == w); };
        auto bsrc = [&](int kc_, int c_) { return A + (r0 + (int64_t)IB * c_) * lda + (c0 - (int64_t)IB * (kpre - kc_)); };
        if constexpr (PREFETCH) {
            res_stage_rows_load<T>(mine + (c0 - (int64_t)IB * kpre), (int)lda, nvalid, nxt);
            advance();
            if (lkc < kpre) res_stage_rows_load<T>(bsrc(lkc, lc), (int)lda, IB, nb);
        }
        for (int kc = 0; kc < kpre; ++kc) {
            const int64_t pc = c0 - (int64_t)IB * (kpre - kc);
            __syncthreads();
            if constexpr (PREFETCH) {
                res_stage_rows_store<T, PT>(nxt, sA);
                if (kc + 1 < kpre) res_stage_rows_load<T>(mine + pc + IB, (int)lda, nvalid, nxt);
            } else {
                res_stage_rows<T, PT>(mine + pc, (int)lda, nvalid, sA);
            }
#pragma unroll
            for (int c = 0; c < RES_MAXSTEPS; ++c) {
                if (c >= nsteps || (future_diag && c > w)) break;
                if (c == w) {
                    __syncthreads();
                    res_prod<T, PT, true>(sA, sA, acc[c], wave, li, lq);
                    continue;
                }
                __syncthreads();                                // sA is in; the previous block's sB has been consumed
                if constexpr (PREFETCH) {
                    res_stage_rows_store<T, PT>(nb, sB);        // (block (kc, c): the loads were issued in this loop's order)
                    advance();
                    if (lkc < kpre) res_stage_rows_load<T>(bsrc(lkc, lc), (int)lda, IB, nb);
                } else {
                    res_stage_rows<T, PT>(A + (r0 + (int64_t)IB * c) * lda + pc, (int)lda, IB, sB);
                }
                __syncthreads();
                res_prod<T, PT, true>(sA, sB, acc[c], wave, li, lq);
            }
        }
    }
    stamp(1);
    v4 dg[4];                                                   // the diagonal block of this workgroup when its step comes
    int dg_step = -1;
#pragma unroll
    for (int j = 0; j < RES_MAXSTEPS; ++j) {
        if (j >= nsteps || dg_step >= 0) continue;              // (no break: one loop exit, so the unrolled body keeps its static indices)
        if (j > 0) stamp(1 + j);
        __syncthreads();                                        // sA / sB of the previous step are consumed
        // column block j of my rows -> LDS by rows (operand of the substitution, or the leaf's input)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) sA[16 * wave + M::row(lane, r)][16 * jj + li] = acc[j][jj][r];
        if (LEAF_MFMA && w == j) {                              // the MFMA leaf, after the loop (ONE copy of its code)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) dg[jj] = acc[j][jj];
            dg_step = j;
            continue;
        }
        if constexpr (!LEAF_MFMA) {
        if (w == j) {
            // ---- the diagonal block: leaf ----
            __syncthreads();
            // the leaf with the pivot wave (gpx_leaf.h): waves 0-2 hold the strictly-lower tiles and X, wave 3 the
            // diagonal tiles, factored one step ahead of the others' rank-4 update
            static_assert(IB == 64, "leaf roles assume a 64 x 64 block");
            const LeafRole ro = leaf_role_256(tid);
            const bool own_a = ro.pivot ? ro.t < IB / 4 : ro.tc < ro.tr;
            const bool own_x = !ro.pivot && ro.tr >= 0;
            const int tr = ro.pivot ? ro.t : ro.tr, tc = ro.pivot ? ro.t : ro.tc;
            T a[4][4], x[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int row = 4 * tr + r, col = 4 * tc + c;
                    a[r][c] = (own_a && col <= row) ? sA[row][col] : (T)0;
                    x[r][c] = (row == col) ? (T)1 : (T)0;
                }
            factor64_pipe<T, true>(a, x, IB, r0 + (int64_t)IB * j, info, ro);      // (r0 = the global index of the panel's first pivot)
            T *W = pub + (int64_t)j * (IB * IB);
            if (own_x) {                                         // (tiles above the diagonal: zero since the buffer was cleared)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    T rowv[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) rowv[c] = (4 * tc + c <= 4 * tr + r) ? x[r][c] : (T)0;
                    constexpr int E = 16 / (int)sizeof(T);
#pragma unroll
                    for (int h = 0; h < 4 / E; ++h) {
                        uint4v q; memcpy(&q, &rowv[h * E], 16);
                        pub_store16(W + (4 * tr + r) * IB + 4 * tc + h * E, q);
                    }
                }
            }
            res_raise(flags + j, serial, strict);
            if (own_a) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int row = 4 * tr + r, col = 4 * tc + c;
                        if (col <= row) A[(r0 + (int64_t)IB * j + row) * lda + c0 + IB * j + col] = a[r][c];
                    }
            }
            stamp(6);
            return;
        }
        }
        // ---- rows below the diagonal block ----
        res_wait(flags + j, serial, &s_ok, !future_diag, strict);
        __syncthreads();
        if (w == j + 1) stamp(8);
        if (!s_ok) {
            if (tid == 0) atomicCAS(info, 0, -7);
            if (future_diag && tid < RES_FLAGS) __hip_atomic_store(flags + tid, serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        // everybody wants W_j at once; only workgroup j + 1 is the chain: the others give its loads a head start
        if (!future_diag) __builtin_amdgcn_s_sleep(48);
        if constexpr (sizeof(T) == 8) res_stage_w<T, PT>(pub + (int64_t)j * (IB * IB), sB);
        else res_stage<T, PT>(pub + (int64_t)j * (IB * IB), sB);
        __syncthreads();
        if (w == j + 1) stamp(9);
        v4 x[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[jj][r] = (T)0;
        if constexpr (sizeof(T) == 8) res_prod_w<T, PT>(sA, sB, x, wave, li, lq);
        else res_prod<T, PT, false>(sA, sB, x, wave, li, lq);
        // X is final: store it, keep it in LDS as an operand, publish it when these are the rows of a later diagonal block
        T *Xp = future_diag ? pub + (int64_t)res_xslot(w, j) * (IB * IB) : nullptr;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lr = 16 * wave + M::row(lane, r);
                sA[lr][16 * jj + li] = x[jj][r];
                if (future_diag) pub_store(Xp + lr * IB + 16 * jj + li, x[jj][r]);
            }
        __syncthreads();                                        // sA is complete
        if (w == j + 1) stamp(10);
        // ---- right-looking updates inside the panel; the own diagonal block first (it is the chain), and the
        // publication's drain hides under its MFMAs ----
        if (future_diag) {
#pragma unroll
            for (int c = 1; c < RES_MAXSTEPS; ++c)
                if (c == w) res_prod<T, PT, true>(sA, sA, acc[c], wave, li, lq);
            if (w == j + 1) stamp(11);
            res_raise(flags + res_xslot(w, j), serial, strict);
            if (w == j + 1) stamp(12);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (rin[r]) Aw[roff[r] + (unsigned)(IB * j + 16 * jj)] = x[jj][r];
#pragma unroll
        for (int c = j + 1; c < RES_MAXSTEPS; ++c) {
            if (c >= nsteps || (future_diag && c >= w)) break;
            res_wait(flags + res_xslot(c, j), serial, &s_ok, 1, strict);
            __syncthreads();                                    // also: the previous block's sB has been consumed
            if (!s_ok) {
                if (tid == 0) atomicCAS(info, 0, -7);
                if (future_diag && tid < RES_FLAGS) __hip_atomic_store(flags + tid, serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            res_stage<T, PT>(pub + (int64_t)res_xslot(c, j) * (IB * IB), sB);
            __syncthreads();
            res_prod<T, PT, true>(sA, sB, acc[c], wave, li, lq);
        }
    }
    if constexpr (LEAF_MFMA && (LV == 4 || LV == 5)) {
        if (dg_step >= 0) {
            // ---- the diagonal block: the leaf of gpx_leaf.h (factor64_wave): the block is in sA by rows; wave 0 factors it
            // without a barrier, wave 1 follows it with the inverse; L is left in sA and W = inv(L) in sB.  (fp32 panels use the
            // same fp64 leaf: it converts on the way in and out of LDS.) ----
            constexpr int RING = LV == 4 ? IB / 4 : 4;            // LV = 5 runs two workgroups a CU: a 10 KB ring instead of 40 KB
            __shared__ __attribute__((aligned(16))) double sLeaf[RING * W1_SLOTS * 64];   // wave 0's per-step operands for wave 1
            __shared__ int sLeafCtl[2];
            const int j = dg_step;
            if (tid < 2) sLeafCtl[tid] = 0;
            __syncthreads();                                      // sA is complete
            if (wave < 2)
                factor64_wave<PT, RING, -1, T>(sA, sB, sLeaf, sLeafCtl, wave, r0 + (int64_t)IB * j, info, lane,
                                               (stamps && w == 0 && blockIdx.y == 0) ? stamps + 2040 * 16 : nullptr);
            __syncthreads();
            // publish W: its 10 lower 16 x 16 tiles, 16-byte agent-scope stores (fp64: what res_stage_w loads; fp32: res_stage
            // loads all 16 tiles, the six above the diagonal are zero since the buffer was cleared)
            T *W = pub + (int64_t)j * (IB * IB);
            constexpr int E = 16 / (int)sizeof(T), UPR = 16 / E;   // elements per 16 bytes; units per tile row
            constexpr int UPT = 16 * UPR, TOTAL = 10 * UPT;         // 16-byte units per tile, lower tiles
#pragma unroll
            for (int i = 0; i < (TOTAL + 255) / 256; ++i) {
                const int u = tid + 256 * i;
                if (u < TOTAL) {
                    const int tile = u / UPT, wq = u % UPT;
                    const int ti = tile < 1 ? 0 : tile < 3 ? 1 : tile < 6 ? 2 : 3, tj = tile - ti * (ti + 1) / 2;
                    const int row = 16 * ti + wq / UPR, col = 16 * tj + (wq % UPR) * E;
                    pub_store16(W + row * IB + col, *reinterpret_cast<const uint4v *>(&sB[row][col]));
                }
            }
            res_raise(flags + j, serial, strict);
#pragma unroll
            for (int i = 0; i < IB * IB / 256; ++i) {
                const int idx = tid + 256 * i, row = idx / IB, col = idx % IB;
                if (col <= row) A[(r0 + (int64_t)IB * j + row) * lda + c0 + IB * j + col] = sA[row][col];
            }
        }
    } else
    if constexpr (LEAF_MFMA) if (dg_step >= 0) {
        // ---- the diagonal block: the leaf on the MFMA pipe (gpx_leaf.h), in place in the accumulator tiles ----
        const int j = dg_step;
        v4 xw[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) xw[jj][r] = (16 * wave + M::row(lane, r) == 16 * jj + li) ? (T)1 : (T)0;
            LeafMfma<T>::run(dg, xw, r0 + (int64_t)IB * j, info, wave, lane, (stamps && w == 0 && blockIdx.y == 0) ? stamps + 2040 * 16 : nullptr);
        T *W = pub + (int64_t)j * (IB * IB);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * wave + M::row(lane, r), col = 16 * jj + li;
                if (jj <= wave) pub_store(W + row * IB + col, (col <= row) ? xw[jj][r] : (T)0);   // (tiles above: zero since the clear)
            }
        res_raise(flags + j, serial, strict);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * wave + M::row(lane, r), col = 16 * jj + li;
                if (col <= row) A[(r0 + (int64_t)IB * j + row) * lda + c0 + IB * j + col] = dg[jj][r];
            }
    }
    stamp(6);
}


static unsigned long long *g_res_stamps = nullptr;             // diagnostic: the launch `g_res_stamp_at` launches from now records
static int g_res_stamp_at = -1;

// per host thread and device: the published blocks and flag words of every matrix of a batch
// The blocks and flag words are ONE set per host thread and device, told apart between launches only by the serial:
// two resident launches of a thread must never be in flight at once.  Launches on one stream are ordered anyway (the
// factorisation's panels all go to the thread's look-ahead stream); when the stream CHANGES (a single-panel matrix on
// the caller's stream, gpx_d_potrf_panel on any stream, the multi-GPU schedule's panel stream, an asynchronous
// gpx_gp_fit followed by another handle's) the new launch waits for the previous one through `ev`.
struct ResScratch {
    void *p = nullptr; size_t bytes = 0; int nbatch = 0; int serial = 0;
    hipEvent_t ev = nullptr;        // after the last launch that was NOT on the look-ahead stream / recorded on demand on it
    hipStream_t last = nullptr; bool have_last = false, last_on_side = false;
};
constexpr int RES_MAXDEV = 16;
static thread_local ResScratch g_res_dev[RES_MAXDEV];       // one per device: a thread that alternates between GPUs keeps both
#define RES_TRACE(...) do { if (tune().trace) { fprintf(stderr, "[gpx] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)
static int res_scratch(int nbatch, size_t es, void **pub, int **flags, ResScratch **out)
{
    int dev = 0;
    GPX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= RES_MAXDEV) { set_error("resident panel: device index %d out of range", dev); return GPX_ERR_ARG; }
    ResScratch &g = g_res_dev[dev];
    if (g.nbatch < nbatch) {
        RES_TRACE("res_scratch: grow %d -> %d matrices (device %d, old %p)", g.nbatch, nbatch, dev, g.p);
        if (g.p) { GPX_HIP(hipDeviceSynchronize()); (void)hipFree(g.p); }
        g.p = nullptr; g.bytes = 0; g.nbatch = 0;
        const size_t fbytes = ((size_t)nbatch * RES_FLAGS * sizeof(int) + 255) / 256 * 256;
        const size_t region = (size_t)nbatch * RES_SLOTS * IB * IB * 8;
        const size_t need = fbytes + region + region / 2;
        GPX_HIP(hipMalloc(&g.p, need));
        GPX_HIP(hipMemset(g.p, 0, need));
        GPX_HIP(hipDeviceSynchronize());
        RES_TRACE("res_scratch: new block %p, %zu bytes, cleared", g.p, need);
        g.bytes = need;
        g.nbatch = nbatch;
        g.serial = 0;
    }
    // the layout follows the CAPACITY, not this call's batch: a slot keeps its address and its role for the life of
    // the buffer (fp64 blocks first, the fp32 ones behind them) -- the tiles above a W block's diagonal are zero from
    // the initial clear and never rewritten
    const size_t fbytes = ((size_t)g.nbatch * RES_FLAGS * sizeof(int) + 255) / 256 * 256;
    const size_t region = (size_t)g.nbatch * RES_SLOTS * IB * IB * 8;
    if (g.serial >= (1 << 30)) {
        // the flags hold the serial of the launch that raised them; long before the counter could wrap, start over
        // (everything that used the block has to be done first)
        GPX_HIP(hipDeviceSynchronize());
        GPX_HIP(hipMemset(g.p, 0, fbytes));
        GPX_HIP(hipDeviceSynchronize());
        g.serial = 0;
    }
    *flags = (int *)g.p;
    *pub = (char *)g.p + fbytes + (es == 8 ? 0 : region);
    *out = &g;
    return GPX_OK;
}

int64_t panel_res_max()
{
    const int64_t v = tune().potrf_res;
    return std::min<int64_t>(v, (int64_t)RES_MAXSTEPS * IB);
}

// ---- the asm-scheduled leaf is checked on the device it runs on --------------------------------------------------------
// factor64_wave (gpx_leaf.h) issues its MFMAs from volatile asm with wait states that were MEASURED on gfx950
// (tools/mfma_hazard_probe_gen.py), not taken from a hazard table the compiler maintains; a miss would not fault, it would
// leave about single precision in rows 12 .. 15 of some tiles.  Before the first panel of a process uses it on a device,
// both of its instantiations (LV = 4: tiles in AGPRs, LV = 5: in VGPRs; fp64 and fp32 storage) factor a full-mantissa
// 256 x 256 panel -- four leaves, three hand-offs -- BESIDE a product that keeps every matrix pipe busy (the probe's
// thresholds moved under contention), and the result is compared with the compiler-scheduled MFMA leaf (LV = 1, whose
// hazards hipcc handles) on the same input: agreement to 1e-12 relative or the library falls back to LV = 1 for the rest of
// the process and says so on stderr (gpx_debug_leaf_selfcheck: 1 passed, 2 failed -> fallback, 0 not run yet).
static thread_local int g_leaf_force = 0;                      // the self-check's own launches: 1 / 4 / 5, no check
static std::mutex g_leaf_mu;
static int g_leaf_state[RES_MAXDEV] = {};                      // 0 unknown, 1 ok, 2 failed, 3 never verified (guarded by g_leaf_mu)
static int g_leaf_tries[RES_MAXDEV] = {};                      // attempts that could not run (guarded by g_leaf_mu)
constexpr int LEAF_CHECK_TRIES = 3;
static int leaf_selfcheck_run(int dev, hipStream_t st);
// true: the asm-scheduled leaf may be used on the current device.  While the check has not PASSED the compiler-scheduled
// leaf is used (round 6: "could not run" used to count as a pass -- under HBM pressure the unverified leaf ran, and every
// panel launch repeated the attempt, host matrix, allocations and syncs included, on the factorisation's critical path);
// after LEAF_CHECK_TRIES attempts that could not run (no memory for its ~135 MB, a stream capture open) the device keeps
// the compiler-scheduled leaf for the rest of the process and says so once.
static bool leaf_asm_ok(hipStream_t st = nullptr)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (dev < 0 || dev >= RES_MAXDEV) return false;
    std::lock_guard<std::mutex> lk(g_leaf_mu);
    if (g_leaf_state[dev] == 0) {
        ++g_prof_mute;
        const int rc = leaf_selfcheck_run(dev, st);
        --g_prof_mute;
        if (rc == 1 || rc == 2) g_leaf_state[dev] = rc;
        else if (++g_leaf_tries[dev] >= LEAF_CHECK_TRIES) g_leaf_state[dev] = 3;
        if (rc == 2)
            fprintf(stderr, "[gpx] WARNING: the asm-scheduled MFMA leaf failed its self-check on device %d: "
                            "falling back to the compiler-scheduled leaf (GPX_LEAF=1)\n", dev);
        else if (g_leaf_state[dev] == 3)
            fprintf(stderr, "[gpx] NOTE: the asm-scheduled MFMA leaf could not be verified on device %d (%d attempts: no memory "
                            "for the check, or a stream capture was open): keeping the compiler-scheduled leaf (GPX_LEAF=1)\n",
                    dev, LEAF_CHECK_TRIES);
    }
    return g_leaf_state[dev] == 1;
}

template <typename T>
static int panel_res_t(T *A, int64_t lda, int64_t n, int64_t r0, int64_t c0, int64_t kb, int *info_dev, hipStream_t st,
                       const Batch *bt, int64_t kpre, hipEvent_t done)
{
    const int nbatch = bt ? bt->count : 1;
    void *pub = nullptr; int *flags = nullptr;
    ResScratch *scr = nullptr;
    GPX_TRY(res_scratch(nbatch, sizeof(T), &pub, &flags, &scr));
    const int64_t rows = n - r0;
    dim3 grid((unsigned)cdiv(rows, IB), (unsigned)nbatch);
    const double kd = (double)kb;
    RES_TRACE("panel_res: n %lld r0 %lld kb %lld kpre %lld batch %d grid %u serial %d", (long long)n, (long long)r0, (long long)kb,
              (long long)kpre, nbatch, grid.x, scr->serial + 1);
    const hipStream_t side = potrf_side_stream();
    if (!scr->ev) GPX_HIP(hipEventCreateWithFlags(&scr->ev, hipEventDisableTiming));
    if (scr->have_last && scr->last != st) {
        if (scr->last_on_side) {
            if (side == scr->last) GPX_HIP(hipEventRecord(scr->ev, side));     // everything queued there so far
            else GPX_HIP(hipDeviceSynchronize());                               // (that stream went away with its device context)
        }
        GPX_HIP(hipStreamWaitEvent(st, scr->ev, 0));
    }
    unsigned long long *stamps = (g_res_stamps && g_res_stamp_at-- == 0) ? g_res_stamps : (unsigned long long *)nullptr;
    const int serial = ++scr->serial;
    const int nsteps = (int)(kb / IB);
    const int strict = tune().res_strict != 0 ? 1 : 0;   // formal release / acquire hand-offs (see res_raise); default since round 4
    // TALL panels go out as TWO launches on the same stream: first the diagonal workgroups alone (the chain of leaves),
    // then all the rows below.  In one launch the row workgroups sit on their CUs for the whole chain -- ~230 us at
    // 32768 rows, of which they compute for ~30 -- and each of them keeps a trailing-update workgroup of the other
    // stream off its CU meanwhile (254 VGPRs: two workgroups fill a CU).  Launched after the chain they find every
    // flag raised, never wait, and hold their CUs for a tenth of the time.  The panel's own latency grows by the
    // row pass (it no longer hides under the chain), which only matters where the panel chain is the critical path:
    // short panels (rows <= GPX_POTRF_TWO_PART_ROWS) stay one launch.
    // (measured: n = 32768 fp32 96.7 -> 95.5 ms with 16384; n = 65536 fp64 unchanged; n = 16384 28.8 -> 29.3 with 12288)
    const int64_t two_part_rows = tune().two_part_rows;
    // fp32: which instantiation holds the leaf (see panel_res_kernel): the MFMA leaf for single launches of up to
    // GPX_LEAF_MFMA_F32_ROWS rows and for the chain part of a two-part panel; the lean kernel for everything taller
    constexpr bool F64 = sizeof(T) == 8;
    const bool mfma_single = F64 || rows <= tune().leaf_mfma_f32_rows;
    // (`done` is recorded behind the launch.  Carrying it as the completion signal of the dispatch packet itself --
    //  hipExtLaunchKernelGGL(..., stopEvent) -- saves ~2.5 us per panel (tools/sync_probe.hip) and was tried; one run of
    //  the parity suite then produced a wrong log_lh at n = 1990 that never reproduced.  Not worth 1 %: dropped.)
    hipEvent_t record_after = done;
    // lock-step batches: the same trade per launch over ALL matrices -- many matrices' row workgroups crowd out the
    // update of the step before (64 x n = 8192: 0.213 -> 0.199 s with two parts, 16 x: 55.0 -> 54.6 ms, 8 x: 29.4 -> 30.8)
    const int64_t nmat = bt ? bt->count : 1;
    const bool two_part = nmat > 1 ? rows * nmat > tune().two_part_batch : rows > two_part_rows;
    // fp64 leaf (gpx_leaf.h): GPX_LEAF = 4 the two-wave leaf without barriers (default for single matrices' short panels), 1 the
    // round-3 leaf on four waves.  (The LV = 4 instantiation runs ONE workgroup per CU -- its accumulator tiles and 107 KB of
    // LDS -- which a single matrix's panels never notice (at most 256 workgroups per launch) but a lock-step batch's would:
    // batches keep 1.)
    const int64_t excl_rows = tune().panel_excl_rows;
    const bool idle_chip = potrf_take_idle_chip_hint();        // (always taken: a hint is for ONE launch)
    const bool excl = !bt && idle_chip && rows <= excl_rows;
    // (panels of up to GPX_LEAF4_ROWS rows take the LV = 4 instantiation beside an update too: n = 8192 5.94 -> 5.79 ms; taller
    //  ones lose more CUs to its one-workgroup-per-CU footprint than the leaf gives back: n = 16384 27.85 -> 28.4 ms)
    // (Several host threads factoring at once -- 4 x 128 one-per-CU workgroups wanting 256 CUs -- cannot deadlock: inside a
    //  launch workgroups are dispatched in id order, a consumer is only ever placed after its producers (ids 0 .. 3), and a
    //  producer waits for nothing but earlier producers of its own launch; what is not placed yet simply waits for a CU.
    //  tests/test_gpu_round5.py::test_four_host_threads_factor_n8192_concurrently holds it: 24 fits, no -7, bit-identical.)
    const int64_t leaf_dflt = (excl || (!bt && rows <= tune().leaf4_rows[F64 ? 0 : 1])) ? 4 : 1;
    // (the asm-scheduled leaf only where it has passed its self-check on this device; the check's own launches force a level)
    const bool asm_ok = g_leaf_force != 0 || leaf_asm_ok(st);
    const int64_t leaf_want = g_leaf_force ? g_leaf_force : (tune().leaf_set ? tune().leaf : -1);
    const bool v4 = asm_ok && (leaf_want >= 0 ? leaf_want : leaf_dflt) == 4;   // (fp32: N = 32768 94.5 -> 94.2 ms with 5120)
    // A CU of its own for every workgroup of a SHORT panel (single matrix, rows <= GPX_PANEL_EXCL_ROWS).  Per-step stamps
    // of every leaf variant say the same thing (profiles/r04_leaf_steps_*.log): a leaf step takes 3 - 4 times longer while
    // workgroups of the trailing update share the CU (matrix pipe, issue slots) -- a panel took 120 us alone and 140 - 230 us
    // beside an update.  The dispatcher cannot be told to keep a CU free, but it cannot place what does not fit: the LV = 4
    // instantiation holds more registers a lane than leave room for the 128 x 128 update kernel (234), and its 107 KB of
    // static LDS (panel blocks + the leaf's 40 KB of per-step operands) plus GPX_PANEL_PAD_LDS = 8 KB of unused dynamic LDS
    // leave less than the 128 x 64 update kernel's 48 KB of the CU's 160.  Panels start on an idle chip, a few us before the
    // update they run beside (potrf()'s launch order), so they get their CUs and keep them: the same panel time in EVERY phase
    // (profiles/r04_timeline_n8192_excl8192.txt).  Each workgroup takes a whole CU from the update for as long as the
    // chain runs, so it pays only while the panel is short: n = 8192 potrf 5.75 -> 5.57 ms with 5120 rows, 5.68 with all
    // (profiles/r04_ab_exclusive_cus.log, measured with the first one-wave leaf).
    size_t pad_lds = 0;
    if (v4 && excl) {
        pad_lds = (size_t)tune().panel_pad_lds[F64 ? 0 : 1];   // (fp32: 75 KB of static LDS)
        GPX_TRY(set_max_lds((const void *)panel_res_kernel<T, true, 4>, (int)pad_lds));
    }
#define GPX_PANEL_LAUNCH_LDS(KERNEL, GRID, W0, DYN)                                                                             \
    hipLaunchKernelGGL((KERNEL), GRID, dim3(256), DYN, st, A, lda, n, r0, c0, nsteps, info_dev, (T *)pub, flags, serial,       \
                       bt ? bt->sA : (int64_t)0, (int)(kpre / IB), stamps, W0, strict)
#define GPX_PANEL_LAUNCH(KERNEL, GRID, W0) GPX_PANEL_LAUNCH_LDS(KERNEL, GRID, W0, 0)
    // every other fp64 panel (lock-step batches, panels taller than GPX_LEAF4_ROWS): the same leaf in the LV = 5 instantiation --
    // two workgroups a CU like the round-3 kernel (a 10 KB ring of operand slots instead of 40 KB, no operand prefetch)
    // (fp32 panels hand their diagonal blocks to the same fp64 leaf wherever they used the fp32 MFMA leaf: GPX_LEAF=1 keeps that one)
    const bool v5 = asm_ok && !v4 && (leaf_want >= 0 ? leaf_want : 5) == 5;
    // (Round 5's "tall" route -- only the diagonal workgroups on the resident kernel, the rows below as products with
    // inv(L_256) on the GEMM kernel -- was measured neutral at every size (N = 65536 fp64 1.3662 vs 1.3668 s, N = 32768 fp32
    // 94.97 vs 94.38 ms: what the update loses in situ is the panel stream's FLOPS on the same matrix pipes, not the shape
    // of the kernels that carry them; profiles/r05_ab_tall_*.log) and removed in round 6 with its inverse kernel.)
    const double res_rows = (double)rows;
    std::optional<ProfScope> prof;
    prof.emplace(PC_POTRF_DIAG, (kd * kd * kd / 3.0 + (res_rows - (double)kb) * kd * kd + 2.0 * res_rows * kd * (double)kpre) * nbatch, st);
    if (two_part && (int64_t)grid.x > nsteps) {
        const bool mfma_chain = F64 || tune().leaf_mfma_f32_rows > 0;
        const dim3 gdiag((unsigned)nsteps, grid.y), grows(grid.x - (unsigned)nsteps, grid.y);
        if (v4) GPX_PANEL_LAUNCH_LDS((panel_res_kernel<T, true, 4>), gdiag, 0, pad_lds);
        else if (v5 && mfma_chain) GPX_PANEL_LAUNCH((panel_res_kernel<T, true, 5>), gdiag, 0);
        else if (mfma_chain) GPX_PANEL_LAUNCH((panel_res_kernel<T, true>), gdiag, 0);
        else GPX_PANEL_LAUNCH((panel_res_kernel<T, F64>), gdiag, 0);
        GPX_PANEL_LAUNCH((panel_res_kernel<T, F64>), grows, nsteps);          // (the rows never run a leaf: the lean instantiation)
    } else if (v4) {
        GPX_PANEL_LAUNCH_LDS((panel_res_kernel<T, true, 4>), grid, 0, pad_lds);
    } else if (v5 && mfma_single) {
        GPX_PANEL_LAUNCH((panel_res_kernel<T, true, 5>), grid, 0);
    } else if (mfma_single) {
        GPX_PANEL_LAUNCH((panel_res_kernel<T, true>), grid, 0);
    } else {
        GPX_PANEL_LAUNCH((panel_res_kernel<T, F64>), grid, 0);
    }
#undef GPX_PANEL_LAUNCH
#undef GPX_PANEL_LAUNCH_LDS
    GPX_LAUNCH_CHECK();
    prof.reset();
    if (record_after) GPX_HIP(hipEventRecord(record_after, st));
    scr->last = st; scr->have_last = true; scr->last_on_side = (side != nullptr && st == side);
    if (!scr->last_on_side) GPX_HIP(hipEventRecord(scr->ev, st));               // (a caller's stream may not outlive this call)
    return GPX_OK;
}

// kpre (a multiple of 64, <= c0): that many columns immediately to the left of the panel, rows [r0, n), are applied to
// it first (P -= R R_d^T); the caller then omits that update
int potrf_panel_res(int dtype, void *A, int64_t lda, int64_t n, int64_t r0, int64_t c0, int64_t kb, int *info_dev,
                    hipStream_t st, const Batch *bt, int64_t kpre, hipEvent_t done)
{
    if (dtype == GPX_F64) return panel_res_t<double>((double *)A, lda, n, r0, c0, kb, info_dev, st, bt, kpre, done);
    return panel_res_t<float>((float *)A, lda, n, r0, c0, kb, info_dev, st, bt, kpre, done);
}

// 1 passed, 2 failed, 0 could not run (see leaf_asm_ok).  Called under g_leaf_mu.
static int leaf_selfcheck_run(int dev, hipStream_t st)
{
    (void)dev;
    constexpr int64_t N = RES_MAXSTEPS * IB, LD = N, GM = 4096, GK = 512;
    // not while `st` is being captured (synchronising calls are illegal then).  A capture open on ANOTHER stream of this
    // thread in global or thread-local mode would be invalidated by the check's hipMalloc / hipMemcpy: the thread's capture
    // mode is relaxed for the duration of the check (its launches go to streams of its own, never into a capture)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (cs != hipStreamCaptureStatusNone) return 0;
    struct CaptureModeGuard {
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed; bool ok = false;
        CaptureModeGuard() { ok = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess; if (!ok) (void)hipGetLastError(); }
        ~CaptureModeGuard() { if (ok) (void)hipThreadExchangeStreamCaptureMode(&mode); }
    } capture_mode__;
    // a full-mantissa SPD panel: A = B B^T / 8 + 2 I, B (N x 32) from a fixed linear congruential stream
    constexpr int64_t KB = 32;
    std::vector<double> hA((size_t)N * LD), hB((size_t)N * KB);
    unsigned long long z = 0x9E3779B97F4A7C15ull;
    for (auto &v : hB) { z = z * 6364136223846793005ull + 1442695040888963407ull; v = (double)(int64_t)(z >> 11) / 9007199254740992.0 - 0.5; }
    for (int64_t i = 0; i < N; ++i)
        for (int64_t j = 0; j <= i; ++j) {
            double acc = 0;
            for (int64_t k = 0; k < KB; ++k) acc += hB[i * KB + k] * hB[j * KB + k];
            hA[i * LD + j] = hA[j * LD + i] = acc / 8.0 + (i == j ? 2.0 : 0.0);
        }
    std::vector<float> hAf(hA.begin(), hA.end());
    void *dA = nullptr, *dG = nullptr, *dC = nullptr; int *dinfo = nullptr;
    hipStream_t s1 = nullptr, s2 = nullptr;
    int verdict = 0;
    auto cleanup = [&]() {
        stream_epoch_bump();
        if (s1) { (void)hipStreamSynchronize(s1); (void)hipStreamDestroy(s1); }
        if (s2) { (void)hipStreamSynchronize(s2); (void)hipStreamDestroy(s2); }
        (void)hipFree(dA); (void)hipFree(dG); (void)hipFree(dC); (void)hipFree(dinfo);
    };
#define GPX_SC(call) do { if ((call) != hipSuccess) { (void)hipGetLastError(); cleanup(); return 0; } } while (0)
    GPX_SC(hipMalloc(&dA, (size_t)N * LD * 8));
    GPX_SC(hipMalloc(&dG, (size_t)GM * GK * 8));
    GPX_SC(hipMalloc(&dC, (size_t)GM * GM * 8));
    GPX_SC(hipMalloc((void **)&dinfo, sizeof(int)));
    GPX_SC(hipMemset(dG, 0, (size_t)GM * GK * 8));
    GPX_SC(hipMemset(dC, 0, (size_t)GM * GM * 8));
    GPX_SC(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    GPX_SC(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto factor = [&](int dtype, int level, bool contended, std::vector<double> &L) -> bool {
        const size_t es = esize(dtype);
        if (hipMemcpy(dA, dtype == GPX_F64 ? (const void *)hA.data() : (const void *)hAf.data(), (size_t)N * LD * es, hipMemcpyHostToDevice) != hipSuccess) return false;
        if (hipMemset(dinfo, 0, sizeof(int)) != hipSuccess) return false;
        if (contended)                                           // ~1 ms of products on every matrix pipe of the chip
            for (int r = 0; r < 3; ++r)
                if (gemm_nt(GPX_F64, GM, GM, GK, dG, GK, dG, GK, dC, GM, 1.0, GPX_FULL, 0, 0, s2) != GPX_OK) return false;
        g_leaf_force = level;
        const int rc = potrf_panel_res(dtype, dA, LD, N, 0, 0, N, dinfo, s1, nullptr, 0, nullptr);
        g_leaf_force = 0;
        if (rc != GPX_OK) return false;
        if (hipStreamSynchronize(s1) != hipSuccess || hipStreamSynchronize(s2) != hipSuccess) return false;
        int info = -1;
        if (hipMemcpy(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess || info != 0) return false;
        L.assign((size_t)N * N, 0.0);
        if (dtype == GPX_F64) {
            std::vector<double> t((size_t)N * LD);
            if (hipMemcpy(t.data(), dA, t.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return false;
            for (int64_t i = 0; i < N; ++i) for (int64_t j = 0; j <= i; ++j) L[i * N + j] = t[i * LD + j];
        } else {
            std::vector<float> t((size_t)N * LD);
            if (hipMemcpy(t.data(), dA, t.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
            for (int64_t i = 0; i < N; ++i) for (int64_t j = 0; j <= i; ++j) L[i * N + j] = t[i * LD + j];
        }
        return true;
    };
    bool ran = true, good = true;
    for (int dtype : {GPX_F64, GPX_F32}) {
        std::vector<double> ref, got;
        // (fp32 storage: the asm leaf works in fp64 on converted blocks and rounds once; the fp32 MFMA leaf it is compared
        //  with accumulates in fp32 -- the bound is fp32 round-off there, which a single-precision slip in an fp64 tile does not
        //  exceed either: the fp64 run is the discriminating one, the fp32 run checks the instantiation's plumbing)
        const double tol = dtype == GPX_F64 ? 1e-12 : 2e-5;
        if (!factor(dtype, 1, false, ref)) { ran = false; break; }
        for (int level : {4, 5})
            for (int pass = 0; pass < 2 && ran; ++pass) {
                if (!factor(dtype, level, pass == 1, got)) { ran = false; break; }
                double err = 0, scale = 0;
                for (size_t i = 0; i < ref.size(); ++i) { err = std::max(err, fabs(got[i] - ref[i])); scale = std::max(scale, fabs(ref[i])); }
                if (!(err <= tol * scale)) good = false;
            }
        if (!ran) break;
    }
#undef GPX_SC
    verdict = !ran ? 0 : (good ? 1 : 2);
    cleanup();
    return verdict;
}

// fold the update by the columns to the left into the panel kernel?  (short panels only: a tall panel's workgroups
// run in several rounds and the tuned GEMM does the same update faster than they do)
bool panel_res_fold(int64_t rows, int64_t kpre, int64_t kb, size_t es, int64_t lda, const void *base)
{
    const int64_t rows_max = tune().fold_rows;
    const int64_t kpre_max = tune().fold_k;
    return kb % IB == 0 && kb <= panel_res_max() && kpre % IB == 0 && kpre > 0 && kpre <= kpre_max && rows <= rows_max &&
           lda % (16 / (int64_t)es) == 0 && ((uintptr_t)base) % 16 == 0;
}

}  // namespace gpx

// the state of the asm leaf's self-check on the current device: 0 not run yet, 1 passed, 2 failed, 3 could not be run in three
// attempts (2 and 3: the library uses the compiler-scheduled leaf); run_now != 0 runs it if it has not run
extern "C" int gpx_debug_leaf_selfcheck(int run_now, int *state)
{
    GPX_TRY(gpx::ensure_device());
    if (!state) return GPX_ERR_ARG;
    if (run_now) (void)gpx::leaf_asm_ok();
    int dev = 0;
    GPX_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(gpx::g_leaf_mu);
    *state = (dev >= 0 && dev < gpx::RES_MAXDEV) ? gpx::g_leaf_state[dev] : 0;
    return GPX_OK;
}

// diagnostic: the `at`-th resident panel launch from now stores 8 words per workgroup into dev_buffer (tools/panel_stamps.py)
extern "C" int gpx_debug_panel_stamps(void *dev_buffer, int at)
{
    gpx::g_res_stamps = (unsigned long long *)dev_buffer;
    gpx::g_res_stamp_at = at;
    return GPX_OK;
}
