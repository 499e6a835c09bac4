// poa_dp.hip.h -- gfx950 device code: workgroup execution context, the sequence-to-DAG DP
// fill and the traceback.  Replaces spoa::AlignmentEngine::Align (call site
// src/smooth.cpp:761); semantics S1-S5 of DESIGN.md / oracle/poa_oracle.c.
//
// Mapping (one workgroup = one alignment, T = 64*NW lanes, wave64):
//   * the matrix is swept one graph row at a time; lane t owns the W consecutive columns
//     [t*W, (t+1)*W) and keeps the previous row's H/F/O for them in VGPRs;
//   * a row whose only predecessor is the previous rank never touches memory for its
//     inputs; other predecessor rows come from a block-private ring of packed rows
//     (H int16|int32 + clamped H-F, H-O deltas) that only rows with a far successor write;
//   * the in-row gap states E/Q are a max-plus prefix problem: pass 1 computes each lane's
//     strip-local carry, a wave64 shuffle scan + a 2-barrier LDS hop across the waves
//     distributes the carries, pass 2 replays the strip with the true carry and records the
//     traceback byte (and the winning-predecessor ordinals on multi-pred rows).
#pragma once
#include <hip/hip_runtime.h>
#include "poa_types.h"

namespace sxg {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));

// ---------------------------------------------------------------------------------------
// workgroup context for poa_graph_dev.h
// (the scratch words are held as an LDS-address-space pointer built from the LDS offset: a generic pointer to them became a
// constant "addrspacecast(smem) + 512" argument of the out-of-line graph phases after inter-procedural constant propagation,
// whose null check -- 0 against the shared aperture -- the AMDGPU back end emits as an illegal VOPC encoding for some kernels)
typedef __attribute__((address_space(3))) int sxg_lds_int;
// GBv: elements per thread and step of the graph phases' loops (poa_graph_dev.h): 4 for workgroups of four waves and more,
// 8 / 16 for two- / one-wave workgroups (the kernels pick the context by their run-time thread count)
template <int GBv>
struct WgCtxT {
    static constexpr int GB = GBv, GBH = GBv > 8 ? 8 : GBv, SCAN_K = GBv;
    sxg_lds_int* lds;  // >= 2*16+2 ints of LDS scratch
    __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
    __device__ __forceinline__ int nthreads() const { return (int)blockDim.x; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ int atomic_add(int32_t* p, int v) const { return atomicAdd(p, v); }
    __device__ __forceinline__ int atomic_min(int32_t* p, int v) const { return atomicMin(p, v); }
    __device__ int scan_excl_add(int v, int* total) const {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
        int x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(x, d);
            if (lane >= d) x += o;
        }
        if (lane == 63) lds[w] = x;
        __syncthreads();
        int base = 0, tot = 0;
        for (int i = 0; i < nw; ++i) {
            const int s = lds[i];
            if (i < w) base += s;
            tot += s;
        }
        __syncthreads();
        *total = tot;
        return base + x - v;
    }
    __device__ int scan_excl_max(int v, int* total) const {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
        const int NONE = -0x7fffffff;
        int x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(x, d);
            if (lane >= d) x = max(x, o);
        }
        int before = __shfl_up(x, 1);
        if (lane == 0) before = NONE;
        if (lane == 63) lds[w] = x;
        __syncthreads();
        int tot = NONE;
        for (int i = 0; i < nw; ++i) {
            const int s = lds[i];
            if (i < w) before = max(before, s);
            tot = max(tot, s);
        }
        __syncthreads();
        *total = tot;
        return before;
    }
    __device__ int reduce_max(int v) const {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
        int x = v;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x = max(x, __shfl_xor(x, d));
        if (lane == 0) lds[w] = x;
        __syncthreads();
        int r = lds[0];
        for (int i = 1; i < nw; ++i) r = max(r, lds[i]);
        __syncthreads();
        return r;
    }
};
typedef WgCtxT<4> WgCtx;

// ---------------------------------------------------------------------------------------
// packed row words of the row pool
// native vector types: usable behind address-space-qualified pointers (HIP's uint2/int4 classes are not)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <bool H16> struct RowWord;
template <> struct RowWord<true> {
    using type = uint32_t;
    static __device__ __forceinline__ type pack(int h, int f, int o) {
        const unsigned df = (unsigned)min(h - f, 255), dq = (unsigned)min(h - o, 255);
        return ((unsigned)h & 0xffffu) | (df << 16) | (dq << 24);
    }
    static __device__ __forceinline__ void unpack(type w, int& h, int& f, int& o) {
        h = (int)(short)(w & 0xffffu);
        f = h - (int)((w >> 16) & 0xffu);
        o = h - (int)(w >> 24);
    }
    static __device__ __forceinline__ int h_of(type w) { return (int)(short)(w & 0xffffu); }
};
template <> struct RowWord<false> {
    using type = u32x2;
    static __device__ __forceinline__ type pack(int h, int f, int o) {
        const unsigned df = (unsigned)min(h - f, 65535), dq = (unsigned)min(h - o, 65535);
        return u32x2{(unsigned)h, df | (dq << 16)};
    }
    static __device__ __forceinline__ void unpack(type w, int& h, int& f, int& o) {
        h = (int)w.x;
        f = h - (int)(w.y & 0xffffu);
        o = h - (int)(w.y >> 16);
    }
    static __device__ __forceinline__ int h_of(type w) { return (int)w.x; }
};

// Pointers that reach the sweeps through structs in private memory lose their address space and
// the compiler falls back to FLAT loads/stores: 64-bit VGPR addresses, and -- worse -- every flat
// access also counts on lgkmcnt, so each "s_waitcnt lgkmcnt(0)" in front of a barrier or behind a
// ds_bpermute waited for all outstanding HBM traffic.  The hot paths therefore re-type their
// base pointers as global once (scalar base + lane offset, vmcnt only).
#define SXG_GLOBAL __attribute__((address_space(1)))
template <class Tp> __device__ __forceinline__ SXG_GLOBAL Tp* sxg_global(Tp* p) { return (SXG_GLOBAL Tp*)p; }
template <class Tp> __device__ __forceinline__ SXG_GLOBAL const Tp* sxg_global(const Tp* p) { return (SXG_GLOBAL const Tp*)p; }
// ... and pin them to scalar registers: read back from a struct they are "divergent" to the compiler,
// which then keeps them in VGPR pairs, does 64-bit VALU address arithmetic per access and -- under
// register pressure -- spills them to scratch, whose reload is an s_waitcnt vmcnt(0) that also waits
// for every store still in flight (measured: 11 % of the row time in front of the mask-plane stores).
template <class Tp> __device__ __forceinline__ SXG_GLOBAL Tp* sxg_uniform(SXG_GLOBAL Tp* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (SXG_GLOBAL Tp*)(((unsigned long long)hi << 32) | lo);
}

// Wave-wide inclusive max-scan and one-lane shift on the DPP data path (gfx9 row_shr / row_bcast /
// wave_shr): 6 + 1 full-rate VALU instructions per value instead of 6 ds_bpermute round trips
// through the LDS crossbar plus their compare/select.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int sxg_dpp_max(const int v) {
    return max(v, __builtin_amdgcn_update_dpp((int)0x80000000, v, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ int sxg_wave_incl_max(int v) {
    v = sxg_dpp_max<0x111, 0xf>(v);  // row_shr:1
    v = sxg_dpp_max<0x112, 0xf>(v);  // row_shr:2
    v = sxg_dpp_max<0x114, 0xf>(v);  // row_shr:4
    v = sxg_dpp_max<0x118, 0xf>(v);  // row_shr:8
    v = sxg_dpp_max<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = sxg_dpp_max<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}
// lane l receives v of lane l-1; lane 0 receives `first`
__device__ __forceinline__ int sxg_wave_shr1(const int v, const int first) {
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false);  // wave_shr:1
}

struct DpBuffers {
    uint8_t* tb;       // [(rows_cap+1) * Lpad] traceback bytes, row-major, row 0 unused
    uint32_t* steps;   // [step_cap * 3 * T] fold-step masks of multi-pred rows: for step s (=
                       //   "predecessor #x folded in") and lane t, words {D,F,O}[t] hold one bit per
                       //   column of the lane's strip: 1 = predecessor #x took over that state
    void* pool;        // [pool_slots * Lpad] packed rows
    void* row0;        // [Lpad] packed virtual source row
    void* park;        // [Lpad] parked previous row when it cannot live in LDS
    int band_strips;   // packed sweep: strips per row kept in the traceback plane (see poa_dp16.hip.h)
    int lds_rows;      // packed sweep: on-chip copies of stored rows the workgroup's LDS holds (RowCaps::lds_rows)
    int band_w;        // banded sweep (poa_band16.hip.h): half-width of the band in columns, wb + (int)(wf * L)
    int band_mode;     // banded sweep: params.banded -- 1 = band around the backbone coordinate (B2), 2 = adaptive band (B4)
    int prio_rank;     // launch rank of this workgroup among its CU's co-residents (see sxg_rotate_prio)
    uint32_t* prio_board;          // this CU's progress board (PRIO_BOARD_SLOTS words) or nullptr
    unsigned long long prio_rem0;  // estimated cells this workgroup still has to do, at row 0 of this sweep
    unsigned long long* row_prof;  // SXG_ROW_PROF builds: 12 per-slot accumulators of row segments
};
constexpr int PRIO_BOARD_SLOTS = 16;   // co-resident workgroups per CU the board can tell apart
constexpr int PRIO_BOARD_CUS = 4096;   // __smid() & 0xfff

// Fair shares between co-resident workgroups.  The CU's instruction arbiter is oldest-wave-first,
// and workgroups k, k+#CU, k+2#CU, ... (launched in that order) share a CU: measured on the headline
// run the four co-residents finished at 0.64 / 0.74 / 0.86 / 0.98 of the kernel time although their
// blocks cost the same, so the CU ran its last quarter with one or two waves per SIMD.  User
// priority beats age, so every workgroup steps through priorities 3,2,1,0 in 41 us slices of the
// 100 MHz wall clock, offset by its launch rank: at any moment the co-residents hold four
// different priorities and over a rotation everybody gets the same share.
__device__ __forceinline__ void sxg_set_prio(const unsigned p) {
    if (p == 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}
// Closed loop on top of that idea (block kernel): every workgroup posts the work it has left on its
// CU's board and takes as priority the number of co-residents that have LESS left, so the one
// furthest from its end always runs first and the four converge on a common finish.
__device__ __forceinline__ void sxg_balance_prio(const DpBuffers& B, const unsigned long long done) {
    const unsigned long long left = B.prio_rem0 > done ? B.prio_rem0 - done : 0;
    const unsigned mine = (unsigned)min(left >> 16, 0xfffffffeull) + 1u;
    if (threadIdx.x == 0) __hip_atomic_store(B.prio_board + B.prio_rank, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the eight ranks come through the scalar data path (s_load, glc: past the constant cache): a
    // vector load here would be waited for with vmcnt, in order behind every store still in flight
    typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
    u32x16 bw;
    const unsigned long long bp = (unsigned long long)B.prio_board;
    const unsigned bhi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bp >> 32));  // (the builtin returns int:
    const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)bp);          //  no sign extension, please)
    const unsigned long long bps = ((unsigned long long)bhi << 32) | blo;
    asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(bw) : "s"(bps) : "memory");
    unsigned behind = 0;
#pragma unroll
    for (int k = 0; k < PRIO_BOARD_SLOTS; ++k)  // every rank of the board: the launches of one round share it (ranks offset per launch)
        behind += (k != B.prio_rank && bw[k] != 0u && bw[k] < mine) ? 1u : 0u;
    sxg_set_prio(min(__builtin_amdgcn_readfirstlane(behind), 3u));
}
__device__ __forceinline__ void sxg_rotate_prio(const int rank) {
    const unsigned ph = ((unsigned)((unsigned long long)wall_clock64() >> 12) + (unsigned)rank) & 3u;
    if (ph == 0) __builtin_amdgcn_s_setprio(3);
    else if (ph == 1) __builtin_amdgcn_s_setprio(2);
    else if (ph == 2) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

struct DpResult {
    int best, bi, bj;  // end cell (row index 1-based); bi < 0: empty alignment
};

// ---------------------------------------------------------------------------------------
// LDS carve-up of one workgroup (dynamic shared memory):
//   [0, 1 KiB)                 control words: scans, exchanges, work ticket
//   [1 KiB, 1 KiB + 8 KiB)     row descriptors (RowMeta) of the current 256-row chunk
//   [9 KiB, 9 KiB + Lpad*word) packed image of the previous row (multi-pred rows only)
constexpr int LDS_CTL_BYTES = 1024;
constexpr int META_CHUNK = 256;
constexpr int LDS_META_BYTES = META_CHUNK * 32;
__host__ __device__ constexpr int dp_lds_bytes(int Lpad, int word_bytes) {
    return LDS_CTL_BYTES + LDS_META_BYTES + Lpad * word_bytes;
}
// LDS to request at launch: the parked row only if it leaves room for >= 2 workgroups per CU
// (or is small); otherwise it goes to the slot's HBM scratch.
__host__ __device__ constexpr bool dp_park_in_lds(int Lpad, int word_bytes) {
    return Lpad * word_bytes <= 64 * 1024;
}
// packed sweep: descriptors + parked row (8 B per lane and column = 4 B per padded column) + the
// query letters of every lane (one word per two columns, rounded up for odd strip widths)
__host__ __device__ constexpr int dp16_let_words(int W) { return (W + 1) / 2; }
// One- and two-wave workgroups (sequences up to ~3 kbp, the usual smoothxg block) are LDS-bound in
// occupancy: they stage 128 row descriptors instead of 256 (and walk a 32-row traceback window),
// 4 KB instead of 8 KB, which lets 14 instead of 10 of them share a CU.
// (round 3: the packed sweep reads its row descriptors through the scalar cache; what is left of the staging area holds the
// wave-to-wave mailboxes during a sweep and the traceback window after it: 4 KB for every geometry)
__host__ __device__ constexpr int dp16_meta_bytes(int) { return LDS_META_BYTES / 2; }
__host__ __device__ constexpr int dp16_row_bytes(int T, int W) { return T * W * 8 + T * 4; }   // a stored row + its left-neighbour words
// control words | mailboxes / traceback window | lds_rows on-chip copies of stored rows | query letters.  The traceback
// keeps its precomputed runs (2.3 KB) where the row copies start: never less than that behind the window.
__host__ __device__ constexpr int dp16_lds_bytes(int T, int W, int lds_rows) {
    const int body = lds_rows * dp16_row_bytes(T, W) + T * dp16_let_words(W) * 4;
    return LDS_CTL_BYTES + dp16_meta_bytes(T) + (body > 2560 ? body : 2560);
}
__host__ __device__ constexpr int dp_lds_launch_bytes(int Lpad, int word_bytes) {
    return dp_park_in_lds(Lpad, word_bytes) ? dp_lds_bytes(Lpad, word_bytes) : LDS_CTL_BYTES + LDS_META_BYTES;
}
// prefetch area: the packed words of one row + one 4-byte left value per lane; only when the
// whole carve-up stays <= 64 KiB (two or more workgroups per CU)
__host__ __device__ constexpr int dp_pf_bytes(int Lpad, int word_bytes, int threads) { return Lpad * word_bytes + threads * 4; }
__host__ __device__ constexpr int dp_pf_offset(int Lpad, int word_bytes, int threads) {
    return dp_lds_launch_bytes(Lpad, word_bytes) + dp_pf_bytes(Lpad, word_bytes, threads) <= 64 * 1024
               ? dp_lds_launch_bytes(Lpad, word_bytes) : -1;
}

// Register discipline: every per-column result is pinned with an empty asm right where it is
// produced and column pairs are fenced with sched_barrier.  Without this hipcc keeps the raw
// candidates of every column alive to derive the traceback bits later (measured: ~17 VGPRs per
// column instead of ~6) and the kernel drops to one wave per SIMD or spills.
#define SXG_ROW_BARRIER() __syncthreads()
#define SXG_PIN(...) asm volatile("" : __VA_ARGS__)
// columns are pinned in groups of SXG_G (a multiple of the unroll step that divides W): inside a
// group the scheduler may interleave the columns' dependent chains, across groups it may not
#ifndef SXG_G
#define SXG_G 1  // measured on MI355X: 1, 2 and 4 run within 1 %; 1 never spills
#endif
#if SXG_G == 1
#define SXG_COLS(a, k) "+v"(a[k])
#elif SXG_G == 2
#define SXG_COLS(a, k) "+v"(a[(k) - 1]), "+v"(a[k])
#else
#define SXG_COLS(a, k) "+v"(a[(k) - 3]), "+v"(a[(k) - 2]), "+v"(a[(k) - 1]), "+v"(a[k])
#endif
#define SXG_GROUP_END(k) (((k) % SXG_G) == SXG_G - 1)

// T (= blockDim.x, a multiple of 64) is a run-time value: one compiled kernel serves every
// strip count, the host picks T = 64 * ceil((L+1) / (64*W)).
template <int W, bool CVX, bool H16, bool SW>
__device__ __forceinline__ void dp_fill(const Scoring& S, const RowsView& R, const int N,
                        const uint8_t* __restrict__ seq, const int L, const DpBuffers& B,
                        char* smem, const bool park_in_lds, const int pf_off, DpResult& res) {
    static_assert(W % 4 == 0 && W <= 24, "W must be a multiple of 4, at most 24");
    const int T = (int)blockDim.x;
    const int NW = T >> 6;
    const int Lpad = T * W;
    constexpr unsigned ALL = (1u << W) - 1u;
    using RWt = RowWord<H16>;
    using Word = typename RWt::type;
    int* lds = (int*)smem;
    const int4* lmeta = (const int4*)(smem + LDS_CTL_BYTES);
    // the parked previous row (rows with >= 3 predecessors) lives in LDS unless it cannot fit
    Word* lrow = park_in_lds ? (Word*)(smem + LDS_CTL_BYTES + LDS_META_BYTES) : (Word*)B.park;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int j0 = t * W;
    // LDS-DMA prefetch area (pf_off < 0: disabled): the stored row the NEXT graph row will fold
    // first is fetched with global_load_lds while the current row computes, so its HBM/L2
    // latency hides behind ~10k cycles of work.  Piece c of lane t lives at c*T*16 + t*16.
    constexpr int PF_PIECES = W * (int)sizeof(Word) / 16;
    // addressed as raw LDS offsets (address space 3): dynamic LDS starts right after the static
    // LDS of the kernel, no generic-pointer casts involved
    const unsigned lds_pf = (unsigned)__builtin_amdgcn_groupstaticsize() + (unsigned)(pf_off >= 0 ? pf_off : 0);
    const unsigned lds_pfl = lds_pf + (unsigned)PF_PIECES * (unsigned)T * 16u;  // [T] 4-byte H left of each strip
    int pf_row = -1;                                                            // row whose words are (being) fetched
    // global, scalar base pointers (see sxg_global / sxg_uniform)
    SXG_GLOBAL Word* const g_row0 = sxg_uniform(sxg_global((Word*)B.row0));
    SXG_GLOBAL Word* const g_pool = sxg_uniform(sxg_global((Word*)B.pool));
    SXG_GLOBAL uint8_t* const g_tb = sxg_uniform(sxg_global(B.tb));
    SXG_GLOBAL uint32_t* const g_steps = sxg_uniform(sxg_global(B.steps));
    SXG_GLOBAL const int32_t* const g_meta = sxg_uniform(sxg_global((const int32_t*)R.meta));
    SXG_GLOBAL const int32_t* const g_preds = sxg_uniform(sxg_global((const int32_t*)R.preds));
    SXG_GLOBAL const int32_t* const g_slot = sxg_uniform(sxg_global((const int32_t*)R.slot));
    const unsigned uj0 = (unsigned)(t * W);
    const int g = S.g, e = S.e, q = S.q, c = S.c, mm = S.m, mn = S.n;
    const int We = W * e, Wc = W * c;
    int* tot_a = lds;        // [16]
    int* tot_b = lds + 16;   // [16]
    int* xch_h = lds + 32;   // [16]
    int* xch_b = lds + 48;   // [16]

    // query letters of my columns, 4 per VGPR: column j pairs with seq[j-1]; 255 = no letter
    unsigned qcw[W / 4];
#pragma unroll
    for (int k4 = 0; k4 < W / 4; ++k4) {
        unsigned v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = j0 + k4 * 4 + b;
            const unsigned ch = (j >= 1 && j <= L) ? (unsigned)seq[j - 1] : 255u;
            v |= (ch > 4u && ch != 255u ? 4u : ch) << (8 * b);
        }
        qcw[k4] = v;
    }

    int Hp[W], Fp[W], Op[W], Hleft;
    // virtual row 0
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const int j = j0 + k;
        int h = 0;
        if (!SW && j > 0) { const int a = g + (j - 1) * e, b = q + (j - 1) * c; h = a > b ? a : b; }
        Hp[k] = h; Fp[k] = NEG; Op[k] = NEG;
    }
    {
        const int j = j0 - 1;
        int h = 0;
        if (!SW && j > 0) { const int a = g + (j - 1) * e, b = q + (j - 1) * c; h = a > b ? a : b; }
        Hleft = j < 0 ? NEG : h;
    }
    {
#pragma unroll
        for (int k = 0; k < W; ++k) (g_row0 + k)[uj0] = RWt::pack(Hp[k], Fp[k], Op[k]);
    }
    // end-cell tracking.  Local: key = H<<5 | (31-k) so that one max per column finds the
    // greatest H and, among equals, the smallest column; rows are compared on H only (strictly).
    int best = SW ? 0 : NEG * 2, bi = -1, bj = -1;  // bj: column index INSIDE my strip
    const int kL = L - j0;                          // strip-local index of the end column L

    for (int i = 1; i <= N; ++i) {
        if (B.prio_board) { if ((i & 127) == 1) sxg_balance_prio(B, (unsigned long long)i * (unsigned long long)L); }
        else if ((i & 63) == 1) sxg_rotate_prio(B.prio_rank);
        const int r = i - 1;
        if ((r & (META_CHUNK - 1)) == 0) {
            // stage the descriptors of the next 256 rows (all waves are past row r-1 here)
            __syncthreads();
            SXG_GLOBAL const i32x4* gm = (SXG_GLOBAL const i32x4*)(g_meta + 8 * (size_t)r);
            i32x4* lm = (i32x4*)(smem + LDS_CTL_BYTES);
            const int nrow = min(META_CHUNK, N - r);
            for (int x = t; x < 2 * nrow; x += T) lm[x] = gm[x];
            __syncthreads();
        }
        const int4 m0 = lmeta[2 * (r & (META_CHUNK - 1))], m1 = lmeta[2 * (r & (META_CHUNK - 1)) + 1];
        const int pb = __builtin_amdgcn_readfirstlane(m0.x);
        const int info = __builtin_amdgcn_readfirstlane(m0.y);
        const int p0 = __builtin_amdgcn_readfirstlane(m0.z);
        const int s0 = __builtin_amdgcn_readfirstlane(m0.w);
        const int p1 = __builtin_amdgcn_readfirstlane(m1.x);
        const int s1 = __builtin_amdgcn_readfirstlane(m1.y);
        const int myslot = __builtin_amdgcn_readfirstlane(m1.z);
        const int tx = __builtin_amdgcn_readfirstlane(m1.w);
        const int np = info & 0xffff, code = (info >> 16) & 0xff, flags = (info >> 24) & 0xff;
#pragma unroll
        for (int k4 = 0; k4 < W / 4; ++k4) SXG_PIN("+v"(qcw[k4]));  // stay packed: no per-column hoisting

        int Hc[W];                  // first the best diagonal source, later the final H of this row
        unsigned fxm = 0, oxm = 0;  // "came from EXTEND" bit of F / O, one bit per column

// my words of the prefetched row (each lane reads back only what its own DMA wrote, so the
// wave's vmcnt is the only ordering needed)
#define SXG_READ_PF(wr, hl)                                                                    \
    do {                                                                                       \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
        _Pragma("unroll") for (int c_ = 0; c_ < PF_PIECES; ++c_) {                             \
            const u32x4 v_ = *(const __attribute__((address_space(3))) u32x4*)(size_t)(lds_pf + (unsigned)c_ * (unsigned)T * 16u + (unsigned)t * 16u); \
            if (H16) {                                                                         \
                *(unsigned*)&wr[4 * c_] = v_.x; *(unsigned*)&wr[4 * c_ + 1] = v_.y;            \
                *(unsigned*)&wr[4 * c_ + 2] = v_.z; *(unsigned*)&wr[4 * c_ + 3] = v_.w;        \
            } else {                                                                           \
                *(uint2*)&wr[2 * c_] = make_uint2(v_.x, v_.y);                                 \
                *(uint2*)&wr[2 * c_ + 1] = make_uint2(v_.z, v_.w);                             \
            }                                                                                  \
        }                                                                                      \
        const unsigned l_ = *(const __attribute__((address_space(3))) unsigned*)(size_t)(lds_pfl + (unsigned)t * 4u); \
        hl = j0 > 0 ? (H16 ? (int)(short)(l_ & 0xffffu) : (int)l_) : NEG;                      \
    } while (0)
// first source of a row: F/O/diag straight from (hs, fs, os, hprev)
#define SXG_INIT(k, hs, fs, os, hprev)                                              \
    do {                                                                            \
        const int c1_ = (hs) + g, c2_ = (fs) + e;                                   \
        const bool x1_ = c2_ > c1_;                                                 \
        const int d1_ = (hs) + q, d2_ = (os) + c;                                   \
        const bool x2_ = CVX && d2_ > d1_;                                          \
        Hc[k] = (hprev);                                                            \
        Fp[k] = x1_ ? c2_ : c1_;                                                    \
        fxm |= (unsigned)x1_ << (k);                                                \
        if (CVX) { Op[k] = x2_ ? d2_ : d1_; oxm |= (unsigned)x2_ << (k); }          \
    } while (0)
#define SXG_PIN_INIT(k) \
    if (SXG_GROUP_END(k)) SXG_PIN(SXG_COLS(Hc, k), SXG_COLS(Fp, k), SXG_COLS(Op, k), "+v"(fxm), "+v"(oxm))

        if (np <= 1 && p0 == i - 1) {
            // ---- the single predecessor is the row in registers: update F and O in place
#pragma unroll
            for (int k = 0; k < W; ++k) {
                SXG_INIT(k, Hp[k], Fp[k], Op[k], (k ? Hp[k - 1] : Hleft));
                SXG_PIN_INIT(k);
            }
        } else {
            // ---- general row.  Sources are folded one at a time: the first one initialises the
            // accumulators, each further one takes a state over where it is better.  List-order
            // tie-breaks (S3): a later predecessor must be STRICTLY better.  With two
            // predecessors of which #1 is the register row, #1 is folded first (in place, no
            // parking) and #0 second with "better or equal" (ge = 1).  Three or more: the
            // register row is parked in LDS (own columns only, so no barrier) and everything is
            // folded in list order.
            const bool reg0 = (p0 == i - 1), reg1 = (np == 2 && p1 == i - 1);
            const bool park = np >= 3;
            if (park) {
#pragma unroll
                for (int k = 0; k < W; ++k) lrow[j0 + k] = RWt::pack(Hp[k], Fp[k], Op[k]);
            }
            const int ge = reg1 ? 1 : 0;
            // ---- first source
            if ((reg0 || reg1) && !park) {
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    SXG_INIT(k, Hp[k], Fp[k], Op[k], (k ? Hp[k - 1] : Hleft));
                    SXG_PIN_INIT(k);
                }
            } else {
                Word wr[W];
                int hl = Hleft;
                if (reg0) {  // parked copy of the register row
#pragma unroll
                    for (int k = 0; k < W; ++k) wr[k] = lrow[j0 + k];
                } else if (p0 == pf_row) {
                    SXG_READ_PF(wr, hl);
                } else {
                    SXG_GLOBAL const Word* sp = (p0 == 0) ? g_row0 : g_pool + (size_t)s0 * Lpad;
#pragma unroll
                    for (int k = 0; k < W; ++k) wr[k] = (sp + k)[uj0];
                    hl = j0 > 0 ? RWt::h_of((sp - 1)[uj0]) : NEG;
                }
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    int hs, fs, os;
                    RWt::unpack(wr[k], hs, fs, os);
                    SXG_INIT(k, hs, fs, os, hl);
                    hl = hs;
                    SXG_PIN_INIT(k);
                }
            }
            // ---- further sources, one fold step each
            for (int x = 1; x < np; ++x) {
                int p, sl;
                if (x == 1) { p = reg1 ? p0 : p1; sl = reg1 ? s0 : s1; }
                else {
                    p = __builtin_amdgcn_readfirstlane(g_preds[pb + x]);
                    sl = (p >= 1 && p != i - 1) ? __builtin_amdgcn_readfirstlane(g_slot[p - 1]) : -1;
                }
                Word wr[W];
                int hl = Hleft;
                if (p == i - 1) {  // only with three or more predecessors: the parked copy
#pragma unroll
                    for (int k = 0; k < W; ++k) wr[k] = lrow[j0 + k];
                } else if (p == pf_row) {
                    SXG_READ_PF(wr, hl);
                } else {
                    SXG_GLOBAL const Word* sp = (p == 0) ? g_row0 : g_pool + (size_t)sl * Lpad;
#pragma unroll
                    for (int k = 0; k < W; ++k) wr[k] = (sp + k)[uj0];
                    hl = j0 > 0 ? RWt::h_of((sp - 1)[uj0]) : NEG;
                }
                // fold: "cand + ge > cur" is (cand > cur) for ge = 0 and (cand >= cur) for ge = 1;
                // the masks start at 0 (ge = 0: set on take-over) or ALL (ge = 1: cleared on
                // take-over), both expressed as one xor
                unsigned dm = ge ? ALL : 0u, fmk = dm, omk = CVX ? dm : 0u;
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    int hs, fs, os;
                    RWt::unpack(wr[k], hs, fs, os);
                    const int c1 = hs + g, c2 = fs + e;
                    const bool x1 = c2 > c1;
                    const int cb = x1 ? c2 : c1;
                    const bool rf = cb + ge > Fp[k];
                    Fp[k] = rf ? cb : Fp[k];
                    fxm = (fxm & ~((unsigned)rf << k)) | ((unsigned)(rf && x1) << k);
                    fmk ^= (unsigned)rf << k;
                    if (CVX) {
                        const int d1 = hs + q, d2 = os + c;
                        const bool x2 = d2 > d1;
                        const int db = x2 ? d2 : d1;
                        const bool ro = db + ge > Op[k];
                        Op[k] = ro ? db : Op[k];
                        oxm = (oxm & ~((unsigned)ro << k)) | ((unsigned)(ro && x2) << k);
                        omk ^= (unsigned)ro << k;
                    }
                    const bool rd = hl + ge > Hc[k];
                    Hc[k] = rd ? hl : Hc[k];
                    dm ^= (unsigned)rd << k;
                    hl = hs;
                    if (SXG_GROUP_END(k))
                        SXG_PIN(SXG_COLS(Hc, k), SXG_COLS(Fp, k), SXG_COLS(Op, k), "+v"(fxm), "+v"(oxm), "+v"(dm), "+v"(fmk), "+v"(omk), "+v"(hl));
                }
                SXG_GLOBAL uint32_t* st = g_steps + ((size_t)(tx + x - 1) * 3) * T;
                st[(unsigned)t] = dm; (st + T)[(unsigned)t] = fmk; (st + 2 * T)[(unsigned)t] = omk;
            }
        }
#undef SXG_INIT
#undef SXG_PIN_INIT
#undef SXG_READ_PF
        if (!CVX) {
#pragma unroll
            for (int k = 0; k < W; ++k) Op[k] = NEG;
        }

        // ---- start fetching what the NEXT row folds first (see pf_area above)
        pf_row = -1;
        if (pf_off >= 0 && i < N && (i & (META_CHUNK - 1)) != 0) {
            const int4 n0 = lmeta[2 * (i & (META_CHUNK - 1))], n1 = lmeta[2 * (i & (META_CHUNK - 1)) + 1];
            const int np1 = __builtin_amdgcn_readfirstlane(n0.y) & 0xffff;
            const int q0 = __builtin_amdgcn_readfirstlane(n0.z), t0 = __builtin_amdgcn_readfirstlane(n0.w);
            const int q1 = __builtin_amdgcn_readfirstlane(n1.x), t1 = __builtin_amdgcn_readfirstlane(n1.y);
            const bool r0n = (q0 == i), r1n = (np1 == 2 && q1 == i);
            int tp = -1, ts = -1;
            if (np1 >= 3 || !(r0n || r1n)) { tp = q0; ts = t0; }               // first source lives in memory
            else if (np1 == 2) { tp = r1n ? q0 : q1; ts = r1n ? t0 : t1; }     // register row first, then this one
            if (tp == i) tp = -1;                                              // (parked copy of the register row)
            if (tp >= 0) {
                const char* src = (const char*)((tp == 0) ? (const Word*)B.row0 : (const Word*)B.pool + (size_t)ts * Lpad);  // (opt-in path)
                const char* mine = src + (size_t)j0 * sizeof(Word);
                // the LDS side of the DMA is a wave-uniform base (M0) + lane * size
                const unsigned wave0 = (unsigned)__builtin_amdgcn_readfirstlane(wv) * 64u;
#pragma unroll
                for (int c_ = 0; c_ < PF_PIECES; ++c_)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(mine + c_ * 16),
                                                     (void __attribute__((address_space(3)))*)(size_t)(lds_pf + (unsigned)c_ * (unsigned)T * 16u + wave0 * 16u),
                                                     16, 0, 0);
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(j0 > 0 ? mine - sizeof(Word) : mine),
                                                 (void __attribute__((address_space(3)))*)(size_t)(lds_pfl + wave0 * 4u), 4, 0, 0);
                pf_row = tp;
            }
        }

        // ---- H before the in-row gaps, strip-local carries (pass 1)
        unsigned fm = 0, om = 0;  // H came from F / from O
        int a = NEG, b = NEG;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const unsigned qb = (qcw[k >> 2] >> (8 * (k & 3))) & 255u;
            int h = Hc[k] + ((int)qb == code ? mm : mn);
            if (Fp[k] > h) { h = Fp[k]; fm |= 1u << k; }
            if (CVX && Op[k] > h) { h = Op[k]; om |= 1u << k; }
            Hc[k] = h;
            const int hc = SW ? max(h, 0) : h;
            a = max(a + e, hc + g);
            if (CVX) b = max(b + c, hc + q);
            if (SXG_GROUP_END(k)) SXG_PIN(SXG_COLS(Hc, k), "+v"(fm), "+v"(om), "+v"(a), "+v"(b));
        }
        fm &= ~om;
        // carries: Ein(t) = max_{s<t} (a_s + (t-1-s)*W*e)
        int ya = a - t * We, yb = CVX ? b - t * Wc : NEG;
        ya = sxg_wave_incl_max(ya);
        if (CVX) yb = sxg_wave_incl_max(yb);
        if (NW > 1) {
            if (lane == 63) { tot_a[wv] = ya; tot_b[wv] = yb; }
            SXG_ROW_BARRIER();  // B1
            int ba = NEG * 2, bb = NEG * 2;
            for (int x = 0; x < wv; ++x) { ba = max(ba, tot_a[x]); bb = max(bb, tot_b[x]); }
            ya = max(ya, ba); yb = max(yb, bb);
            ya = sxg_wave_shr1(ya, ba); yb = sxg_wave_shr1(yb, bb);
        } else {
            ya = sxg_wave_shr1(ya, NEG * 2); yb = sxg_wave_shr1(yb, NEG * 2);
        }
        int E = (t == 0) ? NEG : ya + (t - 1) * We;
        int Q = (t == 0 || !CVX) ? NEG : yb + (t - 1) * Wc;

        // ---- pass 2: final H, traceback bytes
        unsigned tbw[W / 4];
        unsigned ebit = 0, qbit = 0;  // ext flags of the CURRENT column (column 0 patched later)
        int rowkey = 0;               // SW: max over my columns of H<<5 | (31-k)
#pragma unroll
        for (int k = 0; k < W; ++k) {
            int h = Hc[k];
            unsigned src = SRC_D + ((fm >> k) & 1u) + 2u * ((om >> k) & 1u);
            if (E > h) { h = E; src = SRC_E; }
            if (CVX && Q > h) { h = Q; src = SRC_Q; }
            if (SW && h <= 0) { h = 0; src = SRC_STOP; }
            Hc[k] = h;
            const unsigned byte = src | ((fxm >> k) & 1u) * TB_FEXT | ((oxm >> k) & 1u) * TB_OEXT |
                                  ebit * TB_EEXT | qbit * TB_QEXT;
            if ((k & 3) == 0) tbw[k >> 2] = byte; else tbw[k >> 2] |= byte << (8 * (k & 3));
            // pad columns (j > L) can never strictly beat a real cell and lose every tie (their
            // letter never matches), so the local-mode maximum needs no column test
            if (SW) rowkey = max(rowkey, (h << 5) | (31 - k));
            const int c1 = h + g, c2 = E + e;
            ebit = c2 > c1; E = ebit ? c2 : c1;
            if (CVX) { const int d1 = h + q, d2 = Q + c; qbit = d2 > d1; Q = qbit ? d2 : d1; }
            if (SXG_GROUP_END(k)) SXG_PIN(SXG_COLS(Hc, k), "+v"(tbw[k >> 2]), "+v"(E), "+v"(Q), "+v"(ebit), "+v"(qbit), "+v"(rowkey));
        }
        if (SW) {
            if ((rowkey >> 5) > best) { best = rowkey >> 5; bi = i; bj = 31 - (rowkey & 31); }
        } else if (flags & ROW_SINK) {  // global: H[i][L] of sink rows, owned by one lane
            int hL = NEG * 2;
#pragma unroll
            for (int k = 0; k < W; ++k) hL = (k == kL) ? Hc[k] : hL;
            if (kL >= 0 && kL < W && (bi < 0 || hL > best)) { best = hL; bi = i; bj = kL; }
        }
        // hand H of my last column and the ext flags of the next column to the right neighbour
        int xh = Hc[W - 1], xb = (int)(ebit | (qbit << 1));
        int lh = sxg_wave_shr1(xh, 0), lb = sxg_wave_shr1(xb, 0);
        if (NW > 1) {
            if (lane == 63) { xch_h[wv] = xh; xch_b[wv] = xb; }
            SXG_ROW_BARRIER();  // B2
            if (lane == 0 && wv > 0) { lh = xch_h[wv - 1]; lb = xch_b[wv - 1]; }
        }
        if (t == 0) { lh = NEG; lb = 0; }
        tbw[0] |= ((unsigned)(lb & 1) * TB_EEXT) | ((unsigned)((lb >> 1) & 1) * TB_QEXT);

        // ---- stores
        {
            SXG_GLOBAL unsigned* dst = (SXG_GLOBAL unsigned*)(g_tb + (size_t)i * Lpad);  // W bytes per lane
#pragma unroll
            for (int k4 = 0; k4 < W / 4; ++k4) (dst + k4)[uj0 / 4] = tbw[k4];   // (non-temporal, as the packed sweeps' plane: 125.2 -> 126.0 blocks/s, nothing)
        }
        if (flags & ROW_STORE) {
            SXG_GLOBAL Word* dst = g_pool + (size_t)myslot * Lpad;
#pragma unroll
            for (int k = 0; k < W; ++k) (dst + k)[uj0] = RWt::pack(Hc[k], Fp[k], Op[k]);
        }
#pragma unroll
        for (int k = 0; k < W; ++k) Hp[k] = Hc[k];
        Hleft = lh;
        if (NW == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }

    // ---- end cell: greatest score, then smallest row, then smallest column
    unsigned long long key = 0;
    bj += j0;
    if (bi >= 0)
        key = ((unsigned long long)(unsigned)(best + (1 << 27)) << 35) |
              ((unsigned long long)(0xFFFFFu - (unsigned)bi) << 15) | (unsigned long long)(0x7FFFu - (unsigned)bj);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(key, d);
        key = o > key ? o : key;
    }
    __syncthreads();
    unsigned long long* kl = (unsigned long long*)lds;
    if (lane == 0) kl[wv] = key;
    __syncthreads();
    key = kl[0];
    for (int x = 1; x < NW; ++x) key = kl[x] > key ? kl[x] : key;
    __syncthreads();
    if (key == 0) { res.best = 0; res.bi = -1; res.bj = -1; }
    else {
        res.best = (int)(unsigned)(key >> 35) - (1 << 27);
        res.bi = (int)(0xFFFFFu - (unsigned)((key >> 15) & 0xFFFFFu));
        res.bj = (int)(0x7FFFu - (unsigned)(key & 0x7FFFu));
    }
}

// ---------------------------------------------------------------------------------------
// Traceback (S5): replay of the recorded choices by one lane.  Emits pairs in REVERSE order
// into pair_row/pair_pos (row = 1-based row index or 0 for "none", pos or -1) when PAIRS,
// and/or fills posnode[pos] = node id of the aligned node.  Returns the number of pairs.
// Which predecessor (list ordinal) owns state `which` (0 = D, 1 = F, 2 = O) of cell (row r,
// column j): the last fold step that took the state over, else predecessor #0.  For two
// predecessors the single step's bit IS the ordinal (dp_fill keeps that true when it folds #1
// first).
__device__ __forceinline__ int winner_ordinal(const RowsView& R, const DpBuffers& B, const int T, const int W,
                                              const int r, const int np, const int j, const int which) {
    const int tx = R.tbx[r];
    const int lane_t = j / W, k = j - lane_t * W;
    for (int x = np - 1; x >= 1; --x) {
        const unsigned m = B.steps[((size_t)(tx + x - 1) * 3 + which) * T + lane_t];
        if ((m >> k) & 1u) return x;
    }
    return 0;
}

// (views by value: a reference to the kernel's private copy makes the out-of-line callee read it through FLAT pointers
// whose private-aperture null check trips an AMDGPU back-end assertion for some strip widths -- which ones changes with
// every unrelated edit of the kernel)
template <bool PAIRS>
__device__ __noinline__ int traceback(const RowsView R, const DpBuffers B, const int T, const int W, const int sw, int i, int j,
                         int32_t* posnode, int32_t* pair_row, int32_t* pair_pos) {
    const int Lpad = T * W;
    int n = 0, st = SRC_STOP;
    for (;;) {
        if (i == 0) {
            if (j == 0 || sw) break;  // H[0][j] = 0 ends a local alignment
            if (PAIRS) { pair_row[n] = 0; pair_pos[n] = j - 1; }
            ++n; --j;
            continue;
        }
        const int r = i - 1;
        // one round trip per step: the cell's byte and the row descriptor (first two predecessors)
        const unsigned tbyte = B.tb[(size_t)i * Lpad + j];
        const int4 d0 = *(const int4*)(R.meta + 8 * (size_t)r), d1 = *(const int4*)(R.meta + 8 * (size_t)r + 4);
        const int node = R.row_node[r];
        const int pb = d0.x, np = d0.y & 0xffff, q0 = d0.z, q1 = d1.x;
        auto pred_of = [&](int which) -> int {
            if (np == 0) return 0;
            if (np == 1) return q0;
            const int ord = winner_ordinal(R, B, T, W, r, np, j, which);
            return ord == 0 ? q0 : (ord == 1 ? q1 : R.preds[pb + ord]);
        };
        if (st == SRC_STOP) {
            const int src = tbyte & 7;
            if (src == SRC_STOP) break;
            if (src == SRC_D) {
                if (PAIRS) { pair_row[n] = i; pair_pos[n] = j - 1; }
                if (posnode) posnode[j - 1] = node;
                ++n;
                i = pred_of(0);
                --j;
            } else st = src;
        } else if (st == SRC_F || st == SRC_O) {
            const int ext = st == SRC_F ? (tbyte & TB_FEXT) : (tbyte & TB_OEXT);
            if (PAIRS) { pair_row[n] = i; pair_pos[n] = -1; }
            ++n;
            i = pred_of(st == SRC_F ? 1 : 2);
            if (!ext) st = SRC_STOP;
        } else {
            const int ext = st == SRC_E ? (tbyte & TB_EEXT) : (tbyte & TB_QEXT);
            if (PAIRS) { pair_row[n] = 0; pair_pos[n] = j - 1; }
            ++n; --j;
            if (!ext) st = SRC_STOP;
        }
    }
    return n;
}

}  // namespace sxg
