// sxg_poa.hip -- kernels + host side of the C ABI declared in include/sxg_poa.h.
//
// One persistent workgroup ("slot") owns one smoothxg block at a time and runs the whole
// sequential chain of src/smooth.cpp:760-769 on the device:
//     for every sequence: rows <- graph; DP fill; traceback; fuse; re-rank
// Slots pull blocks from a cost-sorted queue (largest first), so thousands of blocks are
// processed per launch with no host round trip between the alignments of a block.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <vector>

#include "../../include/sxg_poa.h"
#include <rccl/rccl.h>   // (types and prototypes only: the library is opened on first use, see rccl_api)
#include <dlfcn.h>
#include "poa_kernels.hip.h"       // slot layout, launch arguments, the persistent kernels (device side)
#include "poa_kern_tables.hip.h"   // kernel classes by geometry; instantiated in the kern_*.hip translation units


// ---------------------------------------------------------------------------------------
// Block graphs (sxg_poa_batch_in::want_block_graph): after the POA kernels, one workgroup per block turns the block's POA
// result -- letters, edges, one node id per base, consensus -- into its normalised block graph (poa_bgraph_dev.h: trim,
// path-supported edges, unchop, topological re-numbering, compact step lists) in a slot-private scratch arena.
struct BgArgs {
    const int32_t* blk_off; const int64_t* seq_off; const uint8_t* bases; const int32_t* paths;
    const uint8_t* node_code; const int32_t* edge_tail; const int32_t* edge_head; const int32_t* cons;
    const int32_t *nn, *ne, *nc; const int32_t* trim; int cons_mode;
    const int32_t* work; int n_work; int32_t* queue;
    uint8_t* arena; size_t slot_bytes; int capV, capE, capC, capT;
    const int64_t *node_o, *edge_o;   // where block b's nodes / edges go in the output arrays
    int32_t *o_len, *o_outdeg; uint8_t* o_indeg; char* o_seq; int32_t* o_eto; int32_t* o_steps; int32_t* o_nsteps;
    int32_t* o_cons; int32_t* o_counts;
};
constexpr int BG_THREADS = 256, BG_HEAP = 3072;
static size_t bg_slot_bytes(int capV, int capE, int capC, int capT) {
    return 4 * (27 * ((size_t)capV + 2) + 3 * ((size_t)capE + capC + 2) + ((size_t)capC + 2) + ((size_t)capT + 2) + 4) + 256;
}
struct BgCtx : WgCtx {
    int* hp;
    __device__ __forceinline__ int load_fresh(const int32_t* p) const { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ int* heap() const { return hp; }
    __device__ __forceinline__ int heap_cap() const { return BG_HEAP; }
};
__global__ __launch_bounds__(BG_THREADS) void block_graph_kernel(const BgArgs A) {
    __shared__ int s_scan[64];
    __shared__ int s_heap[BG_HEAP];
    __shared__ int s_work;
    BgCtx ctx;
    ctx.lds = (sxg_lds_int*)s_scan;
    ctx.hp = s_heap;
    const int t = threadIdx.x;
    int32_t* w = (int32_t*)(A.arena + (size_t)blockIdx.x * A.slot_bytes);
    const size_t C = (size_t)A.capV + 2, EC = (size_t)A.capE + A.capC + 2;
    BgScratch W;
    auto take = [&](size_t n) { int32_t* q = w; w += n; return q; };
    W.vis = take(C); W.front = take(C); W.back = take(C); W.ecnt = take(C); W.eoff = take(C);
    W.ehead = take(EC); W.eused = take(EC);
    W.outdeg = take(C); W.indeg = take(C); W.only = take(C); W.nxt = take(C); W.prv = take(C);
    W.pj0 = take(C); W.pj1 = take(C); W.pd0 = take(C); W.pd1 = take(C); W.cid = take(C);
    W.head_of = take(C); W.tail_of = take(C); W.clen = take(C); W.indc = take(C); W.coff = take(C); W.newid = take(C);
    W.csucc = take(EC); W.cc = take((size_t)A.capC + 2); W.tmp = take((size_t)A.capT + 2);
    W.lenN = take(C); W.odN = take(C); W.soffN = take(C); W.eoffN = take(C); W.flag = take(4);
    for (;;) {
        __syncthreads();
        if (t == 0) s_work = atomicAdd(A.queue, 1);
        __syncthreads();
        const int wi = s_work;
        if (wi >= A.n_work) break;
        const int b = A.work[wi];
        const int s0 = A.blk_off[b], s1 = A.blk_off[b + 1];
        const int64_t base0 = A.seq_off[s0];
        BgIn I;
        I.node_code = A.node_code + base0; I.V = A.nn[b]; I.E = A.ne[b];
        I.e_tail = A.edge_tail + base0; I.e_head = A.edge_head + base0;
        I.paths = A.paths + base0; I.bases = A.bases + base0; I.seq_off = A.seq_off; I.s0 = s0; I.s1 = s1;
        I.trim = A.trim ? A.trim[b] : 0;
        I.cons = A.cons ? A.cons + base0 : nullptr; I.n_cons = A.cons ? A.nc[b] : 0; I.cons_mode = A.cons ? A.cons_mode : 0;
        BgOut O;
        const int64_t no = A.node_o[b], eo = A.edge_o[b];
        O.node_len = A.o_len + no; O.node_outdeg = A.o_outdeg + no; O.node_indeg = A.o_indeg + no; O.seq = A.o_seq + no;
        O.eto = A.o_eto + eo; O.steps = A.o_steps + base0; O.nsteps = A.o_nsteps + s0; O.cons_steps = A.o_cons + no;
        O.counts = A.o_counts + (size_t)BGC_N * b;
        block_graph(ctx, I, W, O);
    }
}

// dense <- worst-case gather (one workgroup per block, grid-stride over blocks)
// upload: letters above 4 read as N (4).  16 bytes per thread and step; the buffer is allocated with 16 bytes of slack.
__global__ void clamp_bases_kernel(uint8_t* bases, const int64_t n) {
    const int64_t n16 = (n + 15) / 16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
        u32x4 v = ((const u32x4*)bases)[i];
        unsigned w[4] = {v.x, v.y, v.z, v.w};
        bool any = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned o = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) { const unsigned c = (w[k] >> (8 * b)) & 0xffu; o |= (c > 4u ? 4u : c) << (8 * b); }
            any |= o != w[k];
            w[k] = o;
        }
        if (any) ((u32x4*)bases)[i] = u32x4{w[0], w[1], w[2], w[3]};
    }
}

template <class Tv>
__global__ void gather_kernel(const Tv* __restrict__ src, Tv* __restrict__ dst, const int64_t* __restrict__ src_off,
                              const int64_t* __restrict__ dst_off, int n) {
    for (int b = blockIdx.x; b < n; b += gridDim.x) {
        const int64_t s = src_off[b], d = dst_off[b], cnt = dst_off[b + 1] - d;
        for (int64_t i = threadIdx.x; i < cnt; i += blockDim.x) dst[d + i] = src[s + i];
    }
}

// =======================================================================================
// host side
// =======================================================================================

// Error text: per calling thread (as the header promises) plus the most recent one of ANY thread -- a caller that ran the
// engine on a worker thread (sxg_smooth's chunk pipeline) and asks from its own thread gets that one instead of nothing.
static thread_local std::string g_err;
#include <mutex>
static std::mutex g_err_mu;
static std::string g_err_any;
static thread_local std::string g_err_ret;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    { std::lock_guard<std::mutex> lk(g_err_mu); g_err_any = msg; }
    return code;
}
#define HIPCHK(x)                                                                              \
    do {                                                                                       \
        hipError_t _e = (x);                                                                   \
        if (_e != hipSuccess)                                                                  \
            return fail(_e == hipErrorOutOfMemory ? SXG_E_NOMEM : SXG_E_NODEVICE,              \
                        std::string(#x) + ": " + hipGetErrorString(_e));                       \
    } while (0)

// Geometry choice.  Inside a workgroup all waves meet at two barriers per row, so the wave
// count should load the four SIMDs of a CU evenly: 1, 2, 3, 4, 8, 12 or 16 waves.  Among the
// (W, NW) pairs that cover the sequence pick the one with the fewest padded columns, then the
// wider strip (less per-row overhead).  Strip widths are bounded by VGPRs: 32-bit sweep 16
// columns (~165 VGPRs, <= 512 threads) / 12 (128 VGPRs); packed sweep 12 (~152) / 8 (124).
// Long local alignments (sequences of 12-26 kbp: smoothxg runs with -l 13k cut at 2 * 13k) get 16-wave workgroups of the
// packed sweep with 10, 12 or 13 columns per strip (128 VGPRs, a handful of spill slots): 20 480 / 24 576 / 26 624 columns.
static bool variant_for_len(int maxlen, int rm, Variant* v, bool sw = false) {
    static const int kNW[] = {1, 2, 3, 4, 8, 12, 16};
    if (rm == 2) {   // development knob: SXG_POA_FORCE_P16="W,NW" forces one packed geometry (A/B runs of a single-class build)
        if (const char* e = getenv("SXG_POA_FORCE_P16")) {
            int fw = 0, fnw = 0;
            if (sscanf(e, "%d,%d", &fw, &fnw) == 2 && 128L * fnw * fw >= maxlen + 1 && (fnw <= 4 || fnw == 8 || fnw == 12 || fnw == 16)) {
                *v = Variant{fw, fnw, fnw <= 4 ? 64 * fnw : (fnw <= 8 ? 512 : 1024), rm};
                return true;
            }
        }
    }
    // (narrow strips, 4-7 columns, for sequences below 1 kbp -- pggb's -l 700 ... 1100 -- in workgroups of up to 4 waves)
    static const int kW32[] = {16, 12, 8}, kW16[] = {13, 12, 11, 10, 9, 8, 7, 6, 5, 4};
    const int* ws = rm == 2 ? kW16 : kW32;
    const int nws = rm == 2 ? 10 : 3;
    const int need = maxlen + 1;
    long best_cols = -1;
    for (int wi = 0; wi < nws; ++wi)
        for (int NW : kNW) {
            const int W = ws[wi];
            const long cols = 64L * NW * W * (rm == 2 ? 2 : 1);
            if (cols < need) continue;
            const bool wide = rm == 2 ? W > 8 : W > 12;   // needs > 128 VGPRs unless squeezed
            const bool long_class = rm == 2 && sw && NW == 16 && (W == 10 || W == 12 || W == 13);
            if (W == 13 && !long_class) continue;
            if (rm == 2 && W < 8 && NW > 4) continue;
            if (wide && NW > 8 && !long_class) continue;
            if (best_cols < 0 || cols < best_cols) {
                best_cols = cols;
                *v = Variant{W, NW, (rm == 2 && NW <= 4) ? 64 * NW : (NW <= 4 ? 256 : (NW <= 8 ? 512 : 1024)), rm};   // (packed: one and two waves have classes of their own)
            }
            break;  // larger NW for this W only adds padding
        }
    return best_cols >= 0;
}

// On-chip copies of stored rows a packed-sweep workgroup gets (SlotLayout::lds_rows): what is left of its share of the CU's
// 160 KB of LDS when as many workgroups share the CU as its registers allow (128 VGPRs: 16 waves per CU).
// (Round 4: giving the workgroups of a launch that does not fill the chip -- 1000 two-wave blocks: four per CU where eight
//  fit -- the LDS the absent ones leave, i.e. 8 on-chip rows instead of 2-3, was measured on c2: 59.6 ms against 57.4 ms.  Dropped.)
static int p16_lds_rows(const int T, const int W) {
    if (const char* e = getenv("SXG_POA_LDS_ROWS")) return std::max(0, std::min(8, atoi(e)));
    const int wg_per_cu = std::max(1, 16 / std::max(T / 64, 1));
    const int share = (160 * 1024) / wg_per_cu - 512;   // (allocation granularity)
    const int rows = (share - dp16_lds_bytes(T, W, 0)) / dp16_row_bytes(T, W);
    return std::max(0, std::min(8, rows));
}

// Plane cell format of the packed sweep for a score set: 2-byte delta codes when the three fields fit 16 bits (poa_dp16.hip.h,
// P16Delta), the 4-byte cells of rounds 2-4 otherwise.  SXG_POA_CELL_BYTES=4 forces the latter (A/B runs, tests).
static int plane_cell_bytes(const Scoring& S) {
    if (const char* e = getenv("SXG_POA_CELL_BYTES")) if (atoi(e) == 4) return 4;
    return p16_delta_fits(S) ? 2 : 4;
}
// ... and of the banded sweep (round 6): 2-byte codes for LOCAL alignments whose delta code fits -- every cell of a band is a real
// score there; a global alignment's bands hold "minus infinity" cells, whose steps no code holds.  SXG_POA_BAND_CELL_BYTES=4: the
// 4-byte cells of rounds 2-5 (A/B runs, tests).
static int band_cell_bytes(const Scoring& S) {
    if (const char* e = getenv("SXG_POA_BAND_CELL_BYTES")) if (atoi(e) == 4) return 4;
    return S.sw ? plane_cell_bytes(S) : 4;
}

// Score ranges.  Local alignment: 0..m*L.  Global: additionally down to the all-gap path through
// `rows` graph rows (sxg_gap_cost).  `rows` is exact for the align-only API; for whole blocks the
// graph grows while the block runs, so the host assumes the smallest possible graph (one node per base
// of the longest sequence) and the kernel re-checks with the true row count before every sweep,
// answering ST_RANGE_OVERFLOW when it no longer fits: the block is then re-run one step wider.
static long score_floor(const Scoring& S, int maxlen, int rows) {
    if (S.sw) return 0;
    return -(sxg_gap_cost(S.g, S.e, S.q, S.c, rows) + sxg_gap_cost(S.g, S.e, S.q, S.c, maxlen));
}
// int16 H in the row words of the 32-bit sweep
static bool h16_safe(const Scoring& S, int maxlen, int rows) {
    return (long)std::abs(S.m) * maxlen < 30000 && score_floor(S, maxlen, rows) < 30000;
}
// Packed-int16 sweep: every reachable score and every intermediate (score +- one penalty, difference
// of two scores) must stay inside int16 with NEGP = -16384 as "-inf".
// clamp_ok: the caller can re-run a block whose walk met a clamped cell (whole blocks on the full matrix: P16_NWFLOOR in
// poa_dp16.hip.h) -- a global alignment then only needs its POSITIVE range to fit; the align-only API and the banded sweep
// keep the strict rule (no re-run / -inf cells of their own).  SXG_POA_NO_NW_CLAMP=1: strict everywhere (A/B runs).
static bool p16_safe(const Scoring& S, int maxlen, int rows, bool clamp_ok = false) {
    if (getenv("SXG_POA_NO_PACKED")) return false;
    // stored rows keep H - max(H+g, F+e) and H - max(H+q, O+c) in one byte each
    if (std::abs(S.g) > 120 || std::abs(S.q) > 120) return false;
    // local alignment: every existing cell has 0 <= H <= m * L and F, O, E, Q >= -|q|; the range is one-sided
    if (S.sw) return (long)std::abs(S.m) * maxlen < 30000;
    if ((long)std::abs(S.m) * maxlen >= 15800) return false;
    // (the walk may dip to -16 000 + m L + m before it is called clamped: 1 500 below zero at the limit -- a global alignment of
    //  related sequences stays far above; one that does not is re-run on the 32-bit sweep while that sweep reaches it, 12 287 letters)
    if (clamp_ok && !getenv("SXG_POA_NO_NW_CLAMP")) return (long)std::abs(S.m) * maxlen + std::abs(S.m) < 14500;
    return score_floor(S, maxlen, rows) < 15800;
}
// narrowest mode >= `at_least` (2 = packed < 0 = int16 row words < 1 = int32 row words)
static int row_mode(const Scoring& S, int maxlen, int rows, int at_least = 2, bool clamp_ok = false) {
    if (at_least == 2 && p16_safe(S, maxlen, rows, clamp_ok)) return 2;
    if (at_least != 1 && h16_safe(S, maxlen, rows)) return 0;
    return 1;
}

// SXG_POA_DEBUG: wall-clock laps of the host side (stderr)
struct HostLaps {
    bool on = getenv("SXG_POA_DEBUG") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[sxg] host %-22s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return SXG_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            want = bytes;
            e = hipMalloc(&p, want);
        }
        if (e != hipSuccess) { p = nullptr; return fail(SXG_E_NOMEM, "hipMalloc failed for " + std::to_string(bytes) + " bytes"); }
        cap = want;
        return SXG_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class Tp> Tp* as() const { return (Tp*)p; }
};

struct BlockMeta {
    int maxlen = 0, nseq = 0, rm = 0; bool fits = false; Variant variant{16, 1, 256, 0};
    int tier = 0;      // arena capacity tier the block runs at next (raised by ROWS/POOL/TBX overflow only)
    bool wide_band = false;   // packed sweep re-run with a traceback plane that keeps EVERY strip (after ST_BAND_MISS)
    bool no_wide = false;     // that plane did not fit the arena budget: the block takes the widening ladder instead
    int64_t sumlen = 0;
    double cost = 0;
    bool cvx = false, sw = true;
    Scoring S;
};

struct PlanRes {
    DevBuf arena, work, queue, est;
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

// Pinned host memory for the one big array of a download (the block graphs' step lists: 1.3 GB on the headline batch).  A
// copy into pageable memory runs at about half the link's rate (the runtime stages it); pinning 1.3 GB takes longer than the
// copy saves, so the buffers are pinned ONCE and lent out: a result holds one until sxg_poa_batch_free (any thread), the pool
// lives as long as the handle or the last result that borrowed from it.  SXG_POA_NO_PINNED=1 switches it off.
struct PinPool {
    struct Buf { void* p; size_t cap; bool busy; };
    std::mutex mu;
    std::vector<Buf> bufs;
    void* acquire(size_t bytes) {   // nullptr: nothing to lend (the caller allocates pageable memory)
        std::lock_guard<std::mutex> g(mu);
        for (auto& b : bufs) if (!b.busy && b.cap >= bytes) { b.busy = true; return b.p; }
        for (size_t i = 0; i < bufs.size(); ++i)
            if (!bufs[i].busy) { (void)hipHostFree(bufs[i].p); bufs.erase(bufs.begin() + (long)i); break; }   // (an idle one that is too small)
        if (bufs.size() >= 4) return nullptr;
        void* q = nullptr;
        const size_t cap = bytes + bytes / 8;
        if (hipHostMalloc(&q, cap, hipHostMallocDefault) != hipSuccess || !q) { (void)hipGetLastError(); return nullptr; }
        bufs.push_back(Buf{q, cap, true});
        return q;
    }
    void release(void* p) {
        std::lock_guard<std::mutex> g(mu);
        for (auto& b : bufs) if (b.p == p) b.busy = false;
    }
    void prewarm(size_t bytes) { if (void* p = acquire(bytes)) release(p); }   // (pin now, lend later)
    ~PinPool() { for (auto& b : bufs) (void)hipHostFree(b.p); }
};
struct sxg_poa_handle {
    std::vector<PlanRes*> planres;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int num_cu = 256;
    uint64_t mem_budget = 0;
    // resident batch
    bool have_batch = false, executed = false;
    int n_blocks = 0;
    int64_t n_seqs = 0, n_bases = 0;
    int want_consensus = 0, want_msa = 0, per_block_params = 0;
    std::vector<int32_t> h_blk_off;
    std::vector<int64_t> h_seq_off;
    std::vector<sxg_poa_params> h_params;
    std::vector<BlockMeta> meta;
    DevBuf d_blk_off, d_seq_off, d_bases, d_weights, d_params;
    bool has_weights = false;
    DevBuf d_status, d_nn, d_ne, d_nc, d_node_code, d_node_rank, d_node_group, d_edge_tail, d_edge_head, d_edge_w,
        d_paths, d_score, d_cells, d_cons, d_work, d_queue, d_arena, d_blk_cycles;
    DevBuf d_tmp_a, d_tmp_b, d_tmp_c, d_tmp_d;
    DevBuf d_board;   // the per-CU progress board (sxg_balance_prio) the launches of a round share
    // block graphs (want_block_graph): inputs, outputs in per-block layouts, the per-block counts of the last execute
    int want_block_graph = 0, bg_cons_visited_only = 0;
    std::shared_ptr<PinPool> pins;   // pinned host buffers lent to results (created with the handle)
    std::thread pin_thread;          // pins the buffer of the coming download while the kernels run (joined before it is needed)
    bool bg_done = false;
    DevBuf d_trim, d_bg_no, d_bg_eo, d_bg_len, d_bg_od, d_bg_id, d_bg_seq, d_bg_eto, d_bg_steps, d_bg_nsteps, d_bg_cons, d_bg_counts,
        d_bg_work, d_bg_queue, d_bg_arena;
    std::vector<int64_t> bg_node_o, bg_edge_o;
    std::vector<int32_t> bg_counts;
    sxg_poa_stats stats{};
    // multi-GPU (sxg_poa_batch_run_sharded): communicator of this rank, result blob of the local shard, and -- on the
    // root -- the blobs of the other ranks, delivered by RCCL
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    int nranks = 1, rank = 0;
    DevBuf d_blob, d_recv, d_counts;
    // staged sharded run: the deal of the last sxg_poa_batch_upload_sharded, the counts every rank announced in the last
    // sxg_poa_batch_execute_sharded and where the root put their blobs
    std::vector<std::vector<int32_t>> sh_parts;
    std::vector<int64_t> sh_counts;
    std::vector<size_t> sh_at;
    bool sh_dealt = false, sh_exchanged = false;
    uint64_t sh_bytes_received = 0;
    int sh_ranks_seen = 0;
    double sh_pack_ms = 0, sh_exchange_ms = 0;   // last sxg_poa_batch_execute_sharded: packing the blob; size all-gathers + blob exchange (waits for the slowest rank)
};

// ---------------------------------------------------------------------------------------
// ROCTx ranges around the phases of the C ABI (the reference brackets its phases with timers on stderr;
// here `rocprofv3 --marker-trace` shows upload / execute / pack / exchange / download next to the kernels).  The marker
// library is looked up at first use -- rocprofiler-sdk's, then roctracer's -- and the ranges are no-ops without it: the
// engine does not link against a profiler.
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        if (getenv("SXG_POA_NO_ROCTX")) return;
        void* lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return;
        push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
        pop = (int (*)())dlsym(lib, "roctxRangePop");
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
static Roctx& roctx() { static Roctx r; return r; }
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
    ~RoctxRange() { if (on) roctx().pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};
}  // namespace
extern "C" int sxg_poa_roctx_available(void) { return roctx().push != nullptr ? 1 : 0; }

// A streaming copy on this device, timed with HIP events on the engine's stream: what SURVEY 8(d) asks to be printed beside the
// 8 TB/s of the data sheet (bench.py: roofline.hbm.copy_GBps_measured).  16 bytes per lane and step, grid-stride, non-temporal
// both ways; bytes read + bytes written over the mean of `reps` launches after one warm-up.
__global__ __launch_bounds__(256) void sxg_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, const size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
extern "C" int sxg_poa_measure_copy(sxg_poa_handle* h, uint64_t bytes, int reps, double* gbps) {
    if (!h || !gbps) return fail(SXG_E_INVALID, "NULL argument");
    *gbps = 0;
    HIPCHK(hipSetDevice(h->device));
    const size_t n16 = (size_t)std::max<uint64_t>(bytes, 1u << 20) / 16;
    DevBuf a, b;
    if (int rc = a.ensure(n16 * 16)) return rc;
    if (int rc = b.ensure(n16 * 16)) return rc;
    HIPCHK(hipMemsetAsync(a.p, 1, n16 * 16, h->stream));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    const int grid = std::max(h->num_cu, 1) * 16;
    reps = std::max(reps, 1);
    hipLaunchKernelGGL(sxg_copy_kernel, dim3((unsigned)grid), dim3(256), 0, h->stream, a.as<u32x4>(), b.as<u32x4>(), n16);
    HIPCHK(hipEventRecord(e0, h->stream));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(sxg_copy_kernel, dim3((unsigned)grid), dim3(256), 0, h->stream, a.as<u32x4>(), b.as<u32x4>(), n16);
    HIPCHK(hipEventRecord(e1, h->stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    a.release(); b.release();
    if (ms > 0) *gbps = 2.0 * (double)(n16 * 16) * reps / ((double)ms * 1e-3) / 1e9;
    return SXG_OK;
}

extern "C" int sxg_poa_abi_version(void) { return SXG_POA_ABI_VERSION; }
extern "C" const char* sxg_poa_last_error(void) {
    if (!g_err.empty()) return g_err.c_str();
    { std::lock_guard<std::mutex> lk(g_err_mu); g_err_ret = g_err_any; }
    return g_err_ret.c_str();
}
extern "C" int sxg_poa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int sxg_poa_create(int device, sxg_poa_handle** out) {
    if (!out) return fail(SXG_E_INVALID, "out is NULL");
    *out = nullptr;
    // Batches of mixed lengths run one launch per geometry on separate streams.  The HIP runtime maps
    // streams onto 4 hardware queues by default and launches sharing a queue run one after the other
    // (measured on a 12-geometry batch: 44.5 s vs 17.9 s with 16 queues).  Takes effect only if the
    // runtime is not initialised yet; callers that initialise HIP first should export it themselves.
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(SXG_E_NODEVICE, "no HIP device available");
    if (device < 0 || device >= n) return fail(SXG_E_INVALID, "device index out of range");
    HIPCHK(hipSetDevice(device));
    sxg_poa_handle* h = new sxg_poa_handle();
    h->device = device;
    h->pins = std::make_shared<PinPool>();
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    h->num_cu = prop.multiProcessorCount;
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&h->ev0));
    HIPCHK(hipEventCreate(&h->ev1));
    *out = h;
    return SXG_OK;
}

static void release_all(sxg_poa_handle* h) {
    DevBuf* bufs[] = {&h->d_blk_off, &h->d_seq_off, &h->d_bases, &h->d_weights, &h->d_params, &h->d_status, &h->d_nn,
                      &h->d_ne, &h->d_nc, &h->d_node_code, &h->d_node_rank, &h->d_node_group, &h->d_edge_tail,
                      &h->d_edge_head, &h->d_edge_w, &h->d_paths, &h->d_score, &h->d_cells, &h->d_cons, &h->d_work,
                      &h->d_queue, &h->d_arena, &h->d_blk_cycles, &h->d_tmp_a, &h->d_tmp_b, &h->d_tmp_c, &h->d_tmp_d, &h->d_board, &h->d_trim, &h->d_bg_no, &h->d_bg_eo,
                      &h->d_bg_len, &h->d_bg_od, &h->d_bg_id, &h->d_bg_seq, &h->d_bg_eto, &h->d_bg_steps, &h->d_bg_nsteps, &h->d_bg_cons,
                      &h->d_bg_counts, &h->d_bg_work, &h->d_bg_queue, &h->d_bg_arena};
    for (DevBuf* b : bufs) b->release();
}

extern "C" void sxg_poa_comm_destroy(sxg_poa_handle* h);
extern "C" void sxg_poa_destroy(sxg_poa_handle* h) {
    if (!h) return;
    if (h->pin_thread.joinable()) h->pin_thread.join();
    (void)hipSetDevice(h->device);
    sxg_poa_comm_destroy(h);
    h->d_blob.release(); h->d_recv.release(); h->d_counts.release();
    release_all(h);
    for (PlanRes* r : h->planres) {
        r->arena.release(); r->work.release(); r->queue.release(); r->est.release();
        if (r->e0) (void)hipEventDestroy(r->e0);
        if (r->e1) (void)hipEventDestroy(r->e1);
        if (r->stream) (void)hipStreamDestroy(r->stream);
        delete r;
    }
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int sxg_poa_set_memory_budget(sxg_poa_handle* h, uint64_t bytes) {
    if (!h) return fail(SXG_E_INVALID, "handle is NULL");
    h->mem_budget = bytes;
    return SXG_OK;
}

extern "C" int sxg_poa_get_stats(sxg_poa_handle* h, sxg_poa_stats* out) {
    if (!h || !out) return fail(SXG_E_INVALID, "NULL argument");
    *out = h->stats;
    return SXG_OK;
}

static uint64_t arena_budget(sxg_poa_handle* h) {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return 8ull << 30;
    uint64_t avail = (uint64_t)fr + h->d_arena.cap;  // the arenas we already hold can be reused
    for (PlanRes* r : h->planres) avail += r->arena.cap;
    uint64_t b = avail / 4 * 3;
    if (h->mem_budget && h->mem_budget < b) b = h->mem_budget;
    return b;
}

// ---------------------------------------------------------------------------------------
extern "C" int sxg_poa_batch_upload(sxg_poa_handle* h, const sxg_poa_batch_in* in) {
    if (!h || !in) return fail(SXG_E_INVALID, "NULL argument");
    if (in->n_blocks < 0 || (in->n_blocks > 0 && (!in->blk_off || !in->seq_off || !in->params)))
        return fail(SXG_E_INVALID, "batch_in has NULL arrays");
    HIPCHK(hipSetDevice(h->device));
    const RoctxRange range_("sxg_poa_batch_upload");
    HostLaps laps;
    h->have_batch = false; h->executed = false;
    const int nb = in->n_blocks;
    if (nb > 0 && in->blk_off[0] != 0) return fail(SXG_E_INVALID, "blk_off[0] must be 0");
    const int64_t ns = nb ? in->blk_off[nb] : 0;
    for (int b = 0; b < nb; ++b)
        if (in->blk_off[b + 1] < in->blk_off[b]) return fail(SXG_E_INVALID, "blk_off not monotone");
    if (ns > 0 && in->seq_off[0] != 0) return fail(SXG_E_INVALID, "seq_off[0] must be 0");
    for (int64_t s = 0; s < ns; ++s)
        if (in->seq_off[s + 1] < in->seq_off[s]) return fail(SXG_E_INVALID, "seq_off not monotone");
    const int64_t nbases = ns ? in->seq_off[ns] : 0;
    if (nbases > 0 && !in->bases) return fail(SXG_E_INVALID, "bases is NULL");
    const int np = in->per_block_params ? nb : 1;
    for (int k = 0; k < np && nb > 0; ++k) {
        const sxg_poa_params& p = in->params[k];
        if (p.m < 0 || p.n > 0 || p.g > 0 || p.e > 0 || p.q > 0 || p.c > 0 || (p.mode & ~(1 | SXG_ORDER_SPOA)) != 0 || p.banded > 2)
            return fail(SXG_E_INVALID, "scores must follow spoa's sign convention (m>=0, others <=0), mode 0|1");
    }
    h->n_blocks = nb; h->n_seqs = ns; h->n_bases = nbases;
    h->want_consensus = in->want_consensus; h->want_msa = in->want_msa;
    h->want_block_graph = in->want_block_graph < 0 || in->want_block_graph > 3 ? 0 : in->want_block_graph;
    if (h->want_block_graph >= 2 && in->want_msa) h->want_block_graph = 1;   // (the MSA is formatted from the per-base paths)
    h->bg_cons_visited_only = in->bg_consensus_visited_only ? 1 : 0;
    h->bg_done = false;
    h->per_block_params = in->per_block_params;
    h->h_blk_off.assign(in->blk_off, in->blk_off + nb + 1);
    h->h_seq_off.assign(in->seq_off, in->seq_off + ns + 1);
    h->h_params.assign(in->params, in->params + np);
    h->meta.assign(nb, BlockMeta());
    for (int b = 0; b < nb; ++b) {
        BlockMeta& m = h->meta[b];
        m.S = normalise(h->h_params[in->per_block_params ? b : 0]);
        m.cvx = m.S.convex; m.sw = m.S.sw;
        m.nseq = in->blk_off[b + 1] - in->blk_off[b];
        double prev = 0;
        for (int s = in->blk_off[b]; s < in->blk_off[b + 1]; ++s) {
            const int64_t len = in->seq_off[s + 1] - in->seq_off[s];
            m.maxlen = (int)std::max<int64_t>(m.maxlen, len);
            m.sumlen += len;
            // cost_b = sum_k L_k * (L_1 + 0.05 * sum_{j<k} L_j)
            const double l1 = (double)(in->seq_off[in->blk_off[b] + 1] - in->seq_off[in->blk_off[b]]);
            if (s > in->blk_off[b]) m.cost += (double)len * (l1 + 0.05 * prev);
            prev += (double)len;
        }
        // (optimistic; the kernel re-checks -- see score_floor -- and a packed global sweep checks the cells its traceback visits)
        m.rm = row_mode(m.S, m.maxlen, m.maxlen, 2, h->h_params[in->per_block_params ? b : 0].banded == 0);
        m.fits = variant_for_len(m.maxlen, m.rm, &m.variant, m.S.sw);
        m.variant.CB = m.rm == 2 ? plane_cell_bytes(m.S) : 4;
        m.variant.DS = m.rm == 2 && m.variant.CB == 2 && p16_default_scores(m.S) && !getenv("SXG_POA_NO_DEFAULT_CLASS");
        // A11: the reference's abPOA path is banded (wb=311, wf=0.03); local alignments whose scores fit the packed
        // sweep run the one-wave banded kernel, everything else asked to be banded runs the full matrix
        // (global alignment: the adaptive band only -- the band of a row without successors holds the end column by
        //  construction, a band around backbone coordinates need not)
        const int bnd = h->h_params[in->per_block_params ? b : 0].banded;
        if (bnd && (m.S.sw || bnd == 2) && m.rm == 2 && m.maxlen <= SXG_POA_MAX_SEQ_LEN) {
            m.rm = 3;
            m.variant = Variant{band_strip_width(m.maxlen), 1, 64, 3};   // decree B2: strip width from the block's longest sequence
            m.variant.CB = band_cell_bytes(m.S);
            m.fits = true;
        }
    }
    // letters > 4 are read as N: clamped on the device after the copy (clamp_bases_kernel; a host scan of the 320 MB of the
    // headline batch was 10 ms of every upload, the unconditional staging copy before it 0.15 s)
    laps.lap("upload: checks");
    const uint8_t* host_bases = in->bases;
    int rc;
    if ((rc = h->d_blk_off.ensure(4 * (size_t)(nb + 1)))) return rc;
    if ((rc = h->d_seq_off.ensure(8 * (size_t)(ns + 1)))) return rc;
    if ((rc = h->d_bases.ensure((size_t)nbases + 16))) return rc;
    if ((rc = h->d_params.ensure(sizeof(sxg_poa_params) * (size_t)std::max(np, 1)))) return rc;
    HIPCHK(hipMemcpyAsync(h->d_blk_off.p, in->blk_off, 4 * (size_t)(nb + 1), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_seq_off.p, in->seq_off, 8 * (size_t)(ns + 1), hipMemcpyHostToDevice, h->stream));
    if (nbases) HIPCHK(hipMemcpyAsync(h->d_bases.p, host_bases, (size_t)nbases, hipMemcpyHostToDevice, h->stream));
    if (nbases) {
        hipLaunchKernelGGL(clamp_bases_kernel, dim3((unsigned)std::min<int64_t>((nbases / 16 + 255) / 256 + 1, 16384)), dim3(256), 0, h->stream,
                           h->d_bases.as<uint8_t>(), nbases);
        HIPCHK(hipGetLastError());
    }
    if (nb) HIPCHK(hipMemcpyAsync(h->d_params.p, in->params, sizeof(sxg_poa_params) * (size_t)np, hipMemcpyHostToDevice, h->stream));
    h->has_weights = in->weights != nullptr;
    if (h->has_weights) {
        if ((rc = h->d_weights.ensure(4 * (size_t)std::max<int64_t>(ns, 1)))) return rc;
        if (ns) HIPCHK(hipMemcpyAsync(h->d_weights.p, in->weights, 4 * (size_t)ns, hipMemcpyHostToDevice, h->stream));
    }
    if (h->want_block_graph) {
        std::vector<int32_t> trim((size_t)std::max(nb, 1), 0);
        for (int b = 0; b < nb; ++b) {
            trim[b] = in->bg_trim ? in->bg_trim[b] : 0;
            if (trim[b] < 0) return fail(SXG_E_INVALID, "bg_trim must not be negative");
        }
        if ((rc = h->d_trim.ensure(4 * trim.size()))) return rc;
        HIPCHK(hipMemcpy(h->d_trim.p, trim.data(), 4 * trim.size(), hipMemcpyHostToDevice));
    }
    const size_t NB = (size_t)std::max<int64_t>(nbases, 1), NS = (size_t)std::max<int64_t>(ns, 1), NBL = (size_t)std::max(nb, 1);
    if ((rc = h->d_status.ensure(4 * NBL)) || (rc = h->d_nn.ensure(4 * NBL)) || (rc = h->d_ne.ensure(4 * NBL)) ||
        (rc = h->d_nc.ensure(4 * NBL)) || (rc = h->d_node_code.ensure(NB)) || (rc = h->d_node_rank.ensure(4 * NB)) ||
        (rc = h->d_node_group.ensure(4 * NB)) || (rc = h->d_edge_tail.ensure(4 * NB)) ||
        (rc = h->d_edge_head.ensure(4 * NB)) || (rc = h->d_edge_w.ensure(4 * NB)) || (rc = h->d_paths.ensure(4 * NB)) ||
        (rc = h->d_score.ensure(4 * NS)) || (rc = h->d_cells.ensure(8 * NS)) || (rc = h->d_queue.ensure(256)) ||
        (rc = h->d_work.ensure(4 * NBL)) || (rc = h->d_blk_cycles.ensure(8 * NBL)))
        return rc;
    if (h->want_consensus && (rc = h->d_cons.ensure(4 * NB))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    laps.lap("upload: copies");
    h->have_batch = true;
    h->sh_dealt = false; h->sh_exchanged = false;   // (a plain upload: no deal the root could reassemble by)
    return SXG_OK;
}

struct LaunchPlan {
    Variant variant; bool cvx, sw; int tier = 0; bool wide_band = false;
    std::vector<int32_t> work;  // block ids, largest cost first
    std::vector<unsigned long long> est;  // cost-model cells per work item (priority balancing)
    // filled by prepare_plan
    SlotLayout lay; KernelFn<BlockArgs> kern = nullptr; int per_cu = 1; int64_t want_slots = 0, n_slots = 0;
    int clock_mhz = 0;   // shader clock this launch ran at, sampled from its slots (see sample_clock)
    int smem = 0, pf_off = -1; bool park_lds = true; uint64_t cells = 0, bytes = 0; float ms = 0;
};

// strips per plane row of the packed sweep: ~1100 columns around the backbone hint (SXG_POA_BAND_COLS narrows it -- a test
// knob that makes tracebacks miss their band, so that the in-kernel hint shift and the wide-plane re-run are exercised)
static int plane_strips_p16(int T, int W) {
    if (const char* e = getenv("SXG_POA_BAND_COLS")) {
        const int cols = std::max(atoi(e), W);
        return std::min(plane_round4((cols + W - 1) / W), 2 * T);
    }
    return p16_band_strips(T, W);
}

static void prepare_plan(sxg_poa_handle* h, LaunchPlan& P, int attempt) {
    const Variant V = P.variant;
    const int Lpad = V.Lpad();
    int nodes_cap = 0, maxlen = 0;
    bool any_adaptive = false;   // banded blocks with the adaptive band (B4): a row's band may span the whole 128-strip window
    bool any_spoa = false;       // blocks that ask for spoa's depth-first order: the only ones whose arena carries its scratch
    double rows_est = 0;  // graph rows a block is expected to reach: every further sequence adds ~1.5 % of its
                          // length in new nodes on pangenome-like input (measured on the synthetic blocks: 1.43 %)
    for (int b : P.work) {
        const BlockMeta& m = h->meta[b];
        any_adaptive = any_adaptive || h->h_params[h->per_block_params ? b : 0].banded == 2;
        any_spoa = any_spoa || (h->h_params[h->per_block_params ? b : 0].mode & SXG_ORDER_SPOA) != 0;
        nodes_cap = (int)std::max<int64_t>(nodes_cap, m.sumlen);
        maxlen = std::max(maxlen, m.maxlen);
        rows_est = std::max(rows_est, (double)m.maxlen * std::max(2.0, 1.0 + 0.0165 * m.nseq));
    }
    nodes_cap += 8;
    // The graph arrays (~100 B per node and edge) used to be sized for the worst case -- one node per base of the
    // block, 34 MB of a 64 x 5 kbp slot whose graph ends at ~10 k nodes and ~14 k edges.  The first two tiers size
    // them from the row estimate (edges and nodes each stay below twice the rows); a block that outgrows them
    // answers NODES_OVERFLOW and moves up a tier like any other capacity overflow.
    const int nodes_full = nodes_cap;
    int rows_cap, pool_slots, step_cap;
    if (attempt == 0) {
        rows_cap = (int)std::min<double>(nodes_cap, rows_est + 1024);
        pool_slots = std::min(rows_cap + 1, 768);
        step_cap = rows_cap;
        nodes_cap = (int)std::min<int64_t>(nodes_full, 2LL * rows_cap + maxlen + 8);
    } else if (attempt == 1) {
        rows_cap = (int)std::min<double>(nodes_cap, 2.5 * rows_est + 4096);
        pool_slots = std::min(rows_cap + 1, 4096);
        step_cap = 3 * rows_cap;
        nodes_cap = (int)std::min<int64_t>(nodes_full, 2LL * rows_cap + maxlen + 8);
    } else if (attempt == 2) {
        rows_cap = nodes_cap; pool_slots = std::min(rows_cap + 1, 32768); step_cap = nodes_cap;  // edges <= nodes_cap
    } else {
        rows_cap = nodes_cap; pool_slots = rows_cap + 1; step_cap = nodes_cap;                    // worst case
    }
    if (rows_cap >= (1 << 20)) rows_cap = (1 << 20) - 1;
    const int wb = V.RM == 1 ? 8 : 4;
    if (V.RM == 3) pool_slots = 1;   // (the banded sweep has no row ring: predecessors come from the plane)
    P.lay = make_layout(nodes_cap, rows_cap, pool_slots, step_cap, V.T(), Lpad, wb, false,
                        V.RM == 3 ? (any_adaptive ? BAND_WIN : band_plane_strips(maxlen, V.W)) : (V.RM == 2 ? (P.wide_band ? 2 * V.T() : plane_strips_p16(V.T(), V.W)) : 0),
                        V.RM >= 2 ? V.CB : 4, any_spoa);
    // (the one-wave 2-byte classes up to W = 11 are compiled for a plane that keeps every strip -- dp_fill_p16's RP = 2; a launch whose
    //  plane was narrowed, SXG_POA_BAND_COLS, runs the two-wave class at 64 threads instead)
    if (V.RM == 2 && V.CB == 2 && V.TMAX == 64 && V.W <= 11 && P.lay.band_strips != 2 * V.T()) P.variant.TMAX = 128;
    P.kern = block_kernel(P.variant, P.cvx, P.sw);
    P.smem = dp_lds_launch_bytes(Lpad, wb);
    P.park_lds = dp_park_in_lds(Lpad, wb);
    if (V.RM == 2) { P.lay.lds_rows = p16_lds_rows(V.T(), V.W); P.smem = dp16_lds_bytes(V.T(), V.W, P.lay.lds_rows); P.park_lds = true; }
    if (V.RM == 3) { P.smem = band_lds_bytes(V.W); P.park_lds = true; }
    P.pf_off = (V.RM < 2 && getenv("SXG_POA_PREFETCH")) ? dp_pf_offset(Lpad, wb, V.T()) : -1;
    if (P.pf_off >= 0) P.smem += dp_pf_bytes(Lpad, wb, V.T());
    if (P.smem > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)P.kern, hipFuncAttributeMaxDynamicSharedMemorySize, P.smem);
    P.per_cu = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&P.per_cu, (const void*)P.kern, V.T(), (size_t)P.smem) != hipSuccess || P.per_cu < 1)
        P.per_cu = 1;
    P.want_slots = std::min<int64_t>((int64_t)P.work.size(), (int64_t)h->num_cu * P.per_cu);
}

// The shader clock a launch ran at: every slot reads the core-clock counter (s_memtime) and the constant 100 MHz counter
// (s_memrealtime) when it starts and when it ends; cycles / ticks * 100 MHz, over a sample of slots.  bench.py
// prices the VALU roof with it -- the boxes of the pool sustain different clocks under this kernel.
static int sample_clock(const LaunchPlan& P, const PlanRes& R) {
    unsigned long long cyc = 0, ticks = 0;
    const int64_t step = std::max<int64_t>(1, P.n_slots / 24);
    for (int64_t sl = 0; sl < P.n_slots; sl += step) {
        unsigned long long v[10];
        if (hipMemcpy(v, R.arena.as<uint8_t>() + (size_t)sl * P.lay.total + P.lay.hdr + 64, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return 0;
        if (v[9] <= v[8] || v[7] <= v[6]) continue;
        cyc += v[7] - v[6];
        ticks += v[9] - v[8];
    }
    return ticks ? (int)((double)cyc / (double)ticks * 100.0 + 0.5) : 0;
}

static int launch_plan(sxg_poa_handle* h, LaunchPlan& P, PlanRes& R, const int prio_base) {
    const Variant V = P.variant;
    int rc;
    if (!P.kern) return fail(SXG_E_INVALID, "no kernel class built for this geometry");
    if ((rc = R.arena.ensure((size_t)P.n_slots * P.lay.total))) return rc;
    // Workgroup k runs on CU k mod #CU (measured, profiles/tools/slot_report.py): when the slots do not
    // divide evenly, the CUs with one workgroup less run theirs faster -- they get the costliest blocks, so
    // that those are not what the launch ends on.  P.work arrives sorted by cost, largest first.
    {
        const int64_t ns = P.n_slots, ncu = std::max(h->num_cu, 1);
        const int64_t rem = ns % ncu, full = ns / ncu;
        if (rem > 0 && full >= 1 && (int64_t)P.work.size() >= ns) {
            std::vector<int32_t> first((size_t)ns, -1);
            size_t next = 0;
            for (int64_t p = 0; p < ns; ++p)          // light CUs first
                if (p % ncu >= rem && p / ncu < full) first[(size_t)p] = P.work[next++];
            // the crowded CUs are dealt the rest in snake order (round 0 left to right, round 1 right to
            // left, ...), which evens out the sum of costs per CU
            for (int64_t round = 0; round <= full; ++round)
                for (int64_t c = 0; c < rem; ++c) {
                    const int64_t cu = (round & 1) ? rem - 1 - c : c, p = round * ncu + cu;
                    if (p < ns && first[(size_t)p] < 0) first[(size_t)p] = P.work[next++];
                }
            std::copy(first.begin(), first.end(), P.work.begin());
        }
    }
    if ((rc = R.work.ensure(4 * P.work.size())) || (rc = R.queue.ensure(256)) || (rc = R.est.ensure(8 * P.work.size()))) return rc;
    HIPCHK(hipMemcpyAsync(R.work.p, P.work.data(), 4 * P.work.size(), hipMemcpyHostToDevice, R.stream));
    P.est.resize(P.work.size());
    for (size_t k = 0; k < P.work.size(); ++k) P.est[k] = (unsigned long long)std::max(h->meta[P.work[k]].cost, 1.0);
    HIPCHK(hipMemcpyAsync(R.est.p, P.est.data(), 8 * P.est.size(), hipMemcpyHostToDevice, R.stream));
    HIPCHK(hipMemsetAsync(R.queue.p, 0, 256, R.stream));
    BlockArgs A;
    A.blk_off = h->d_blk_off.as<int32_t>(); A.seq_off = h->d_seq_off.as<int64_t>(); A.bases = h->d_bases.as<uint8_t>();
    A.weights = h->has_weights ? h->d_weights.as<uint32_t>() : nullptr;
    A.params = h->d_params.as<sxg_poa_params>(); A.per_block_params = h->per_block_params;
    A.work = R.work.as<int32_t>(); A.n_work = (int)P.work.size(); A.queue = R.queue.as<int32_t>();
    A.arena = R.arena.as<uint8_t>(); A.lay = P.lay;
    A.status = h->d_status.as<int32_t>(); A.n_nodes = h->d_nn.as<int32_t>(); A.n_edges = h->d_ne.as<int32_t>();
    A.n_cons = h->d_nc.as<int32_t>(); A.node_code = h->d_node_code.as<uint8_t>();
    A.node_rank = h->d_node_rank.as<int32_t>(); A.node_group = h->d_node_group.as<int32_t>();
    A.edge_tail = h->d_edge_tail.as<int32_t>(); A.edge_head = h->d_edge_head.as<int32_t>();
    A.edge_weight = h->d_edge_w.as<uint32_t>(); A.paths = h->d_paths.as<int32_t>(); A.score = h->d_score.as<int32_t>();
    A.cells = h->d_cells.as<unsigned long long>(); A.cons_nodes = h->d_cons.as<int32_t>();
    A.want_consensus = h->want_consensus;
    A.park_in_lds = P.park_lds ? 1 : 0;
    A.pf_off = P.pf_off;
    A.num_cu = std::max(h->num_cu, 1);
    // One board for all launches of the round: workgroups of different geometries share CUs, and each takes as priority
    // the number of co-residents -- of ANY launch -- with less work left.  (Round 3 kept a board per launch: two launches
    // side by side were balanced pair by pair and lost more than their narrower geometry saved.)
    A.prio_board = getenv("SXG_POA_NO_BALANCE") ? nullptr : h->d_board.as<uint32_t>();
    A.prio_base = prio_base;
    A.est = R.est.as<unsigned long long>();
    A.blk_cycles = h->d_blk_cycles.as<unsigned long long>();
    A.lds_bytes = getenv("SXG_POA_RESORT_NO_LDS") ? 0 : P.smem;   // (test knob: the S7' re-sort keeps its states in the slot's scratch, as it does for graphs beyond its LDS)
    // every slot's header (counters, phase times, clock readings) starts a launch at zero: one strided memset
    HIPCHK(hipMemset2DAsync(R.arena.as<uint8_t>() + P.lay.hdr, P.lay.total, 0, 512, (size_t)P.n_slots, R.stream));
    HIPCHK(hipStreamWaitEvent(R.stream, h->ev0, 0));
    HIPCHK(hipEventRecord(R.e0, R.stream));
    hipLaunchKernelGGL(P.kern, dim3((unsigned)P.n_slots), dim3(V.T()), (size_t)P.smem, R.stream, A);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(R.e1, R.stream));
    HIPCHK(hipStreamWaitEvent(h->stream, R.e1, 0));
    return SXG_OK;
}

static int debug_plan(sxg_poa_handle* h, LaunchPlan& P, PlanRes& R, int attempt) {
    const Variant V = P.variant;
    unsigned long long acc[8] = {0}, smin = ~0ull, smax = 0;
    for (int64_t sl = 0; sl < P.n_slots; ++sl) {
        unsigned long long one[8];
        HIPCHK(hipMemcpy(one, R.arena.as<uint8_t>() + (size_t)sl * P.lay.total + P.lay.hdr + 64, 64, hipMemcpyDeviceToHost));
        for (int k = 0; k < 8; ++k) acc[k] += one[k];
        unsigned long long st = 0;
        for (int k = 0; k < 6; ++k) st += one[k];
        smin = std::min(smin, st); smax = std::max(smax, st);
    }
    if (const char* path = getenv("SXG_POA_SLOT_CSV")) {  // per-slot placement and wall-clock span
        if (FILE* f = fopen(path, "a")) {
            for (int64_t sl = 0; sl < P.n_slots; ++sl) {
                unsigned long long v[26];
                HIPCHK(hipMemcpy(v, R.arena.as<uint8_t>() + (size_t)sl * P.lay.total + P.lay.hdr + 64, sizeof(v), hipMemcpyDeviceToHost));
                fprintf(f, "%lld,%llu,%llu", (long long)sl, v[8], v[9]);
                for (int w = 0; w < V.NW && w < 16; ++w) fprintf(f, ",%llx", v[10 + w]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
#ifdef SXG_ROW_PROF
    {
        unsigned long long ra[13] = {0};
        for (int64_t sl = 0; sl < P.n_slots; ++sl) {
            unsigned long long one[13];
            HIPCHK(hipMemcpy(one, R.arena.as<uint8_t>() + (size_t)sl * P.lay.total + P.lay.hdr + 64 + 28 * 8, sizeof(one), hipMemcpyDeviceToHost));
            for (int k = 0; k < 13; ++k) ra[k] += one[k];
        }
        double rt = 1e-9;
        for (int k = 0; k < 8; ++k) rt += (double)ra[k];
        static const char* seg16[8] = {"setup", "pass1+scan", "wait left wave", "combine+pass2", "wait right wave", "hand-over+end cell", "stored row fetch", "outgoing+row store"};
        static const char* segb[8] = {"descriptor+band", "predecessor fetch+fold", "pass1+scan+pass2", "end cell+best-cell search", "outgoing+band stores", "-", "-", "-"};
        const char* const* seg = V.RM == 3 ? segb : seg16;
        fprintf(stderr, "[sxg]   row profile:");
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.1f%%", seg[k], 100.0 * (double)ra[k] / rt);
        fprintf(stderr, "\n");
        fprintf(stderr, "[sxg]   traceback: %.3g steps, %.1f steps per window, %.0f %% of the windows prefetched, %.0f cycles per step of which %.0f in window fills\n",
                (double)ra[8], (double)ra[8] / std::max<double>((double)ra[9], 1), 100.0 * (double)ra[12] / std::max<double>((double)ra[9], 1),
                (double)ra[10] / std::max<double>((double)ra[8], 1), (double)ra[11] / std::max<double>((double)ra[8], 1));
    }
#endif
#ifdef SXG_RESORT_PROF
    {   // phases of the S7' re-sort (poa_graph_dev.h::spoa_resort_par), thread 0's clock
        unsigned long long ra[6] = {0};
        for (int64_t sl = 0; sl < P.n_slots; ++sl) {
            unsigned long long one[6];
            HIPCHK(hipMemcpy(one, R.arena.as<uint8_t>() + (size_t)sl * P.lay.total + P.lay.hdr + 64 + 41 * 8, sizeof(one), hipMemcpyDeviceToHost));
            for (int k = 0; k < 6; ++k) ra[k] += one[k];
        }
        double rt = 1e-9;
        for (int k = 0; k < 5; ++k) rt += (double)ra[k];
        fprintf(stderr, "[sxg]   re-sort: records %.1f%% first() %.1f%% counts %.1f%% walks %.1f%% ranks %.1f%%; %.3g cycles per slot, %.1f rounds per slot\n",
                100 * ra[0] / rt, 100 * ra[1] / rt, 100 * ra[2] / rt, 100 * ra[3] / rt, 100 * ra[4] / rt, rt / (double)P.n_slots, (double)ra[5] / (double)P.n_slots);
    }
#endif
    double tot = 1e-9;
    for (int k = 0; k < 6; ++k) tot += (double)acc[k];
    size_t fr = 0, tt = 0;
    (void)hipMemGetInfo(&fr, &tt);
    fprintf(stderr, "[sxg] variant T=%d W=%d cvx=%d rowmode=%d attempt=%d work=%zu slots=%lld per_cu=%d smem=%d slot_bytes=%zu free=%zu ms=%.2f\n",
            V.T(), V.W, (int)P.cvx, V.RM, attempt, P.work.size(), (long long)P.n_slots, P.per_cu, P.smem, P.lay.total, fr, P.ms);
    fprintf(stderr, "[sxg]   slot busy (of the longest slot): min %.1f%% mean %.1f%%\n", 100.0 * (double)smin / (double)std::max(smax, 1ull),
            100.0 * tot / (double)P.n_slots / (double)std::max(smax, 1ull));
    fprintf(stderr, "[sxg]   slot time: other %.1f%% prep_rows %.1f%% dp_fill %.1f%% traceback %.1f%% add_alignment %.1f%% output %.1f%%\n",
            100 * acc[0] / tot, 100 * acc[1] / tot, 100 * acc[2] / tot, 100 * acc[3] / tot, 100 * acc[4] / tot, 100 * acc[5] / tot);
    return SXG_OK;
}

// A9 + A10 of every block on the device (block_graph_kernel), after the batch's alignments: outputs in per-block
// layouts (nodes: prefix of the POA node counts, edges: prefix of POA edges + consensus nodes, steps: the input layout).
static int run_block_graphs(sxg_poa_handle* h, std::vector<int32_t>& status) {
    const int nb = h->n_blocks;
    h->bg_done = false;
    std::vector<int32_t> nn(std::max(nb, 1), 0), ne(std::max(nb, 1), 0), nc(std::max(nb, 1), 0);
    if (nb) {
        HIPCHK(hipMemcpy(nn.data(), h->d_nn.p, 4 * (size_t)nb, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ne.data(), h->d_ne.p, 4 * (size_t)nb, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(nc.data(), h->d_nc.p, 4 * (size_t)nb, hipMemcpyDeviceToHost));
    }
    h->bg_node_o.assign((size_t)nb + 1, 0); h->bg_edge_o.assign((size_t)nb + 1, 0);
    int capV = 1, capE = 1, capC = 1, capT = 1;
    std::vector<int32_t> work;
    for (int b = 0; b < nb; ++b) {
        const bool ok = status[b] == ST_OK && h->meta[b].nseq > 0;
        const int cn = h->want_consensus ? nc[b] : 0;
        h->bg_node_o[(size_t)b + 1] = h->bg_node_o[(size_t)b] + (ok ? nn[b] : 0);
        h->bg_edge_o[(size_t)b + 1] = h->bg_edge_o[(size_t)b] + (ok ? ne[b] + cn : 0);
        if (!ok) continue;
        work.push_back(b);
        capV = std::max(capV, nn[b]); capE = std::max(capE, ne[b]); capC = std::max(capC, cn);
        capT = std::max(capT, std::max(std::max(nn[b], h->meta[b].maxlen), cn));
    }
    std::stable_sort(work.begin(), work.end(), [&](int a, int b) { return h->meta[a].sumlen > h->meta[b].sumlen; });
    const size_t NT = (size_t)std::max<int64_t>(h->bg_node_o[(size_t)nb], 1), ET = (size_t)std::max<int64_t>(h->bg_edge_o[(size_t)nb], 1);
    const size_t NB = (size_t)std::max<int64_t>(h->n_bases, 1), NS = (size_t)std::max<int64_t>(h->n_seqs, 1), NBL = (size_t)std::max(nb, 1);
    int rc;
    if ((rc = h->d_bg_no.ensure(8 * (NBL + 1))) || (rc = h->d_bg_eo.ensure(8 * (NBL + 1))) || (rc = h->d_bg_len.ensure(4 * NT)) ||
        (rc = h->d_bg_od.ensure(4 * NT)) || (rc = h->d_bg_id.ensure(NT)) || (rc = h->d_bg_seq.ensure(NT)) || (rc = h->d_bg_eto.ensure(4 * ET)) ||
        (rc = h->d_bg_steps.ensure(4 * NB)) || (rc = h->d_bg_nsteps.ensure(4 * NS)) || (rc = h->d_bg_cons.ensure(4 * NT)) ||
        (rc = h->d_bg_counts.ensure(4 * (size_t)BGC_N * NBL)) || (rc = h->d_bg_work.ensure(4 * std::max<size_t>(work.size(), 1))) ||
        (rc = h->d_bg_queue.ensure(256)))
        return rc;
    HIPCHK(hipMemcpyAsync(h->d_bg_no.p, h->bg_node_o.data(), 8 * ((size_t)nb + 1), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_bg_eo.p, h->bg_edge_o.data(), 8 * ((size_t)nb + 1), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(h->d_bg_counts.p, 0, 4 * (size_t)BGC_N * NBL, h->stream));
    HIPCHK(hipMemsetAsync(h->d_bg_nsteps.p, 0, 4 * NS, h->stream));
    HIPCHK(hipMemsetAsync(h->d_bg_queue.p, 0, 256, h->stream));
    if (!work.empty()) {
        HIPCHK(hipMemcpyAsync(h->d_bg_work.p, work.data(), 4 * work.size(), hipMemcpyHostToDevice, h->stream));
        const size_t slot = bg_slot_bytes(capV, capE, capC, capT);
        int per_cu = 1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)block_graph_kernel, BG_THREADS, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        int64_t slots = std::min<int64_t>((int64_t)work.size(), (int64_t)h->num_cu * per_cu);
        slots = std::max<int64_t>(1, std::min<int64_t>(slots, (int64_t)(arena_budget(h) / slot)));
        if ((rc = h->d_bg_arena.ensure((size_t)slots * slot))) return rc;
        BgArgs A;
        A.blk_off = h->d_blk_off.as<int32_t>(); A.seq_off = h->d_seq_off.as<int64_t>(); A.bases = h->d_bases.as<uint8_t>();
        A.paths = h->d_paths.as<int32_t>(); A.node_code = h->d_node_code.as<uint8_t>(); A.edge_tail = h->d_edge_tail.as<int32_t>();
        A.edge_head = h->d_edge_head.as<int32_t>(); A.cons = h->want_consensus ? h->d_cons.as<int32_t>() : nullptr;
        A.nn = h->d_nn.as<int32_t>(); A.ne = h->d_ne.as<int32_t>(); A.nc = h->d_nc.as<int32_t>(); A.trim = h->d_trim.as<int32_t>();
        A.cons_mode = h->bg_cons_visited_only ? 2 : 1;
        A.work = h->d_bg_work.as<int32_t>(); A.n_work = (int)work.size(); A.queue = h->d_bg_queue.as<int32_t>();
        A.arena = h->d_bg_arena.as<uint8_t>(); A.slot_bytes = slot; A.capV = capV; A.capE = capE; A.capC = capC; A.capT = capT;
        A.node_o = h->d_bg_no.as<int64_t>(); A.edge_o = h->d_bg_eo.as<int64_t>();
        A.o_len = h->d_bg_len.as<int32_t>(); A.o_outdeg = h->d_bg_od.as<int32_t>(); A.o_indeg = h->d_bg_id.as<uint8_t>();
        A.o_seq = h->d_bg_seq.as<char>(); A.o_eto = h->d_bg_eto.as<int32_t>(); A.o_steps = h->d_bg_steps.as<int32_t>();
        A.o_nsteps = h->d_bg_nsteps.as<int32_t>(); A.o_cons = h->d_bg_cons.as<int32_t>(); A.o_counts = h->d_bg_counts.as<int32_t>();
        HIPCHK(hipEventRecord(h->ev0, h->stream));
        hipLaunchKernelGGL(block_graph_kernel, dim3((unsigned)slots), dim3(BG_THREADS), 0, h->stream, A);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(h->ev1, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (!work.empty()) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1)); h->stats.bg_ms = ms; }
    h->bg_counts.assign((size_t)BGC_N * NBL, 0);
    if (nb) HIPCHK(hipMemcpy(h->bg_counts.data(), h->d_bg_counts.p, 4 * (size_t)BGC_N * (size_t)nb, hipMemcpyDeviceToHost));
    bool changed = false;
    for (int b : work)
        if (h->bg_counts[(size_t)BGC_N * b + BGC_STATUS] != ST_OK) { status[b] = ST_INTERNAL; changed = true; }
    if (changed) HIPCHK(hipMemcpy(h->d_status.p, status.data(), 4 * (size_t)nb, hipMemcpyHostToDevice));
    h->bg_done = true;
    return SXG_OK;
}

extern "C" int sxg_poa_batch_execute(sxg_poa_handle* h) {
    if (!h) return fail(SXG_E_INVALID, "handle is NULL");
    if (!h->have_batch) return fail(SXG_E_INVALID, "no batch uploaded");
    HIPCHK(hipSetDevice(h->device));
    const RoctxRange range_("sxg_poa_batch_execute");
    h->stats = sxg_poa_stats{};
    const bool dbg = getenv("SXG_POA_DEBUG") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        auto T1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[sxg] host %-18s %.3f ms\n", what, std::chrono::duration<double, std::milli>(T1 - T0).count());
        T0 = T1;
    };
    const int nb = h->n_blocks;
    std::vector<int32_t> status(std::max(nb, 1), 0);
    float ms_total = 0;
    // blocks whose longest sequence exceeds the largest variant fail up front
    std::vector<int32_t> pending;
    for (int b = 0; b < nb; ++b) {
        if (!h->meta[b].fits) status[b] = ST_TOO_LONG; else pending.push_back(b);
    }
    if (nb) HIPCHK(hipMemcpyAsync(h->d_status.p, status.data(), 4 * (size_t)nb, hipMemcpyHostToDevice, h->stream));
    if (nb) {
        HIPCHK(hipMemsetAsync(h->d_nn.p, 0, 4 * (size_t)nb, h->stream));
        HIPCHK(hipMemsetAsync(h->d_ne.p, 0, 4 * (size_t)nb, h->stream));
        HIPCHK(hipMemsetAsync(h->d_nc.p, 0, 4 * (size_t)nb, h->stream));
        HIPCHK(hipMemsetAsync(h->d_blk_cycles.p, 0, 8 * (size_t)nb, h->stream));
    }
    lap("setup");
    std::vector<LaunchPlan> all_plans;
    std::vector<int32_t> nomem_blocks;  // failed for lack of arena memory: their status is final
    for (int b = 0; b < nb; ++b) h->meta[b].tier = 0;
    // Rounds.  A block comes back for two independent reasons, each with its own ladder: its arena was too
    // small (ROWS / POOL / TBX overflow: capacity tier 0..3, the last one is the worst case) or its sweep was
    // too narrow (RANGE overflow / BAND miss: packed -> int16 row words -> int32 row words, at the SAME
    // capacity tier; a band miss first repeats the packed sweep with a plane that keeps every strip).  3 + 3 steps at most, so 7
    // rounds always suffice; neither internal status is ever final.
    for (int attempt = 0; attempt < 7 && !pending.empty(); ++attempt) {
        // group by (variant, convex, local, capacity tier): one launch per group, all groups concurrently
        std::vector<LaunchPlan> plans;
        for (int b : pending) {
            const BlockMeta& m = h->meta[b];
            LaunchPlan* pl = nullptr;
            for (auto& q : plans)
                if (q.variant.W == m.variant.W && q.variant.NW == m.variant.NW && q.variant.RM == m.variant.RM && q.variant.CB == m.variant.CB && q.variant.DS == m.variant.DS && q.cvx == m.cvx && q.sw == m.sw && q.tier == m.tier && q.wide_band == m.wide_band) { pl = &q; break; }
            if (!pl) { plans.emplace_back(); pl = &plans.back(); pl->variant = m.variant; pl->cvx = m.cvx; pl->sw = m.sw; pl->tier = m.tier; pl->wide_band = m.wide_band; }
            pl->work.push_back(b);
        }
        std::sort(plans.begin(), plans.end(), [](const LaunchPlan& a, const LaunchPlan& b) { return a.variant.Lpad() > b.variant.Lpad(); });
        // a geometry with at least 75 % of the columns of a wider one of the same kind joins it: fewer,
        // fuller launches beat many partial ones, but every joined block sweeps the wider geometry's columns
        // (measured on the mixed batch, round 2 with over-subscribed launches: 0.60 39.6 s, 0.75 33.8 s, 0.90 34.3 s)
        // Round 4, with ONE priority board for all launches of a round: the headline's two geometries (5 120 and 5 632 columns)
        // run 2.4 % faster apart than merged (2 137 -> 2 088 ms, same box) and the mixed batch 1.8 %; the small geometries of
        // config 2 (three launches of 1-2 waves per block) still lose 15 % apart.  So: wide geometries merge only within 8 %,
        // the others within 25 % as before.
        const double merge_env = getenv("SXG_POA_MERGE") ? atof(getenv("SXG_POA_MERGE")) : 0.0;
        for (size_t i = 0; i < plans.size(); ++i)
            for (size_t j = i + 1; j < plans.size();) {
                const LaunchPlan &a = plans[i], &b = plans[j];
                if (a.variant.RM == b.variant.RM && a.variant.CB == b.variant.CB && a.variant.DS == b.variant.DS && a.variant.RM != 3 && a.cvx == b.cvx && a.sw == b.sw && a.tier == b.tier && a.wide_band == b.wide_band &&
                    (double)b.variant.Lpad() >= (merge_env > 0 ? merge_env : (a.variant.Lpad() >= 4096 ? 0.92 : 0.75)) * (double)a.variant.Lpad()) {
                    plans[i].work.insert(plans[i].work.end(), b.work.begin(), b.work.end());
                    plans.erase(plans.begin() + (long)j);
                } else ++j;
            }
        // A batch that cannot fill the device with one workgroup per block (1000 blocks of 1 kbp are 1000 waves of 4096)
        // gives every block twice the waves at half the strip width: the same columns, shorter rows.
        if (!getenv("SXG_POA_NO_SPREAD")) {
            uint64_t waves = 0;
            for (auto& pl : plans) waves += (uint64_t)pl.work.size() * (uint64_t)pl.variant.NW;
            for (auto& pl : plans) {
                Variant& v = pl.variant;
                while (v.RM == 2 && v.NW <= 2 && v.W % 2 == 0 && v.W / 2 >= 4 && 2 * waves <= (uint64_t)h->num_cu * 16u) {
                    waves += (uint64_t)pl.work.size() * (uint64_t)v.NW;
                    v = Variant{v.W / 2, 2 * v.NW, 64 * 2 * v.NW, 2, v.CB, v.DS};   // (NW 1 -> 2: the two-wave class; 2 -> 4: the four-wave class)
                }
            }
        }
        const uint64_t budget = arena_budget(h);
        uint64_t want_bytes = 0;
        for (auto& pl : plans) {
            std::stable_sort(pl.work.begin(), pl.work.end(), [&](int a, int b) { return h->meta[a].cost > h->meta[b].cost; });
            prepare_plan(h, pl, pl.tier);
            want_bytes += (uint64_t)pl.want_slots * pl.lay.total;
        }
        // A geometry whose single arena does not fit the budget fails ITS blocks (per-block status, as the
        // ABI promises) and the rest of the batch goes on.
        for (size_t i = 0; i < plans.size();) {
            if ((uint64_t)plans[i].lay.total > budget && plans[i].wide_band) {
                // the every-strip plane (rows x columns dwords) is what does not fit: these blocks are not final -- their
                // status stays ST_BAND_MISS and the round's bookkeeping below sends them down the widening ladder
                // (packed -> 32-bit sweep, one byte per cell), which is where a band miss went before the wide plane existed
                for (int b : plans[i].work) { h->meta[b].wide_band = false; h->meta[b].no_wide = true; }
                plans.erase(plans.begin() + (long)i);
            } else if ((uint64_t)plans[i].lay.total > budget) {
                for (int b : plans[i].work) status[b] = plans[i].tier >= 2 ? ST_POOL_OVERFLOW : ST_ROWS_OVERFLOW;
                g_err = "memory budget too small for a block arena of " + std::to_string(plans[i].lay.total) + " bytes";
                nomem_blocks.insert(nomem_blocks.end(), plans[i].work.begin(), plans[i].work.end());
                plans.erase(plans.begin() + (long)i);
            } else ++i;
        }
        // Slots per launch.  One geometry: as many as there are blocks / as fit.  Several geometries
        // running side by side should end together, so their WAVES are made proportional to their
        // work (cost model of SURVEY 8e): slots_p = lambda * cost_p / waves_per_slot_p, with the largest
        // lambda that respects the arena budget and the wave capacity of the device.
        // (a block swept by a wider geometry than its own -- merged launches -- costs the columns of THAT geometry)
        std::vector<double> pcost(plans.size(), 0.0);
        for (size_t i = 0; i < plans.size(); ++i)
            for (int b : plans[i].work)
                pcost[i] += std::max(h->meta[b].cost, 1.0) * (plans[i].variant.RM == 3 ? 1.0 : (double)plans[i].variant.Lpad() / (double)std::max(h->meta[b].maxlen, 1));
        const int oversub = getenv("SXG_POA_OVERSUB") ? std::max(1, atoi(getenv("SXG_POA_OVERSUB"))) : 2;
        auto slots_at = [&](size_t i, double lambda) {
            const double s = lambda * pcost[i] / (double)plans[i].variant.NW;
            return (int64_t)std::min<double>((double)plans[i].want_slots, std::max(1.0, std::floor(s)));
        };
        auto fits = [&](double lambda) {
            uint64_t bytes = 0, waves = 0;
            for (size_t i = 0; i < plans.size(); ++i) {
                const int64_t n = slots_at(i, lambda);
                bytes += (uint64_t)n * plans[i].lay.total;
                waves += (uint64_t)n * (uint64_t)plans[i].variant.NW;
            }
            // Side by side the launches may ask for up to TWICE the wave slots of the device: workgroups that find no room wait
            // in their hardware queue and start -- pulling from their launch's block queue -- as soon as another launch
            // drains, so a launch the cost model under-estimated is not left alone on a half-empty chip at the end.
            return bytes <= budget && (plans.size() == 1 || waves <= (uint64_t)h->num_cu * 16u * (uint64_t)oversub);
        };
        double lam_lo = 0.0, lam_hi = 1.0;
        while (lam_hi < 1e30 && fits(lam_hi)) {
            bool all_full = true;
            for (size_t i = 0; i < plans.size(); ++i) all_full = all_full && slots_at(i, lam_hi) >= plans[i].want_slots;
            if (all_full) break;
            lam_hi *= 2.0;
        }
        if (!fits(lam_hi))
            for (int it = 0; it < 100; ++it) {
                const double mid = 0.5 * (lam_lo + lam_hi);
                if (fits(mid)) lam_lo = mid; else lam_hi = mid;
            }
        const double lambda = fits(lam_hi) ? lam_hi : lam_lo;
        (void)want_bytes;
        while (h->planres.size() < plans.size()) {
            PlanRes* r = new PlanRes();
            HIPCHK(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreate(&r->e0));
            HIPCHK(hipEventCreate(&r->e1));
            h->planres.push_back(r);
        }
        for (size_t i = 0; i < plans.size(); ++i) plans[i].n_slots = slots_at(i, lambda);
        lap("plan");
        {
            const size_t board_bytes = (size_t)PRIO_BOARD_CUS * PRIO_BOARD_SLOTS * 4;
            int rcb = h->d_board.ensure(board_bytes);
            if (rcb) return rcb;
            HIPCHK(hipMemsetAsync(h->d_board.p, 0, board_bytes, h->stream));
        }
        HIPCHK(hipEventRecord(h->ev0, h->stream));
        int prio_base = 0;
        for (size_t i = 0; i < plans.size(); ++i) {
            int rc = launch_plan(h, plans[i], *h->planres[i], prio_base);
            if (rc) return rc;
            prio_base += (int)((plans[i].n_slots + h->num_cu - 1) / std::max(h->num_cu, 1));
        }
        HIPCHK(hipEventRecord(h->ev1, h->stream));
        // While the kernels run: pin the host buffer the step lists will be downloaded into (first call of a handle, or a
        // bigger batch than before; afterwards the pool already has it).  n_bases bounds the number of steps.
        if (attempt == 0 && h->want_block_graph && h->pins && h->n_bases >= ((int64_t)16 << 20) && !getenv("SXG_POA_NO_PINNED")) {
            if (h->pin_thread.joinable()) h->pin_thread.join();
            try {
                h->pin_thread = std::thread([pool = h->pins, dev = h->device, bytes = 4 * (size_t)h->n_bases] {
                    if (hipSetDevice(dev) == hipSuccess) pool->prewarm(bytes);
                });
            } catch (...) {}   // (no thread: the download pins, or falls back to pageable memory, itself)
        }
        HIPCHK(hipStreamSynchronize(h->stream));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
        ms_total += ms;
        for (size_t i = 0; i < plans.size(); ++i) {
            HIPCHK(hipEventElapsedTime(&plans[i].ms, h->planres[i]->e0, h->planres[i]->e1));
            plans[i].clock_mhz = sample_clock(plans[i], *h->planres[i]);
            h->stats.dp_launches += 1;
            h->stats.n_slots += (int)plans[i].n_slots;
            h->stats.device_bytes += (uint64_t)plans[i].n_slots * plans[i].lay.total;
            if (dbg) debug_plan(h, plans[i], *h->planres[i], attempt);
        }
        lap("launches");
        {
            std::vector<int32_t> dev(std::max(nb, 1));
            HIPCHK(hipMemcpy(dev.data(), h->d_status.p, 4 * (size_t)nb, hipMemcpyDeviceToHost));
            for (auto& pl : plans) for (int b : pl.work) status[b] = dev[b];   // only blocks that ran this round
        }
        std::vector<int32_t> again;
        if (dbg) {
            int cnt[16] = {0};
            for (int b : pending) cnt[status[b] & 15]++;
            fprintf(stderr, "[sxg] round %d: ok %d rows %d pool %d steps %d nodes %d long %d range %d band %d\n", attempt, cnt[0], cnt[1], cnt[2], cnt[3],
                    cnt[4], cnt[5], cnt[6], cnt[7]);
        }
        for (int b : pending) {
            BlockMeta& m = h->meta[b];
            if (std::find(nomem_blocks.begin(), nomem_blocks.end(), b) != nomem_blocks.end()) continue;
            if (status[b] == ST_ROWS_OVERFLOW || status[b] == ST_POOL_OVERFLOW || status[b] == ST_TBX_OVERFLOW || status[b] == ST_NODES_OVERFLOW) {
                if (m.tier < 3) { m.tier += 1; again.push_back(b); }
            } else if (status[b] == ST_BAND_MISS && m.rm == 2 && !m.wide_band && !m.no_wide) {
                // The traceback kept leaving the ~1100 columns the plane holds around the backbone hints (a structural variant
                // that carries the alignment further off).  The same packed sweep is repeated with a plane that keeps EVERY
                // strip of every row -- rows x columns dwords, up to ~6 GB for a 26 kbp block, but it cannot miss, it covers
                // every length the packed sweep covers (the 32-bit sweeps end at SXG_POA_MAX_SEQ_LEN_WIDE) and it runs at the
                // packed sweep's speed.
                m.wide_band = true;
                again.push_back(b);
            } else if (status[b] == ST_RANGE_OVERFLOW || status[b] == ST_BAND_MISS) {
                // one step wider at the same capacity tier: packed -> int16 row words -> int32 row words
                if (m.rm != 1) {
                    m.rm = row_mode(m.S, m.maxlen, m.maxlen, m.rm >= 2 ? 0 : 1);
                    m.fits = variant_for_len(m.maxlen, m.rm, &m.variant, m.S.sw);
                    m.variant.CB = m.rm == 2 ? plane_cell_bytes(m.S) : 4;
        m.variant.DS = m.rm == 2 && m.variant.CB == 2 && p16_default_scores(m.S) && !getenv("SXG_POA_NO_DEFAULT_CLASS");
                    if (m.fits) again.push_back(b);
                    else status[b] = ST_TOO_LONG;    // (a score range beyond int16 on a sequence beyond SXG_POA_MAX_SEQ_LEN_WIDE)
                } else status[b] = ST_TOO_LONG;      // (unreachable: the int32 sweep reports neither)
            }
        }
        h->stats.retries += (int)again.size();
        pending.swap(again);
        for (auto& pl : plans) all_plans.push_back(std::move(pl));
    }
    for (int b : pending) status[b] = ST_POOL_OVERFLOW;  // (unreachable: the ladders are shorter than the round limit)
    if (nb) HIPCHK(hipMemcpy(h->d_status.p, status.data(), 4 * (size_t)nb, hipMemcpyHostToDevice));
    lap("status");
    // accounting
    std::vector<unsigned long long> cells((size_t)std::max<int64_t>(h->n_seqs, 1));
    if (h->n_seqs) HIPCHK(hipMemcpy(cells.data(), h->d_cells.p, 8 * (size_t)h->n_seqs, hipMemcpyDeviceToHost));
    uint64_t total = 0, bytes = 0;
    std::vector<uint64_t> blk_cells(std::max(nb, 1), 0), blk_bytes(std::max(nb, 1), 0);
    for (int b = 0; b < nb; ++b) {
        if (status[b] != ST_OK) continue;
        uint64_t cb = 0;
        for (int s = h->h_blk_off[b]; s < h->h_blk_off[b + 1]; ++s) cb += cells[s];
        const BlockMeta& m = h->meta[b];
        const int ncross = m.S.convex ? 3 : (m.S.g == m.S.e ? 1 : 2);
        const int sz = m.rm == 1 ? 4 : 2;   // (banded: cells = band cells, as counted by the kernel)
        blk_cells[b] = cb; blk_bytes[b] = cb * (uint64_t)(2 * ncross * sz + 1);
        total += cb;
        bytes += blk_bytes[b];
    }
    // dominant launch = the one that evaluated the most cells
    for (auto& pl : all_plans) {
        for (int b : pl.work) if (status[b] == ST_OK) { pl.cells += blk_cells[b]; pl.bytes += blk_bytes[b]; }
        if (pl.cells >= h->stats.dom_cells) {
            h->stats.dom_cells = pl.cells; h->stats.dom_algo_bytes = pl.bytes; h->stats.dom_kernel_ms = pl.ms;
            h->stats.dom_threads = pl.variant.T(); h->stats.dom_cols_per_lane = pl.variant.W * (pl.variant.RM >= 2 ? 2 : 1);
            h->stats.dom_row_mode = pl.variant.RM;
            h->stats.dom_clock_mhz = pl.clock_mhz;
        }
    }
    lap("accounting");
    h->stats.kernel_ms = ms_total;
    h->stats.cells = total;
    h->stats.algo_bytes = bytes;
    h->executed = true;
    if (h->want_block_graph) {
        int rc = run_block_graphs(h, status);
        if (rc) return rc;
        lap("block graphs");
    }
    for (int b = 0; b < nb; ++b)
        if (status[b] != ST_OK) return fail(SXG_E_BLOCK, "block " + std::to_string(b) + " failed with status " + std::to_string(status[b]));
    return SXG_OK;
}

// ---------------------------------------------------------------------------------------
// Host result arrays: allocated without std::vector's serial zero-fill (the download overwrites every element), and
// from 8 MiB on 2 MiB-aligned and marked for transparent huge pages -- the headline batch downloads 1.7 GB, and the
// first-touch faults of 4 KiB pages under the copy (and their unmapping at _free) were a third of the
// provider's time outside the kernels.
static void* host_big_alloc(size_t bytes) {
    if (bytes < ((size_t)8 << 20)) return malloc(bytes ? bytes : 1);
    void* q = nullptr;
    if (posix_memalign(&q, (size_t)2 << 20, bytes)) return nullptr;
    madvise(q, bytes, MADV_HUGEPAGE);
    return q;
}
template <class T> struct host_noinit_alloc : std::allocator<T> {
    template <class U> struct rebind { typedef host_noinit_alloc<U> other; };
    T* allocate(size_t n) {
        void* q = host_big_alloc(n * sizeof(T));
        if (!q) throw std::bad_alloc();
        return (T*)q;
    }
    void deallocate(T* q, size_t) noexcept { free(q); }
    template <class U> void construct(U* q) noexcept { ::new ((void*)q) U; }
    template <class U, class... A> void construct(U* q, A&&... a) { ::new ((void*)q) U(std::forward<A>(a)...); }
};
template <class T> using hvec = std::vector<T, host_noinit_alloc<T>>;
struct OutOwner {
    std::shared_ptr<PinPool> pin_pool;
    int32_t* bg_steps_pinned = nullptr;   // the step lists when they were downloaded into a borrowed pinned buffer
    std::vector<int32_t> status, score, msa_cols;
    hvec<int32_t> node_rank, node_group, edge_tail, edge_head, cons_nodes;
    int32_t* seq_path_nodes = nullptr;   // one node id per base: the big one (1.3 GB on the headline batch), never zero-filled
    ~OutOwner() { free(seq_path_nodes); if (bg_steps_pinned && pin_pool) pin_pool->release(bg_steps_pinned); }
    std::vector<int64_t> node_off, edge_off, cons_off, msa_off;
    hvec<uint8_t> node_code;
    hvec<uint32_t> edge_weight;
    std::vector<uint64_t> cells, block_cycles;
    std::vector<char> msa;
    // block graphs
    std::vector<int64_t> bg_node_off, bg_seq_off, bg_edge_off, bg_step_off, bg_cons_off;
    hvec<int32_t> bg_node_len, bg_node_outdeg, bg_edge_to, bg_steps, bg_cons_steps;
    hvec<uint8_t> bg_node_indeg, bg_seq;
};

template <class Tv>
static int gather_download(sxg_poa_handle* h, const DevBuf& src, const std::vector<int64_t>& dst_off, DevBuf& d_srcoff,
                           DevBuf& d_dstoff, DevBuf& d_dense, hvec<Tv>& out) {
    const int nb = h->n_blocks;
    const int64_t total = dst_off[nb];
    out.resize((size_t)std::max<int64_t>(total, 1));
    if (total == 0) return SXG_OK;
    int rc;
    if ((rc = d_dense.ensure(sizeof(Tv) * (size_t)total))) return rc;
    hipLaunchKernelGGL((gather_kernel<Tv>), dim3((unsigned)std::min(nb, 4096)), dim3(256), 0, h->stream, src.as<Tv>(),
                       d_dense.as<Tv>(), d_srcoff.as<int64_t>(), d_dstoff.as<int64_t>(), nb);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out.data(), d_dense.p, sizeof(Tv) * (size_t)total, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SXG_OK;
}

extern "C" int sxg_poa_batch_download(sxg_poa_handle* h, sxg_poa_batch_out* out) {
    if (!h || !out) return fail(SXG_E_INVALID, "NULL argument");
    memset(out, 0, sizeof(*out));
    if (!h->have_batch || !h->executed) return fail(SXG_E_INVALID, "no executed batch to download");
    HIPCHK(hipSetDevice(h->device));
    const RoctxRange range_("sxg_poa_batch_download");
    HostLaps laps;
    const int nb = h->n_blocks;
    const int64_t ns = h->n_seqs;
    OutOwner* o = new OutOwner();
    out->_owner = o;
    out->n_blocks = nb; out->n_seqs = ns;
    o->status.resize(std::max(nb, 1));
    std::vector<int32_t> nn(std::max(nb, 1)), ne(std::max(nb, 1)), nc(std::max(nb, 1));
    if (nb) {
        HIPCHK(hipMemcpy(o->status.data(), h->d_status.p, 4 * (size_t)nb, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(nn.data(), h->d_nn.p, 4 * (size_t)nb, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ne.data(), h->d_ne.p, 4 * (size_t)nb, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(nc.data(), h->d_nc.p, 4 * (size_t)nb, hipMemcpyDeviceToHost));
    }
    o->node_off.assign(nb + 1, 0); o->edge_off.assign(nb + 1, 0); o->cons_off.assign(nb + 1, 0);
    std::vector<int64_t> src_off(nb + 1, 0);
    for (int b = 0; b < nb; ++b) {
        o->node_off[b + 1] = o->node_off[b] + nn[b];
        o->edge_off[b + 1] = o->edge_off[b] + ne[b];
        o->cons_off[b + 1] = o->cons_off[b] + nc[b];
        src_off[b] = h->h_seq_off[h->h_blk_off[b]];
    }
    int rc;
    if ((rc = h->d_tmp_a.ensure(8 * (size_t)(nb + 1))) || (rc = h->d_tmp_b.ensure(8 * (size_t)(nb + 1)))) { sxg_poa_batch_free(out); return rc; }
    HIPCHK(hipMemcpy(h->d_tmp_a.p, src_off.data(), 8 * (size_t)(nb + 1), hipMemcpyHostToDevice));
    auto put_off = [&](const std::vector<int64_t>& off) -> int {
        hipError_t e = hipMemcpy(h->d_tmp_b.p, off.data(), 8 * (size_t)(nb + 1), hipMemcpyHostToDevice);
        return e == hipSuccess ? SXG_OK : fail(SXG_E_NODEVICE, hipGetErrorString(e));
    };
#define GD(T, src, off, dst)                                                                         \
    if ((rc = put_off(off)) || (rc = gather_download<T>(h, src, off, h->d_tmp_a, h->d_tmp_b, h->d_tmp_c, dst))) { \
        sxg_poa_batch_free(out);                                                                     \
        return rc;                                                                                   \
    }
    const bool with_graphs = !(h->want_block_graph == 3 && h->bg_done);   // (3: the caller laces block graphs and reads nothing else)
    if (with_graphs) {
    GD(uint8_t, h->d_node_code, o->node_off, o->node_code)
    GD(int32_t, h->d_node_rank, o->node_off, o->node_rank)
    GD(int32_t, h->d_node_group, o->node_off, o->node_group)
    GD(int32_t, h->d_edge_tail, o->edge_off, o->edge_tail)
    GD(int32_t, h->d_edge_head, o->edge_off, o->edge_head)
    GD(uint32_t, h->d_edge_w, o->edge_off, o->edge_weight)
    if (h->want_consensus) { GD(int32_t, h->d_cons, o->cons_off, o->cons_nodes) }
    }
#undef GD
    laps.lap("download: POA graphs");
    const bool with_paths = !(h->want_block_graph >= 2 && h->bg_done);
    if (with_paths) {
        o->seq_path_nodes = (int32_t*)host_big_alloc(4 * (size_t)std::max<int64_t>(h->n_bases, 1));
        if (!o->seq_path_nodes) { sxg_poa_batch_free(out); return fail(SXG_E_NOMEM, "host allocation of the path array failed"); }
        if (h->n_bases) HIPCHK(hipMemcpy(o->seq_path_nodes, h->d_paths.p, 4 * (size_t)h->n_bases, hipMemcpyDeviceToHost));
    }
    o->score.resize((size_t)std::max<int64_t>(ns, 1));
    o->cells.resize((size_t)std::max<int64_t>(ns, 1));
    laps.lap("download: paths");
    if (h->want_block_graph && h->bg_done) {
        // dense offsets from the per-block counts of the block-graph kernel; the arrays are gathered on the device
        o->bg_node_off.assign(nb + 1, 0); o->bg_seq_off.assign(nb + 1, 0); o->bg_edge_off.assign(nb + 1, 0); o->bg_cons_off.assign(nb + 1, 0);
        std::vector<int64_t> blk_steps(nb + 1, 0), src_no(nb + 1, 0), src_eo(nb + 1, 0);
        std::vector<int32_t> nsteps((size_t)std::max<int64_t>(ns, 1), 0);
        if (ns) HIPCHK(hipMemcpy(nsteps.data(), h->d_bg_nsteps.p, 4 * (size_t)ns, hipMemcpyDeviceToHost));
        o->bg_step_off.assign((size_t)ns + 1, 0);
        for (int64_t sq = 0; sq < ns; ++sq) o->bg_step_off[(size_t)sq + 1] = o->bg_step_off[(size_t)sq] + nsteps[(size_t)sq];
        for (int b = 0; b < nb; ++b) {
            const int32_t* c = h->bg_counts.data() + (size_t)BGC_N * b;
            const bool ok = o->status[b] == ST_OK;
            o->bg_node_off[b + 1] = o->bg_node_off[b] + (ok ? c[BGC_NODES] : 0);
            o->bg_seq_off[b + 1] = o->bg_seq_off[b] + (ok ? c[BGC_SEQ] : 0);
            o->bg_edge_off[b + 1] = o->bg_edge_off[b] + (ok ? c[BGC_EDGES] : 0);
            o->bg_cons_off[b + 1] = o->bg_cons_off[b] + (ok && h->want_consensus ? c[BGC_CONS] : 0);
            blk_steps[b + 1] = o->bg_step_off[(size_t)h->h_blk_off[b + 1]];
            src_no[b] = h->bg_node_o[(size_t)b]; src_eo[b] = h->bg_edge_o[(size_t)b];
        }
        auto put_src = [&](const std::vector<int64_t>& off) -> int {
            hipError_t e = hipMemcpy(h->d_tmp_a.p, off.data(), 8 * (size_t)(nb + 1), hipMemcpyHostToDevice);
            return e == hipSuccess ? SXG_OK : fail(SXG_E_NODEVICE, hipGetErrorString(e));
        };
#define GB_(T, srcoff, src, off, dst)                                                                                      \
        if ((rc = put_src(srcoff)) || (rc = put_off(off)) || (rc = gather_download<T>(h, src, off, h->d_tmp_a, h->d_tmp_b, h->d_tmp_c, dst))) { \
            sxg_poa_batch_free(out);                                                                                       \
            return rc;                                                                                                     \
        }
        GB_(int32_t, src_no, h->d_bg_len, o->bg_node_off, o->bg_node_len)
        GB_(int32_t, src_no, h->d_bg_od, o->bg_node_off, o->bg_node_outdeg)
        GB_(uint8_t, src_no, h->d_bg_id, o->bg_node_off, o->bg_node_indeg)
        GB_(uint8_t, src_no, h->d_bg_seq, o->bg_seq_off, o->bg_seq)
        GB_(int32_t, src_eo, h->d_bg_eto, o->bg_edge_off, o->bg_edge_to)
        {   // the step lists: into a borrowed pinned buffer when the pool has one (see PinPool)
            const int64_t total = blk_steps[nb];
            if (h->pin_thread.joinable()) h->pin_thread.join();
            if (total >= ((int64_t)16 << 20) && h->pins && !getenv("SXG_POA_NO_PINNED")) {
                if (void* q = h->pins->acquire(4 * (size_t)total)) { o->pin_pool = h->pins; o->bg_steps_pinned = (int32_t*)q; }
            }
            if (o->bg_steps_pinned) {
                if ((rc = put_src(src_off)) || (rc = put_off(blk_steps)) || (rc = h->d_tmp_c.ensure(4 * (size_t)total))) { sxg_poa_batch_free(out); return rc; }
                hipLaunchKernelGGL((gather_kernel<int32_t>), dim3((unsigned)std::min(nb, 4096)), dim3(256), 0, h->stream, h->d_bg_steps.as<int32_t>(),
                                   h->d_tmp_c.as<int32_t>(), h->d_tmp_a.as<int64_t>(), h->d_tmp_b.as<int64_t>(), nb);
                hipError_t e = hipGetLastError();
                if (e == hipSuccess) e = hipMemcpyAsync(o->bg_steps_pinned, h->d_tmp_c.p, 4 * (size_t)total, hipMemcpyDeviceToHost, h->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
                if (e != hipSuccess) { sxg_poa_batch_free(out); return fail(SXG_E_NODEVICE, std::string("download of the step lists: ") + hipGetErrorString(e)); }
            } else {
                GB_(int32_t, src_off, h->d_bg_steps, blk_steps, o->bg_steps)
            }
        }
        if (h->want_consensus) { GB_(int32_t, src_no, h->d_bg_cons, o->bg_cons_off, o->bg_cons_steps) }
#undef GB_
        out->bg_node_off = o->bg_node_off.data(); out->bg_node_len = o->bg_node_len.data(); out->bg_node_outdeg = o->bg_node_outdeg.data();
        out->bg_node_indeg = o->bg_node_indeg.data(); out->bg_seq_off = o->bg_seq_off.data(); out->bg_seq = (char*)o->bg_seq.data();
        out->bg_edge_off = o->bg_edge_off.data(); out->bg_edge_to = o->bg_edge_to.data(); out->bg_step_off = o->bg_step_off.data();
        out->bg_steps = o->bg_steps_pinned ? o->bg_steps_pinned : o->bg_steps.data();
        if (h->want_consensus) { out->bg_cons_off = o->bg_cons_off.data(); out->bg_cons_steps = o->bg_cons_steps.data(); }
        laps.lap("download: block graphs");
    }
    o->block_cycles.assign((size_t)std::max(nb, 1), 0);
    if (nb) HIPCHK(hipMemcpy(o->block_cycles.data(), h->d_blk_cycles.p, 8 * (size_t)nb, hipMemcpyDeviceToHost));
    out->block_cycles = o->block_cycles.data();
    if (ns) {
        HIPCHK(hipMemcpy(o->score.data(), h->d_score.p, 4 * (size_t)ns, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(o->cells.data(), h->d_cells.p, 8 * (size_t)ns, hipMemcpyDeviceToHost));
    }
    out->status = o->status.data();
    out->node_off = o->node_off.data(); out->edge_off = o->edge_off.data();
    if (with_graphs) {
        out->node_code = o->node_code.data(); out->node_rank = o->node_rank.data(); out->node_group = o->node_group.data();
        out->edge_tail = o->edge_tail.data(); out->edge_head = o->edge_head.data(); out->edge_weight = o->edge_weight.data();
    }
    out->seq_path_nodes = o->seq_path_nodes; out->score = o->score.data(); out->cells = o->cells.data();
    if (h->want_consensus) { out->cons_off = o->cons_off.data(); if (with_graphs) out->cons_nodes = o->cons_nodes.data(); }
    if (h->want_msa && o->seq_path_nodes) {
        // S8: MSA column = aligned group in rank order; pure formatting of device results
        static const char dec[5] = {'A', 'C', 'G', 'T', 'N'};
        o->msa_off.assign(nb + 1, 0); o->msa_cols.assign(std::max(nb, 1), 0);
        std::vector<std::vector<int32_t>> cols(nb);
        for (int b = 0; b < nb; ++b) {
            const int64_t n0 = o->node_off[b];
            const int n = nn[b];
            std::vector<int32_t> by_rank(n);
            for (int v = 0; v < n; ++v) by_rank[o->node_rank[n0 + v]] = v;
            cols[b].assign(n, 0);
            int ncol = 0;
            for (int r = 0; r < n; ++r) {
                const int v = by_rank[r];
                if (r > 0 && o->node_group[n0 + by_rank[r - 1]] == o->node_group[n0 + v]) cols[b][v] = ncol - 1;
                else cols[b][v] = ncol++;
            }
            o->msa_cols[b] = ncol;
            const int rows = (h->h_blk_off[b + 1] - h->h_blk_off[b]) + (h->want_consensus ? 1 : 0);
            o->msa_off[b + 1] = o->msa_off[b] + (o->status[b] == ST_OK ? (int64_t)rows * ncol : 0);
        }
        o->msa.assign((size_t)std::max<int64_t>(o->msa_off[nb], 1), '-');
        for (int b = 0; b < nb; ++b) {
            if (o->status[b] != ST_OK) continue;
            const int64_t n0 = o->node_off[b];
            const int ncol = o->msa_cols[b];
            char* base = o->msa.data() + o->msa_off[b];
            int row = 0;
            for (int s = h->h_blk_off[b]; s < h->h_blk_off[b + 1]; ++s, ++row)
                for (int64_t k = h->h_seq_off[s]; k < h->h_seq_off[s + 1]; ++k) {
                    const int v = o->seq_path_nodes[k];
                    base[(int64_t)row * ncol + cols[b][v]] = dec[o->node_code[n0 + v]];
                }
            if (h->want_consensus)
                for (int64_t k = o->cons_off[b]; k < o->cons_off[b + 1]; ++k) {
                    const int v = o->cons_nodes[k];
                    base[(int64_t)row * ncol + cols[b][v]] = dec[o->node_code[n0 + v]];
                }
        }
        out->msa_off = o->msa_off.data(); out->msa_cols = o->msa_cols.data(); out->msa = o->msa.data();
    }
    return SXG_OK;
}

extern "C" int sxg_poa_batch_device_view(sxg_poa_handle* h, sxg_poa_device_view* v) {
    if (!h || !v) return fail(SXG_E_INVALID, "NULL argument");
    memset(v, 0, sizeof(*v));
    if (!h->have_batch || !h->executed) return fail(SXG_E_INVALID, "no executed batch");
    v->n_blocks = h->n_blocks; v->n_seqs = h->n_seqs; v->n_bases = h->n_bases;
    v->status = h->d_status.as<int32_t>(); v->n_nodes = h->d_nn.as<int32_t>(); v->n_edges = h->d_ne.as<int32_t>();
    v->node_code = h->d_node_code.as<uint8_t>(); v->node_rank = h->d_node_rank.as<int32_t>();
    v->node_group = h->d_node_group.as<int32_t>(); v->edge_tail = h->d_edge_tail.as<int32_t>();
    v->edge_head = h->d_edge_head.as<int32_t>(); v->edge_weight = h->d_edge_w.as<uint32_t>();
    v->seq_path_nodes = h->d_paths.as<int32_t>(); v->score = h->d_score.as<int32_t>();
    return SXG_OK;
}

extern "C" void sxg_poa_batch_free(sxg_poa_batch_out* out) {
    if (!out) return;
    delete (OutOwner*)out->_owner;
    memset(out, 0, sizeof(*out));
}

extern "C" int sxg_poa_batch_run(sxg_poa_handle* h, const sxg_poa_batch_in* in, sxg_poa_batch_out* out) {
    if (out) memset(out, 0, sizeof(*out));
    int rc = sxg_poa_batch_upload(h, in);
    if (rc) return rc;
    rc = sxg_poa_batch_execute(h);
    if (rc && rc != SXG_E_BLOCK) return rc;
    const std::string keep = g_err;
    int rc2 = sxg_poa_batch_download(h, out);
    if (rc2) return rc2;
    if (rc) g_err = keep;
    return rc;
}

// ---------------------------------------------------------------------------------------
// Multi-GPU: blocks are independent (src/smooth.cpp:1931), so every rank aligns its own share and the only
// exchange is the reassembly of the per-block results on the rank that laces (src/main.cpp:599+).
// sxg_poa_batch_run_sharded is sxg_poa_batch_run for a communicator: EVERY rank calls it with the SAME batch,
// the blocks are dealt by cost (longest processing time first), each rank runs its share, packs its
// dense results into one device blob, the blob sizes are all-gathered and the blobs travel to the root with ONE
// grouped ncclSend/ncclRecv per peer of exactly that size -- device to device over xGMI, no padding to the
// largest rank, buffers kept by the handle.  The root assembles the results in the ORIGINAL block order.
// RCCL is OPTIONAL at load time: a single-GPU user of libsxgpoa.so needs no librccl.so.  The entry points are resolved
// with dlopen/dlsym the first time a communicator is asked for (in a process that already loaded RCCL -- PyTorch-ROCm --
// the SONAME resolves to that very instance).
namespace {
struct RcclApi {
    decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&::ncclCommInitRank) CommInitRank = nullptr;
    decltype(&::ncclCommDestroy) CommDestroy = nullptr;
    decltype(&::ncclCommAbort) CommAbort = nullptr;
    decltype(&::ncclCommGetAsyncError) CommGetAsyncError = nullptr;
    decltype(&::ncclAllGather) AllGather = nullptr;
    decltype(&::ncclGroupStart) GroupStart = nullptr;
    decltype(&::ncclGroupEnd) GroupEnd = nullptr;
    decltype(&::ncclSend) Send = nullptr;
    decltype(&::ncclRecv) Recv = nullptr;
    decltype(&::ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
    std::string why;
};
RcclApi& rccl_api() {
    static RcclApi A = [] {
        RcclApi a;
        void* lib = nullptr;
        // SXG_POA_RCCL_LIB names the library instead of the default search (tests point it at a file that does not exist to
        // take the RCCL-missing path: SXG_E_NODEVICE with a message, not a crash)
        if (const char* forced = getenv("SXG_POA_RCCL_LIB")) lib = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
        else
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!lib) {   // (dlerror() clears the error it returns: ask once)
            const char* e = dlerror();
            a.why = std::string("librccl.so not found (") + (e ? e : "dlopen failed") + ")";
            return a;
        }
        bool all = true;
#define SXG_RCCL_SYM(f) do { a.f = (decltype(a.f))dlsym(lib, "nccl" #f); if (!a.f) { all = false; a.why += " nccl" #f; } } while (0)
        SXG_RCCL_SYM(GetUniqueId); SXG_RCCL_SYM(CommInitRank); SXG_RCCL_SYM(CommDestroy); SXG_RCCL_SYM(CommAbort);
        SXG_RCCL_SYM(CommGetAsyncError); SXG_RCCL_SYM(AllGather); SXG_RCCL_SYM(GroupStart); SXG_RCCL_SYM(GroupEnd);
        SXG_RCCL_SYM(Send); SXG_RCCL_SYM(Recv); SXG_RCCL_SYM(GetErrorString);
#undef SXG_RCCL_SYM
        a.ok = all;
        if (!all) a.why = "librccl.so lacks" + a.why;
        return a;
    }();
    return A;
}
}  // namespace
#define RCCL_NEEDED() do { if (!rccl_api().ok) return fail(SXG_E_NODEVICE, "multi-GPU needs RCCL: " + rccl_api().why); } while (0)
#define ncclGetUniqueId rccl_api().GetUniqueId
#define ncclCommInitRank rccl_api().CommInitRank
#define ncclCommDestroy rccl_api().CommDestroy
#define ncclCommAbort rccl_api().CommAbort
#define ncclCommGetAsyncError rccl_api().CommGetAsyncError
#define ncclAllGather rccl_api().AllGather
#define ncclGroupStart rccl_api().GroupStart
#define ncclGroupEnd rccl_api().GroupEnd
#define ncclSend rccl_api().Send
#define ncclRecv rccl_api().Recv
#define ncclGetErrorString rccl_api().GetErrorString

#define NCCLCHK(x)                                                                                      \
    do {                                                                                               \
        ncclResult_t _r = (x);                                                                         \
        if (_r != ncclSuccess) return fail(SXG_E_NODEVICE, std::string(#x) + ": " + ncclGetErrorString(_r)); \
    } while (0)

constexpr int BC_N_WORDS = 16;   // words per rank in the count exchange (BC_N below)
extern "C" int sxg_poa_comm_unique_id(uint8_t* id) {
    if (!id) return fail(SXG_E_INVALID, "NULL argument");
    static_assert(sizeof(ncclUniqueId) == SXG_POA_COMM_ID_BYTES, "ncclUniqueId size");
    RCCL_NEEDED();
    ncclUniqueId u;
    NCCLCHK(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return SXG_OK;
}
extern "C" void sxg_poa_comm_destroy(sxg_poa_handle* h) {
    if (!h) return;
    if (h->comm && h->own_comm) (void)ncclCommDestroy(h->comm);
    h->comm = nullptr; h->own_comm = false; h->nranks = 1; h->rank = 0;
}
extern "C" int sxg_poa_comm_init(sxg_poa_handle* h, const uint8_t* id, int nranks, int rank) {
    if (!h || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(SXG_E_INVALID, "bad argument");
    RCCL_NEEDED();
    HIPCHK(hipSetDevice(h->device));
    sxg_poa_comm_destroy(h);
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    NCCLCHK(ncclCommInitRank(&h->comm, nranks, u, rank));
    h->own_comm = true; h->nranks = nranks; h->rank = rank;
    // (the exchange buffer of the sharded run's counts: allocated here so that the run itself cannot fail before its collectives)
    return h->d_counts.ensure(8 * (size_t)BC_N_WORDS * (size_t)(nranks + 1));
}
extern "C" int sxg_poa_comm_attach(sxg_poa_handle* h, void* nccl_comm, int nranks, int rank) {
    if (!h || !nccl_comm || nranks < 1 || rank < 0 || rank >= nranks) return fail(SXG_E_INVALID, "bad argument");
    RCCL_NEEDED();
    sxg_poa_comm_destroy(h);
    h->comm = (ncclComm_t)nccl_comm; h->own_comm = false; h->nranks = nranks; h->rank = rank;
    HIPCHK(hipSetDevice(h->device));
    return h->d_counts.ensure(8 * (size_t)BC_N_WORDS * (size_t)(nranks + 1));
}

namespace {
// blocks by descending cost (ties: lower id) onto the least-loaded rank (ties: lower rank)
void lpt_partition(const sxg_poa_batch_in* in, int nranks, std::vector<std::vector<int32_t>>& parts) {
    const int nb = in->n_blocks;
    std::vector<double> cost(std::max(nb, 1), 0.0);
    for (int b = 0; b < nb; ++b) {
        double prev = 0;
        const double l1 = in->blk_off[b + 1] > in->blk_off[b] ? (double)(in->seq_off[in->blk_off[b] + 1] - in->seq_off[in->blk_off[b]]) : 0.0;
        for (int sq = in->blk_off[b]; sq < in->blk_off[b + 1]; ++sq) {
            const double len = (double)(in->seq_off[sq + 1] - in->seq_off[sq]);
            if (sq > in->blk_off[b]) cost[b] += len * (l1 + 0.05 * prev);
            prev += len;
        }
    }
    std::vector<int32_t> order(nb);
    for (int b = 0; b < nb; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return cost[a] > cost[c]; });
    parts.assign(nranks, {});
    std::vector<double> load(nranks, 0.0);
    for (int b : order) {
        int r = 0;
        for (int k = 1; k < nranks; ++k) if (load[k] < load[r]) r = k;
        parts[r].push_back(b);
        load[r] += cost[b];
    }
    for (auto& pt : parts) std::sort(pt.begin(), pt.end());
}
struct LocalBatch {
    std::vector<int32_t> blk_off{0};
    std::vector<int64_t> seq_off{0};
    std::vector<uint8_t> bases;
    std::vector<uint32_t> weights;
    std::vector<sxg_poa_params> params;
    std::vector<int32_t> trims;
    sxg_poa_batch_in in;
};
void build_local(const sxg_poa_batch_in* in, const std::vector<int32_t>& part, LocalBatch& L) {
    for (int b : part) {
        for (int sq = in->blk_off[b]; sq < in->blk_off[b + 1]; ++sq) {
            L.bases.insert(L.bases.end(), in->bases + in->seq_off[sq], in->bases + in->seq_off[sq + 1]);
            L.seq_off.push_back((int64_t)L.bases.size());
            L.weights.push_back(in->weights ? in->weights[sq] : 1u);
        }
        L.blk_off.push_back((int32_t)(L.seq_off.size() - 1));
        if (in->per_block_params) L.params.push_back(in->params[b]);
    }
    if (!in->per_block_params) L.params.push_back(in->params[0]);
    if (L.bases.empty()) L.bases.push_back(0);
    if (L.weights.empty()) L.weights.push_back(1);
    memset(&L.in, 0, sizeof(L.in));
    L.in.n_blocks = (int32_t)part.size(); L.in.blk_off = L.blk_off.data(); L.in.seq_off = L.seq_off.data();
    L.in.bases = L.bases.data(); L.in.weights = L.weights.data(); L.in.params = L.params.data();
    L.in.per_block_params = in->per_block_params; L.in.want_consensus = in->want_consensus; L.in.want_msa = 0;
    // the owning rank builds the block graphs of its blocks (the root only laces); the MSA is formatted on the root from
    // the per-base paths, which then travel as well
    L.in.want_block_graph = in->want_block_graph >= 2 ? (in->want_msa ? 1 : 2) : in->want_block_graph;   // (3: the raw POA graphs travel with the blob; the root drops them)
    L.in.bg_consensus_visited_only = in->bg_consensus_visited_only;
    if (in->want_block_graph) {
        for (int b : part) L.trims.push_back(in->bg_trim ? in->bg_trim[b] : 0);
        if (L.trims.empty()) L.trims.push_back(0);
        L.in.bg_trim = L.trims.data();
    }
}
// blob of one rank: eight counts, then the arrays, every section 16-byte aligned
enum { BC_NB = 0, BC_NS, BC_NBASES, BC_NODES, BC_EDGES, BC_CONS, BC_BYTES, BC_PAD /* error code of the rank (0 = fine) */,
       // block graphs of the rank's blocks (want_block_graph): the owning rank builds them, the root only laces
       BC_BG /* 0 = none, 1 = with, 2 = instead of the per-base paths */, BC_BG_NODES, BC_BG_EDGES, BC_BG_SEQ, BC_BG_STEPS, BC_BG_CONS,
       BC_RES0, BC_RES1, BC_N };
static_assert(BC_N == BC_N_WORDS, "count words");
struct BlobLayout { size_t status, nn, ne, nc, score, cells, code, rank, group, et, eh, ew, paths, cons,
                    bg_counts, bg_nsteps, bg_len, bg_od, bg_id, bg_seq, bg_eto, bg_steps, bg_cons, total; };
BlobLayout blob_layout(const int64_t* c) {
    BlobLayout B;
    size_t cur = 0;
    auto sec = [&](size_t bytes) { size_t o = cur; cur += (bytes + 15) & ~(size_t)15; return o; };
    B.status = sec(4 * (size_t)c[BC_NB]); B.nn = sec(4 * (size_t)c[BC_NB]); B.ne = sec(4 * (size_t)c[BC_NB]); B.nc = sec(4 * (size_t)c[BC_NB]);
    B.score = sec(4 * (size_t)c[BC_NS]); B.cells = sec(8 * (size_t)c[BC_NS]);
    B.code = sec((size_t)c[BC_NODES]); B.rank = sec(4 * (size_t)c[BC_NODES]); B.group = sec(4 * (size_t)c[BC_NODES]);
    B.et = sec(4 * (size_t)c[BC_EDGES]); B.eh = sec(4 * (size_t)c[BC_EDGES]); B.ew = sec(4 * (size_t)c[BC_EDGES]);
    B.paths = sec(c[BC_BG] == 2 ? 0 : 4 * (size_t)c[BC_NBASES]); B.cons = sec(4 * (size_t)c[BC_CONS]);
    const bool bg = c[BC_BG] != 0;
    B.bg_counts = sec(bg ? 4 * (size_t)BGC_N * (size_t)c[BC_NB] : 0); B.bg_nsteps = sec(bg ? 4 * (size_t)c[BC_NS] : 0);
    B.bg_len = sec(bg ? 4 * (size_t)c[BC_BG_NODES] : 0); B.bg_od = sec(bg ? 4 * (size_t)c[BC_BG_NODES] : 0); B.bg_id = sec(bg ? (size_t)c[BC_BG_NODES] : 0);
    B.bg_seq = sec(bg ? (size_t)c[BC_BG_SEQ] : 0); B.bg_eto = sec(bg ? 4 * (size_t)c[BC_BG_EDGES] : 0);
    B.bg_steps = sec(bg ? 4 * (size_t)c[BC_BG_STEPS] : 0); B.bg_cons = sec(bg ? 4 * (size_t)c[BC_BG_CONS] : 0);
    B.total = cur;
    return B;
}
// dense results of the handle's executed batch, packed into h->d_blob on the device
int pack_blob(sxg_poa_handle* h, int64_t* counts) {
    const int nb = h->n_blocks;
    std::vector<int32_t> nn(std::max(nb, 1)), ne(std::max(nb, 1)), nc(std::max(nb, 1));
    if (nb) {
        HIPCHK(hipMemcpy(nn.data(), h->d_nn.p, 4 * (size_t)nb, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ne.data(), h->d_ne.p, 4 * (size_t)nb, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(nc.data(), h->d_nc.p, 4 * (size_t)nb, hipMemcpyDeviceToHost));
    }
    std::vector<int64_t> noff(nb + 1, 0), eoff(nb + 1, 0), coff(nb + 1, 0), soff(nb + 1, 0);
    for (int b = 0; b < nb; ++b) {
        noff[b + 1] = noff[b] + nn[b]; eoff[b + 1] = eoff[b] + ne[b]; coff[b + 1] = coff[b] + (h->want_consensus ? nc[b] : 0);
        soff[b] = h->h_seq_off[h->h_blk_off[b]];
    }
    memset(counts, 0, sizeof(int64_t) * BC_N);
    counts[BC_NB] = nb; counts[BC_NS] = h->n_seqs; counts[BC_NBASES] = h->n_bases;
    counts[BC_NODES] = noff[nb]; counts[BC_EDGES] = eoff[nb]; counts[BC_CONS] = coff[nb];
    // block graphs: dense offsets from the block-graph kernel's counts (as sxg_poa_batch_download computes them)
    const bool bg = h->want_block_graph && h->bg_done;
    std::vector<int64_t> gno(nb + 1, 0), gso(nb + 1, 0), geo(nb + 1, 0), gco(nb + 1, 0), gpo(nb + 1, 0), src_no(nb + 1, 0), src_eo(nb + 1, 0);
    if (bg) {
        std::vector<int32_t> st(std::max(nb, 1)), nsteps((size_t)std::max<int64_t>(h->n_seqs, 1), 0);
        if (nb) HIPCHK(hipMemcpy(st.data(), h->d_status.p, 4 * (size_t)nb, hipMemcpyDeviceToHost));
        if (h->n_seqs) HIPCHK(hipMemcpy(nsteps.data(), h->d_bg_nsteps.p, 4 * (size_t)h->n_seqs, hipMemcpyDeviceToHost));
        for (int b = 0; b < nb; ++b) {
            const int32_t* c = h->bg_counts.data() + (size_t)BGC_N * b;
            const bool ok = st[b] == ST_OK;
            gno[b + 1] = gno[b] + (ok ? c[BGC_NODES] : 0); gso[b + 1] = gso[b] + (ok ? c[BGC_SEQ] : 0);
            geo[b + 1] = geo[b] + (ok ? c[BGC_EDGES] : 0); gco[b + 1] = gco[b] + (ok && h->want_consensus ? c[BGC_CONS] : 0);
            int64_t stp = 0;
            for (int sq = h->h_blk_off[b]; sq < h->h_blk_off[b + 1]; ++sq) stp += nsteps[(size_t)sq];
            gpo[b + 1] = gpo[b] + (ok ? stp : 0);
            src_no[b] = h->bg_node_o[(size_t)b]; src_eo[b] = h->bg_edge_o[(size_t)b];
        }
        counts[BC_BG] = h->want_block_graph; counts[BC_BG_NODES] = gno[nb]; counts[BC_BG_EDGES] = geo[nb]; counts[BC_BG_SEQ] = gso[nb];
        counts[BC_BG_STEPS] = gpo[nb]; counts[BC_BG_CONS] = gco[nb];
    }
    const BlobLayout B = blob_layout(counts);
    counts[BC_BYTES] = (int64_t)B.total;
    int rc;
    if ((rc = h->d_blob.ensure(B.total + 16))) return rc;
    if ((rc = h->d_tmp_a.ensure(8 * (size_t)(nb + 1))) || (rc = h->d_tmp_b.ensure(8 * (size_t)(nb + 1)))) return rc;
    uint8_t* blob = h->d_blob.as<uint8_t>();
    if (nb) {
        HIPCHK(hipMemcpyAsync(blob + B.status, h->d_status.p, 4 * (size_t)nb, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(blob + B.nn, h->d_nn.p, 4 * (size_t)nb, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(blob + B.ne, h->d_ne.p, 4 * (size_t)nb, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(blob + B.nc, h->d_nc.p, 4 * (size_t)nb, hipMemcpyDeviceToDevice, h->stream));
    }
    if (h->n_seqs) {
        HIPCHK(hipMemcpyAsync(blob + B.score, h->d_score.p, 4 * (size_t)h->n_seqs, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(blob + B.cells, h->d_cells.p, 8 * (size_t)h->n_seqs, hipMemcpyDeviceToDevice, h->stream));
    }
    if (h->n_bases && counts[BC_BG] != 2) HIPCHK(hipMemcpyAsync(blob + B.paths, h->d_paths.p, 4 * (size_t)h->n_bases, hipMemcpyDeviceToDevice, h->stream));
    if (bg && nb) HIPCHK(hipMemcpyAsync(blob + B.bg_counts, h->d_bg_counts.p, 4 * (size_t)BGC_N * (size_t)nb, hipMemcpyDeviceToDevice, h->stream));
    if (bg && h->n_seqs) HIPCHK(hipMemcpyAsync(blob + B.bg_nsteps, h->d_bg_nsteps.p, 4 * (size_t)h->n_seqs, hipMemcpyDeviceToDevice, h->stream));
    const std::vector<int64_t>* cur_src = &soff;
    auto gather = [&](auto tag, const DevBuf& src, const std::vector<int64_t>& off, size_t at) -> int {
        typedef decltype(tag) Tv;
        if (off[nb] == 0) return SXG_OK;
        HIPCHK(hipMemcpyAsync(h->d_tmp_a.p, cur_src->data(), 8 * (size_t)(nb + 1), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_tmp_b.p, off.data(), 8 * (size_t)(nb + 1), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL((gather_kernel<Tv>), dim3((unsigned)std::min(nb, 4096)), dim3(256), 0, h->stream, src.as<Tv>(), (Tv*)(blob + at),
                           h->d_tmp_a.as<int64_t>(), h->d_tmp_b.as<int64_t>(), nb);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));   // (d_tmp_b is reused by the next array)
        return SXG_OK;
    };
    if ((rc = gather(uint8_t(), h->d_node_code, noff, B.code)) || (rc = gather(int32_t(), h->d_node_rank, noff, B.rank)) ||
        (rc = gather(int32_t(), h->d_node_group, noff, B.group)) || (rc = gather(int32_t(), h->d_edge_tail, eoff, B.et)) ||
        (rc = gather(int32_t(), h->d_edge_head, eoff, B.eh)) || (rc = gather(uint32_t(), h->d_edge_w, eoff, B.ew)))
        return rc;
    if (h->want_consensus && (rc = gather(int32_t(), h->d_cons, coff, B.cons))) return rc;
    if (bg) {
        cur_src = &src_no;
        if ((rc = gather(int32_t(), h->d_bg_len, gno, B.bg_len)) || (rc = gather(int32_t(), h->d_bg_od, gno, B.bg_od)) ||
            (rc = gather(uint8_t(), h->d_bg_id, gno, B.bg_id)) || (rc = gather(uint8_t(), h->d_bg_seq, gso, B.bg_seq)))
            return rc;
        if (h->want_consensus && (rc = gather(int32_t(), h->d_bg_cons, gco, B.bg_cons))) return rc;
        cur_src = &src_eo;
        if ((rc = gather(int32_t(), h->d_bg_eto, geo, B.bg_eto))) return rc;
        cur_src = &soff;
        if ((rc = gather(int32_t(), h->d_bg_steps, gpo, B.bg_steps))) return rc;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return SXG_OK;
}
void format_msa(OutOwner* o, const sxg_poa_batch_in* in, bool want_consensus);
// root: results of all ranks (host copies of their blobs) -> one result set in the batch's block order
int assemble(const sxg_poa_batch_in* in, const std::vector<std::vector<int32_t>>& parts, const std::vector<std::vector<uint8_t>>& blobs,
             const std::vector<int64_t>& counts, sxg_poa_batch_out* out) {
    const int nb = in->n_blocks, nranks = (int)parts.size();
    const int64_t ns = nb ? in->blk_off[nb] : 0, nbases = ns ? in->seq_off[ns] : 0;
    OutOwner* o = new OutOwner();
    out->_owner = o;
    out->n_blocks = nb; out->n_seqs = ns;
    o->status.assign(std::max(nb, 1), 0); o->score.assign((size_t)std::max<int64_t>(ns, 1), 0); o->cells.assign((size_t)std::max<int64_t>(ns, 1), 0);
    o->node_off.assign(nb + 1, 0); o->edge_off.assign(nb + 1, 0); o->cons_off.assign(nb + 1, 0);
    int bg_mode = 0;   // what the ranks sent (the same on all of them: every rank was handed the same request)
    for (int r = 0; r < nranks; ++r) if (counts[(size_t)r * BC_N + BC_NB] > 0) bg_mode = std::max(bg_mode, (int)counts[(size_t)r * BC_N + BC_BG]);
    const bool with_paths = bg_mode != 2;
    const bool with_graphs = true;   // (a sharded run always carries the raw POA graphs to the root)
    if (with_paths) {
        o->seq_path_nodes = (int32_t*)host_big_alloc(4 * (size_t)std::max<int64_t>(nbases, 1));
        if (!o->seq_path_nodes) return fail(SXG_E_NOMEM, "host allocation of the path array failed");
    }
    // where every block sits: (rank, index in the rank's shard, offsets inside that rank's arrays)
    struct Where { int rank, idx; int64_t n0, e0, c0, s0, b0, gn0, gs0, ge0, gc0, gp0; };
    std::vector<Where> where(std::max(nb, 1));
    for (int r = 0; r < nranks; ++r) {
        const int64_t* c = counts.data() + (size_t)r * BC_N;
        const BlobLayout B = blob_layout(c);
        const uint8_t* blob = blobs[r].data();
        const int32_t *nn = (const int32_t*)(blob + B.nn), *ne = (const int32_t*)(blob + B.ne), *nc = (const int32_t*)(blob + B.nc);
        int64_t n0 = 0, e0 = 0, c0 = 0, s0 = 0, b0 = 0, gn0 = 0, gs0 = 0, ge0 = 0, gc0 = 0, gp0 = 0;
        const int32_t* gcnt = (const int32_t*)(blob + B.bg_counts);
        const int32_t* gnst = (const int32_t*)(blob + B.bg_nsteps);
        const int32_t* rst = (const int32_t*)(blob + B.status);
        for (size_t k = 0; k < parts[r].size(); ++k) {
            const int b = parts[r][k];
            where[b] = Where{r, (int)k, n0, e0, c0, s0, b0, gn0, gs0, ge0, gc0, gp0};
            if (c[BC_BG] && rst[k] == ST_OK) {
                const int32_t* q = gcnt + (size_t)BGC_N * k;
                gn0 += q[BGC_NODES]; gs0 += q[BGC_SEQ]; ge0 += q[BGC_EDGES]; gc0 += in->want_consensus ? q[BGC_CONS] : 0;
                for (int64_t sq = s0; sq < s0 + (in->blk_off[b + 1] - in->blk_off[b]); ++sq) gp0 += gnst[sq];
            }
            o->node_off[b + 1] = nn[k]; o->edge_off[b + 1] = ne[k]; o->cons_off[b + 1] = in->want_consensus ? nc[k] : 0;
            n0 += nn[k]; e0 += ne[k]; c0 += in->want_consensus ? nc[k] : 0;
            s0 += in->blk_off[b + 1] - in->blk_off[b];
            b0 += in->seq_off[in->blk_off[b + 1]] - in->seq_off[in->blk_off[b]];
        }
    }
    for (int b = 0; b < nb; ++b) { o->node_off[b + 1] += o->node_off[b]; o->edge_off[b + 1] += o->edge_off[b]; o->cons_off[b + 1] += o->cons_off[b]; }
    o->node_code.resize((size_t)std::max<int64_t>(o->node_off[nb], 1)); o->node_rank.resize(o->node_code.size()); o->node_group.resize(o->node_code.size());
    o->edge_tail.resize((size_t)std::max<int64_t>(o->edge_off[nb], 1)); o->edge_head.resize(o->edge_tail.size()); o->edge_weight.resize(o->edge_tail.size());
    o->cons_nodes.resize((size_t)std::max<int64_t>(o->cons_off[nb], 1));
    for (int b = 0; b < nb; ++b) {
        const Where& wv = where[b];
        const BlobLayout B = blob_layout(counts.data() + (size_t)wv.rank * BC_N);
        const uint8_t* blob = blobs[wv.rank].data();
        o->status[b] = ((const int32_t*)(blob + B.status))[wv.idx];
        const int64_t nn = o->node_off[b + 1] - o->node_off[b], ne = o->edge_off[b + 1] - o->edge_off[b], nc = o->cons_off[b + 1] - o->cons_off[b];
        memcpy(o->node_code.data() + o->node_off[b], blob + B.code + wv.n0, (size_t)nn);
        memcpy(o->node_rank.data() + o->node_off[b], blob + B.rank + 4 * wv.n0, 4 * (size_t)nn);
        memcpy(o->node_group.data() + o->node_off[b], blob + B.group + 4 * wv.n0, 4 * (size_t)nn);
        memcpy(o->edge_tail.data() + o->edge_off[b], blob + B.et + 4 * wv.e0, 4 * (size_t)ne);
        memcpy(o->edge_head.data() + o->edge_off[b], blob + B.eh + 4 * wv.e0, 4 * (size_t)ne);
        memcpy(o->edge_weight.data() + o->edge_off[b], blob + B.ew + 4 * wv.e0, 4 * (size_t)ne);
        if (nc) memcpy(o->cons_nodes.data() + o->cons_off[b], blob + B.cons + 4 * wv.c0, 4 * (size_t)nc);
        const int64_t nsq = in->blk_off[b + 1] - in->blk_off[b], nbs = in->seq_off[in->blk_off[b + 1]] - in->seq_off[in->blk_off[b]];
        memcpy(o->score.data() + in->blk_off[b], blob + B.score + 4 * wv.s0, 4 * (size_t)nsq);
        memcpy(o->cells.data() + in->blk_off[b], blob + B.cells + 8 * wv.s0, 8 * (size_t)nsq);
        if (with_paths) memcpy(o->seq_path_nodes + in->seq_off[in->blk_off[b]], blob + B.paths + 4 * wv.b0, 4 * (size_t)nbs);
    }
    if (bg_mode) {
        // block graphs in the batch's block order: offsets first, then every block's pieces from its rank's blob
        o->bg_node_off.assign(nb + 1, 0); o->bg_seq_off.assign(nb + 1, 0); o->bg_edge_off.assign(nb + 1, 0); o->bg_cons_off.assign(nb + 1, 0);
        o->bg_step_off.assign((size_t)ns + 1, 0);
        for (int b = 0; b < nb; ++b) {
            const Where& wv = where[b];
            const int64_t* c = counts.data() + (size_t)wv.rank * BC_N;
            const BlobLayout B = blob_layout(c);
            const uint8_t* blob = blobs[wv.rank].data();
            const bool ok = c[BC_BG] && o->status[b] == ST_OK;
            const int32_t* q = (const int32_t*)(blob + B.bg_counts) + (size_t)BGC_N * wv.idx;
            o->bg_node_off[b + 1] = o->bg_node_off[b] + (ok ? q[BGC_NODES] : 0);
            o->bg_seq_off[b + 1] = o->bg_seq_off[b] + (ok ? q[BGC_SEQ] : 0);
            o->bg_edge_off[b + 1] = o->bg_edge_off[b] + (ok ? q[BGC_EDGES] : 0);
            o->bg_cons_off[b + 1] = o->bg_cons_off[b] + (ok && in->want_consensus ? q[BGC_CONS] : 0);
            const int32_t* gnst = (const int32_t*)(blob + B.bg_nsteps) + wv.s0;
            for (int sq = in->blk_off[b]; sq < in->blk_off[b + 1]; ++sq)
                o->bg_step_off[(size_t)sq + 1] = o->bg_step_off[(size_t)sq] + (ok ? gnst[sq - in->blk_off[b]] : 0);
        }
        o->bg_node_len.resize((size_t)std::max<int64_t>(o->bg_node_off[nb], 1)); o->bg_node_outdeg.resize(o->bg_node_len.size());
        o->bg_node_indeg.resize(o->bg_node_len.size()); o->bg_seq.resize((size_t)std::max<int64_t>(o->bg_seq_off[nb], 1));
        o->bg_edge_to.resize((size_t)std::max<int64_t>(o->bg_edge_off[nb], 1)); o->bg_steps.resize((size_t)std::max<int64_t>(o->bg_step_off[(size_t)ns], 1));
        o->bg_cons_steps.resize((size_t)std::max<int64_t>(o->bg_cons_off[nb], 1));
        for (int b = 0; b < nb; ++b) {
            const Where& wv = where[b];
            const BlobLayout B = blob_layout(counts.data() + (size_t)wv.rank * BC_N);
            const uint8_t* blob = blobs[wv.rank].data();
            const int64_t n = o->bg_node_off[b + 1] - o->bg_node_off[b], sb = o->bg_seq_off[b + 1] - o->bg_seq_off[b], ne = o->bg_edge_off[b + 1] - o->bg_edge_off[b];
            const int64_t nc = o->bg_cons_off[b + 1] - o->bg_cons_off[b];
            const int64_t p0 = o->bg_step_off[(size_t)in->blk_off[b]], np = o->bg_step_off[(size_t)in->blk_off[b + 1]] - p0;
            if (n) {
                memcpy(o->bg_node_len.data() + o->bg_node_off[b], blob + B.bg_len + 4 * wv.gn0, 4 * (size_t)n);
                memcpy(o->bg_node_outdeg.data() + o->bg_node_off[b], blob + B.bg_od + 4 * wv.gn0, 4 * (size_t)n);
                memcpy(o->bg_node_indeg.data() + o->bg_node_off[b], blob + B.bg_id + wv.gn0, (size_t)n);
            }
            if (sb) memcpy(o->bg_seq.data() + o->bg_seq_off[b], blob + B.bg_seq + wv.gs0, (size_t)sb);
            if (ne) memcpy(o->bg_edge_to.data() + o->bg_edge_off[b], blob + B.bg_eto + 4 * wv.ge0, 4 * (size_t)ne);
            if (nc) memcpy(o->bg_cons_steps.data() + o->bg_cons_off[b], blob + B.bg_cons + 4 * wv.gc0, 4 * (size_t)nc);
            if (np) memcpy(o->bg_steps.data() + p0, blob + B.bg_steps + 4 * wv.gp0, 4 * (size_t)np);
        }
        out->bg_node_off = o->bg_node_off.data(); out->bg_node_len = o->bg_node_len.data(); out->bg_node_outdeg = o->bg_node_outdeg.data();
        out->bg_node_indeg = o->bg_node_indeg.data(); out->bg_seq_off = o->bg_seq_off.data(); out->bg_seq = (char*)o->bg_seq.data();
        out->bg_edge_off = o->bg_edge_off.data(); out->bg_edge_to = o->bg_edge_to.data(); out->bg_step_off = o->bg_step_off.data();
        out->bg_steps = o->bg_steps.data();
        if (in->want_consensus) { out->bg_cons_off = o->bg_cons_off.data(); out->bg_cons_steps = o->bg_cons_steps.data(); }
    }
    out->status = o->status.data();
    out->node_off = o->node_off.data(); out->edge_off = o->edge_off.data();
    if (with_graphs) {
        out->node_code = o->node_code.data(); out->node_rank = o->node_rank.data(); out->node_group = o->node_group.data();
        out->edge_tail = o->edge_tail.data(); out->edge_head = o->edge_head.data(); out->edge_weight = o->edge_weight.data();
    }
    out->seq_path_nodes = o->seq_path_nodes; out->score = o->score.data(); out->cells = o->cells.data();
    if (in->want_consensus) { out->cons_off = o->cons_off.data(); out->cons_nodes = o->cons_nodes.data(); }
    if (in->want_msa && o->seq_path_nodes) {
        format_msa(o, in, in->want_consensus != 0);
        out->msa_off = o->msa_off.data(); out->msa_cols = o->msa_cols.data(); out->msa = o->msa.data();
    }
    for (int b = 0; b < nb; ++b)
        if (o->status[b] != ST_OK) return fail(SXG_E_BLOCK, "block " + std::to_string(b) + " failed with status " + std::to_string(o->status[b]));
    return SXG_OK;
}
// S8: MSA column = aligned group in rank order; pure formatting of the results (same rule as sxg_poa_batch_download)
void format_msa(OutOwner* o, const sxg_poa_batch_in* in, bool want_consensus) {
    static const char dec[5] = {'A', 'C', 'G', 'T', 'N'};
    const int nb = in->n_blocks;
    o->msa_off.assign(nb + 1, 0); o->msa_cols.assign(std::max(nb, 1), 0);
    std::vector<std::vector<int32_t>> cols(nb);
    for (int b = 0; b < nb; ++b) {
        const int64_t n0 = o->node_off[b];
        const int n = (int)(o->node_off[b + 1] - n0);
        std::vector<int32_t> by_rank(n);
        for (int v = 0; v < n; ++v) by_rank[o->node_rank[n0 + v]] = v;
        cols[b].assign(n, 0);
        int ncol = 0;
        for (int r = 0; r < n; ++r) {
            const int v = by_rank[r];
            if (r > 0 && o->node_group[n0 + by_rank[r - 1]] == o->node_group[n0 + v]) cols[b][v] = ncol - 1; else cols[b][v] = ncol++;
        }
        o->msa_cols[b] = ncol;
        const int rows = (in->blk_off[b + 1] - in->blk_off[b]) + (want_consensus ? 1 : 0);
        o->msa_off[b + 1] = o->msa_off[b] + (o->status[b] == ST_OK ? (int64_t)rows * ncol : 0);
    }
    o->msa.assign((size_t)std::max<int64_t>(o->msa_off[nb], 1), '-');
    for (int b = 0; b < nb; ++b) {
        if (o->status[b] != ST_OK) continue;
        const int64_t n0 = o->node_off[b];
        const int ncol = o->msa_cols[b];
        char* base = o->msa.data() + o->msa_off[b];
        int row = 0;
        for (int sq = in->blk_off[b]; sq < in->blk_off[b + 1]; ++sq, ++row)
            for (int64_t k = in->seq_off[sq]; k < in->seq_off[sq + 1]; ++k) base[(int64_t)row * ncol + cols[b][o->seq_path_nodes[k]]] = dec[o->node_code[n0 + o->seq_path_nodes[k]]];
        if (want_consensus)
            for (int64_t k = o->cons_off[b]; k < o->cons_off[b + 1]; ++k) base[(int64_t)row * ncol + cols[b][o->cons_nodes[k]]] = dec[o->node_code[n0 + o->cons_nodes[k]]];
    }
}
int run_shard(sxg_poa_handle* h, const sxg_poa_batch_in* in, const std::vector<int32_t>& part, int64_t* counts) {
    LocalBatch L;
    build_local(in, part, L);
    int rc = sxg_poa_batch_upload(h, &L.in);
    if (rc) return rc;
    rc = sxg_poa_batch_execute(h);
    if (rc && rc != SXG_E_BLOCK) return rc;   // (per-block failures travel in the status array)
    return pack_blob(h, counts);
}
}  // namespace

// Waiting for a collective with a deadline: a peer that died (or never called) would otherwise park this rank in
// hipStreamSynchronize for ever.  The stream is polled; after SXG_POA_COMM_TIMEOUT_S seconds (default 600) or on an
// asynchronous RCCL error the communicator is aborted and the call returns SXG_E_NODEVICE.
static void comm_abort(sxg_poa_handle* h) {
    if (h->comm && h->own_comm) (void)ncclCommAbort(h->comm);   // (an attached communicator stays the caller's to abort)
    h->comm = nullptr; h->own_comm = false; h->nranks = 1; h->rank = 0;
}
static int comm_wait(sxg_poa_handle* h, const char* what) {
    static const double limit = [] { const char* e = getenv("SXG_POA_COMM_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return v > 0 ? v : 600.0; }();
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        const hipError_t q = hipStreamQuery(h->stream);
        if (q == hipSuccess) return SXG_OK;
        if (q != hipErrorNotReady) { comm_abort(h); return fail(SXG_E_NODEVICE, std::string(what) + ": " + hipGetErrorString(q)); }
        ncclResult_t ar = ncclSuccess;
        if (h->comm && (spin & 255) == 255 && (ncclCommGetAsyncError(h->comm, &ar) != ncclSuccess || (ar != ncclSuccess && ar != ncclInProgress))) {
            comm_abort(h);
            return fail(SXG_E_NODEVICE, std::string(what) + ": RCCL reported " + ncclGetErrorString(ar));
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            comm_abort(h);
            return fail(SXG_E_NODEVICE, std::string(what) + ": no completion after " + std::to_string((int)limit) + " s (a peer rank failed or never called); communicator aborted");
        }
        if (spin > 1000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

static int check_batch(const sxg_poa_batch_in* in) {
    if (in->n_blocks < 0 || (in->n_blocks > 0 && (!in->blk_off || !in->seq_off || !in->params || !in->bases))) return fail(SXG_E_INVALID, "batch_in has NULL arrays");
    return SXG_OK;
}

// The sharded run in three stages, like upload / execute / download of the single-GPU call:
//   sxg_poa_batch_upload_sharded   deals the blocks of the SAME batch on every rank (LPT) and uploads this rank's share;
//   sxg_poa_batch_execute_sharded  aligns the uploaded share (whatever uploaded it: the deal above, or a plain
//                                  sxg_poa_batch_upload of a share the caller cut himself), packs the results into one
//                                  device blob and brings every blob to rank 0 -- the collective part;
//   sxg_poa_batch_download_sharded rank 0: results of all ranks in the batch's block order; others: SXG_NOT_ROOT.
extern "C" int sxg_poa_batch_upload_sharded(sxg_poa_handle* h, const sxg_poa_batch_in* in) {
    if (!h || !in) return fail(SXG_E_INVALID, "NULL argument");
    int rc = check_batch(in);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    const int nranks = h->comm ? h->nranks : 1, rank = h->comm ? h->rank : 0;
    lpt_partition(in, nranks, h->sh_parts);
    h->sh_dealt = false; h->sh_exchanged = false;
    LocalBatch L;
    build_local(in, h->sh_parts[rank], L);
    if ((rc = sxg_poa_batch_upload(h, &L.in))) return rc;
    h->sh_dealt = true;
    return SXG_OK;
}

extern "C" int sxg_poa_batch_execute_sharded(sxg_poa_handle* h) {
    if (!h) return fail(SXG_E_INVALID, "handle is NULL");
    HIPCHK(hipSetDevice(h->device));
    const int nranks = h->comm ? h->nranks : 1, rank = h->comm ? h->rank : 0;
    h->sh_exchanged = false; h->sh_bytes_received = 0; h->sh_ranks_seen = 0;
    std::vector<int64_t>& counts = h->sh_counts;
    counts.assign((size_t)nranks * BC_N, 0);
    int64_t* mine = counts.data() + (size_t)rank * BC_N;
    int rc = h->have_batch ? sxg_poa_batch_execute(h) : fail(SXG_E_INVALID, "no batch uploaded");
    const auto t_pack = std::chrono::steady_clock::now();
    { const RoctxRange range_("sxg_poa sharded: pack"); if (!rc || rc == SXG_E_BLOCK) rc = pack_blob(h, mine); }   // (per-block failures travel in the status array)
    const auto t_xch = std::chrono::steady_clock::now();
    const RoctxRange range_("sxg_poa sharded: exchange");
    h->sh_pack_ms = std::chrono::duration<double, std::milli>(t_xch - t_pack).count();
    h->sh_exchange_ms = 0;
    if (!h->comm) {
        if (rc) return rc;
        h->sh_at.assign(2, 0); h->sh_exchanged = true; h->sh_ranks_seen = 1;
        return SXG_OK;
    }
    {   // (a communicator of ONE rank takes the same way: the collectives run, nothing is sent)
        // A rank-local failure (allocation, HIP error) must not leave the peers blocked in a collective: the failing rank
        // still takes part in the size exchange, with its error code in BC_PAD and nothing to send, and EVERY rank then
        // returns that failure -- the ranks fail together.  (d_counts is allocated by sxg_poa_comm_init / _attach.)
        std::string local_err = rc ? std::string(sxg_poa_last_error()) : std::string();
        if (rc) { memset(mine, 0, sizeof(int64_t) * BC_N); mine[BC_PAD] = rc; }
        int64_t* dc = h->d_counts.as<int64_t>();
        if (!dc) return fail(SXG_E_INVALID, "sharded run without sxg_poa_comm_init / sxg_poa_comm_attach");
        auto gathered_failure = [&](const char* what) -> int {
            for (int r = 0; r < nranks; ++r)
                if (counts[(size_t)r * BC_N + BC_PAD]) {
                    const int code = (int)counts[(size_t)r * BC_N + BC_PAD];
                    return fail(code, std::string("sharded run: rank ") + std::to_string(r) + " failed " + what + " (code " + std::to_string(code) + ")" +
                                      (r == rank && !local_err.empty() ? ": " + local_err : std::string()));
                }
            return SXG_OK;
        };
        // sizes first (RCCL has no all-gather-v) ...
        std::vector<int64_t> my_counts(mine, mine + BC_N);
        HIPCHK(hipMemcpyAsync(dc + (size_t)nranks * BC_N, my_counts.data(), 8 * BC_N, hipMemcpyHostToDevice, h->stream));
        NCCLCHK(ncclAllGather(dc + (size_t)nranks * BC_N, dc, BC_N, ncclInt64, h->comm, h->stream));
        HIPCHK(hipMemcpyAsync(counts.data(), dc, 8 * (size_t)BC_N * nranks, hipMemcpyDeviceToHost, h->stream));
        if ((rc = comm_wait(h, "size all-gather"))) return rc;
        if ((rc = gathered_failure("while aligning its share"))) return rc;
        // ... the root makes room (the one step after the size exchange that can fail on one rank only: its outcome is
        // all-gathered too, one word per rank, before anybody posts a send) ...
        std::vector<size_t>& at = h->sh_at;
        at.assign(nranks + 1, 0);
        for (int r = 1; r < nranks; ++r) at[r + 1] = at[r] + (((size_t)counts[(size_t)r * BC_N + BC_BYTES] + 255) & ~(size_t)255);
        int64_t ready = 0;
        if (rank == 0 && (rc = h->d_recv.ensure(at[nranks] + 256))) { ready = rc; local_err = sxg_poa_last_error(); }
        std::vector<int64_t> readies((size_t)nranks, 0);
        HIPCHK(hipMemcpyAsync(dc + (size_t)nranks * BC_N, &ready, 8, hipMemcpyHostToDevice, h->stream));
        NCCLCHK(ncclAllGather(dc + (size_t)nranks * BC_N, dc, 1, ncclInt64, h->comm, h->stream));
        HIPCHK(hipMemcpyAsync(readies.data(), dc, 8 * (size_t)nranks, hipMemcpyDeviceToHost, h->stream));
        if ((rc = comm_wait(h, "receive-buffer all-gather"))) return rc;
        for (int r = 0; r < nranks; ++r) counts[(size_t)r * BC_N + BC_PAD] = readies[r];
        if ((rc = gathered_failure("to allocate the receive buffer"))) return rc;
        // ... then every blob straight to the root, exact sizes, one group: the seven peers use their own xGMI links.
        // ncclGroupStart is ALWAYS paired with ncclGroupEnd: an error inside the group is reported after the group is closed.
        ncclResult_t gr = ncclGroupStart(), g2;
        if (gr == ncclSuccess) {
            if (rank == 0) {
                for (int r = 1; r < nranks && gr == ncclSuccess; ++r)
                    if (counts[(size_t)r * BC_N + BC_BYTES] > 0)
                        gr = ncclRecv(h->d_recv.as<uint8_t>() + at[r], (size_t)counts[(size_t)r * BC_N + BC_BYTES], ncclUint8, r, h->comm, h->stream);
            } else if (mine[BC_BYTES] > 0)
                gr = ncclSend(h->d_blob.p, (size_t)mine[BC_BYTES], ncclUint8, 0, h->comm, h->stream);
            g2 = ncclGroupEnd();
            if (gr == ncclSuccess) gr = g2;
        }
        if (gr != ncclSuccess) { comm_abort(h); return fail(SXG_E_NODEVICE, std::string("blob exchange: ") + ncclGetErrorString(gr)); }
        if ((rc = comm_wait(h, "blob exchange"))) return rc;
        for (int r = 0; r < nranks; ++r) {
            h->sh_ranks_seen += 1;   // (every rank's counts arrived; a rank with an empty share sends nothing)
            if (r != 0 && rank == 0) h->sh_bytes_received += (uint64_t)counts[(size_t)r * BC_N + BC_BYTES];
        }
    }
    h->sh_exchange_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_xch).count();
    h->sh_exchanged = true;
    return SXG_OK;
}

extern "C" int sxg_poa_sharded_timing(sxg_poa_handle* h, double* pack_ms, double* exchange_ms) {
    if (!h) return fail(SXG_E_INVALID, "handle is NULL");
    if (pack_ms) *pack_ms = h->sh_pack_ms;
    if (exchange_ms) *exchange_ms = h->sh_exchange_ms;
    return SXG_OK;
}

extern "C" int sxg_poa_batch_download_sharded(sxg_poa_handle* h, const sxg_poa_batch_in* in, sxg_poa_batch_out* out) {
    if (!h || !in || !out) return fail(SXG_E_INVALID, "NULL argument");
    memset(out, 0, sizeof(*out));
    if (!h->sh_exchanged) return fail(SXG_E_INVALID, "no sharded batch executed");
    if (!h->sh_dealt) return fail(SXG_E_INVALID, "the batch was not dealt by sxg_poa_batch_upload_sharded: the root cannot know its block order");
    HIPCHK(hipSetDevice(h->device));
    const int nranks = h->comm ? h->nranks : 1, rank = h->comm ? h->rank : 0;
    if (rank != 0) return SXG_NOT_ROOT;
    if ((int)h->sh_parts.size() != nranks) return fail(SXG_E_INVALID, "communicator changed since the upload");
    std::vector<std::vector<uint8_t>> blobs(nranks);
    for (int r = 0; r < nranks; ++r) {
        const size_t bytes = (size_t)h->sh_counts[(size_t)r * BC_N + BC_BYTES];
        blobs[r].resize(std::max<size_t>(bytes, 16));
        if (bytes) HIPCHK(hipMemcpy(blobs[r].data(), r == 0 ? h->d_blob.p : (void*)(h->d_recv.as<uint8_t>() + h->sh_at[r]), bytes, hipMemcpyDeviceToHost));
    }
    int rc = assemble(in, h->sh_parts, blobs, h->sh_counts, out);
    if (rc && rc != SXG_E_BLOCK) sxg_poa_batch_free(out);
    return rc;
}

extern "C" int sxg_poa_sharded_info(sxg_poa_handle* h, int32_t* ranks_seen, uint64_t* bytes_received) {
    if (!h) return fail(SXG_E_INVALID, "handle is NULL");
    if (ranks_seen) *ranks_seen = h->sh_ranks_seen;
    if (bytes_received) *bytes_received = h->sh_bytes_received;
    return SXG_OK;
}

extern "C" int sxg_poa_batch_run_sharded(sxg_poa_handle* h, const sxg_poa_batch_in* in, sxg_poa_batch_out* out) {
    if (!h || !in || !out) return fail(SXG_E_INVALID, "NULL argument");
    memset(out, 0, sizeof(*out));
    // (an upload that fails on one rank only -- a host or device allocation -- must still reach the collective: the
    //  execute stage then reports "no batch" as this rank's error and all ranks fail together)
    const int urc = sxg_poa_batch_upload_sharded(h, in);
    const std::string uerr = urc ? std::string(sxg_poa_last_error()) : std::string();
    if (urc && !(h->comm && h->nranks > 1)) return urc;
    if (urc) h->have_batch = false;
    int rc = sxg_poa_batch_execute_sharded(h);
    if (rc) return urc ? fail(urc, uerr) : rc;
    return sxg_poa_batch_download_sharded(h, in, out);
}

// TEST ENTRY: the sharded run with `nranks` SIMULATED ranks on this one GPU -- every shard is aligned here, one after
// the other, and its blob is put where RCCL would have delivered it.  Exercises the partition, the blob packing
// and the root's assembly with more than one rank on a box that has one GPU; only ncclSend/ncclRecv are not taken.
extern "C" int sxg_poa_batch_run_sharded_local(sxg_poa_handle* h, const sxg_poa_batch_in* in, int nranks, sxg_poa_batch_out* out) {
    if (!h || !in || !out || nranks < 1) return fail(SXG_E_INVALID, "bad argument");
    memset(out, 0, sizeof(*out));
    int rc = check_batch(in);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    std::vector<std::vector<int32_t>> parts;
    lpt_partition(in, nranks, parts);
    std::vector<int64_t> counts((size_t)nranks * BC_N, 0);
    std::vector<std::vector<uint8_t>> blobs(nranks);
    for (int r = 0; r < nranks; ++r) {
        if ((rc = run_shard(h, in, parts[r], counts.data() + (size_t)r * BC_N))) return rc;
        const size_t bytes = (size_t)counts[(size_t)r * BC_N + BC_BYTES];
        blobs[r].resize(std::max<size_t>(bytes, 16));
        if (bytes) HIPCHK(hipMemcpy(blobs[r].data(), h->d_blob.p, bytes, hipMemcpyDeviceToHost));
    }
    rc = assemble(in, parts, blobs, counts, out);
    if (rc && rc != SXG_E_BLOCK) sxg_poa_batch_free(out);
    return rc;
}

// ---------------------------------------------------------------------------------------
struct AlignOwner {
    std::vector<int32_t> status, score, pair_row, pair_pos;
    std::vector<int64_t> pair_off;
};

extern "C" void sxg_poa_align_free(sxg_poa_align_out* out) {
    if (!out) return;
    delete (AlignOwner*)out->_owner;
    memset(out, 0, sizeof(*out));
}

extern "C" int sxg_poa_align_batch(sxg_poa_handle* h, const sxg_poa_align_in* in, sxg_poa_align_out* out) {
    if (!h || !in || !out) return fail(SXG_E_INVALID, "NULL argument");
    memset(out, 0, sizeof(*out));
    const int n = in->n;
    if (n < 0 || (n > 0 && (!in->row_off || !in->pred_off || !in->seq_off || !in->params)))
        return fail(SXG_E_INVALID, "align_in has NULL arrays");
    HIPCHK(hipSetDevice(h->device));
    h->have_batch = false; h->executed = false;
    if (n > 0 && (in->row_off[0] != 0 || in->seq_off[0] != 0)) return fail(SXG_E_INVALID, "row_off[0] and seq_off[0] must be 0");
    const int64_t rows = n ? in->row_off[n] : 0, nbases = n ? in->seq_off[n] : 0;
    if (rows < 0 || nbases < 0) return fail(SXG_E_INVALID, "negative totals");
    if (rows > 0 && (!in->row_code || !in->row_sink)) return fail(SXG_E_INVALID, "row_code / row_sink is NULL");
    if (nbases > 0 && !in->bases) return fail(SXG_E_INVALID, "bases is NULL");
    if (rows > 0 && in->pred_off[0] != 0) return fail(SXG_E_INVALID, "pred_off[0] must be 0");
    for (int64_t r = 0; r < rows; ++r)
        if (in->pred_off[r + 1] < in->pred_off[r]) return fail(SXG_E_INVALID, "pred_off not monotone");
    const int64_t ne = rows ? in->pred_off[rows] : 0;
    if (ne > 0 && !in->preds) return fail(SXG_E_INVALID, "preds is NULL");
    // validate topology: every predecessor is an earlier row of the same problem
    for (int p = 0; p < n; ++p) {
        const int64_t r0 = in->row_off[p], r1 = in->row_off[p + 1];
        if (r1 < r0 || in->seq_off[p + 1] < in->seq_off[p]) return fail(SXG_E_INVALID, "offsets not monotone");
        if (r1 - r0 >= (1 << 20)) return fail(SXG_E_INVALID, "graph too large");
        for (int64_t r = r0; r < r1; ++r)
            for (int64_t k = in->pred_off[r]; k < in->pred_off[r + 1]; ++k)
                if (in->preds[k] < 1 || in->preds[k] > r - r0) return fail(SXG_E_INVALID, "predecessor rows must precede their successor");
    }
    AlignOwner* o = new AlignOwner();
    out->_owner = o; out->n = n;
    o->status.assign(std::max(n, 1), 0); o->score.assign(std::max(n, 1), 0); o->pair_off.assign(n + 1, 0);
    DevBuf d_row_off, d_code, d_sink, d_pred_off, d_preds, d_seq_off, d_bases, d_params, d_status, d_score, d_np, d_pr, d_pp, d_work,
        d_queue, d_arena;
    auto cleanup = [&]() {
        DevBuf* bs[] = {&d_row_off, &d_code, &d_sink, &d_pred_off, &d_preds, &d_seq_off, &d_bases, &d_params, &d_status, &d_score,
                        &d_np, &d_pr, &d_pp, &d_work, &d_queue, &d_arena};
        for (DevBuf* b : bs) b->release();
    };
#define CK(x) do { int _rc = (x); if (_rc) { cleanup(); sxg_poa_align_free(out); return _rc; } } while (0)
#define HCK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { cleanup(); sxg_poa_align_free(out); return fail(SXG_E_NODEVICE, std::string(#x) + ": " + hipGetErrorString(_e)); } } while (0)
    const int np = in->per_problem_params ? n : 1;
    std::vector<uint8_t> stage((size_t)std::max<int64_t>(nbases, 1));
    for (int64_t i = 0; i < nbases; ++i) stage[i] = in->bases[i] > 4 ? 4 : in->bases[i];
    std::vector<uint8_t> codes((size_t)std::max<int64_t>(rows, 1));
    for (int64_t i = 0; i < rows; ++i) codes[i] = in->row_code[i] > 4 ? 4 : in->row_code[i];
    const size_t outcap = (size_t)(rows + nbases + 1);
    CK(d_row_off.ensure(8 * (size_t)(n + 1))); CK(d_code.ensure((size_t)rows + 16)); CK(d_sink.ensure((size_t)rows + 16));
    CK(d_pred_off.ensure(8 * (size_t)(rows + 1))); CK(d_preds.ensure(4 * (size_t)(ne + 1))); CK(d_seq_off.ensure(8 * (size_t)(n + 1)));
    CK(d_bases.ensure((size_t)nbases + 16)); CK(d_params.ensure(sizeof(sxg_poa_params) * (size_t)std::max(np, 1)));
    CK(d_status.ensure(4 * (size_t)std::max(n, 1))); CK(d_score.ensure(4 * (size_t)std::max(n, 1))); CK(d_np.ensure(4 * (size_t)std::max(n, 1)));
    CK(d_pr.ensure(4 * outcap)); CK(d_pp.ensure(4 * outcap)); CK(d_work.ensure(4 * (size_t)std::max(n, 1))); CK(d_queue.ensure(256));
    if (n) {
        HCK(hipMemcpy(d_row_off.p, in->row_off, 8 * (size_t)(n + 1), hipMemcpyHostToDevice));
        HCK(hipMemcpy(d_seq_off.p, in->seq_off, 8 * (size_t)(n + 1), hipMemcpyHostToDevice));
        HCK(hipMemcpy(d_pred_off.p, in->pred_off, 8 * (size_t)(rows + 1), hipMemcpyHostToDevice));
        HCK(hipMemcpy(d_params.p, in->params, sizeof(sxg_poa_params) * (size_t)np, hipMemcpyHostToDevice));
        if (rows) {
            HCK(hipMemcpy(d_code.p, codes.data(), (size_t)rows, hipMemcpyHostToDevice));
            HCK(hipMemcpy(d_sink.p, in->row_sink, (size_t)rows, hipMemcpyHostToDevice));
        }
        if (ne) HCK(hipMemcpy(d_preds.p, in->preds, 4 * (size_t)ne, hipMemcpyHostToDevice));
        if (nbases) HCK(hipMemcpy(d_bases.p, stage.data(), (size_t)nbases, hipMemcpyHostToDevice));
        HCK(hipMemset(d_status.p, 0, 4 * (size_t)n)); HCK(hipMemset(d_score.p, 0, 4 * (size_t)n)); HCK(hipMemset(d_np.p, 0, 4 * (size_t)n));

    }
    // plans
    struct APlan { Variant variant; bool cvx, sw; std::vector<int32_t> work; int rows_cap = 0; };
    std::vector<APlan> plans;
    for (int p = 0; p < n; ++p) {
        const Scoring S = normalise(in->params[in->per_problem_params ? p : 0]);
        const int len = (int)(in->seq_off[p + 1] - in->seq_off[p]), N = (int)(in->row_off[p + 1] - in->row_off[p]);
        Variant v;
        if (!variant_for_len(len, row_mode(S, len, N), &v, S.sw)) { o->status[p] = ST_TOO_LONG; continue; }
        APlan* pl = nullptr;
        for (auto& q : plans)
            if (q.variant.W == v.W && q.variant.NW == v.NW && q.variant.RM == v.RM && q.cvx == (bool)S.convex && q.sw == (bool)S.sw) { pl = &q; break; }
        if (!pl) { plans.push_back(APlan{v, (bool)S.convex, (bool)S.sw, {}, 0}); pl = &plans.back(); }
        pl->work.push_back(p);
        pl->rows_cap = std::max(pl->rows_cap, N);
    }
    if (n) HCK(hipMemcpy(d_status.p, o->status.data(), 4 * (size_t)n, hipMemcpyHostToDevice));
    for (auto& pl : plans) {
        const Variant V = pl.variant;
        const int rows_cap = pl.rows_cap + 1;
        int64_t maxe = 0;
        for (int p : pl.work) maxe = std::max<int64_t>(maxe, in->pred_off[in->row_off[p + 1]] - in->pred_off[in->row_off[p]]);
        // r_preds is sized by nodes_cap in make_layout: give it the edge count
        const int wb = V.RM == 1 ? 8 : 4;
        SlotLayout lay2 = make_layout((int)std::max<int64_t>(maxe + 8, rows_cap + 8), rows_cap, rows_cap + 1,
                                      (int)maxe + 8, V.T(), V.Lpad(), wb, true, V.RM == 2 ? 2 * V.T() : 0);  // every strip kept
        if (V.RM == 2) lay2.lds_rows = p16_lds_rows(V.T(), V.W);
        auto kern = align_kernel(pl.variant, pl.cvx, pl.sw);
        int per_cu = 1;
        int smem = V.RM == 2 ? dp16_lds_bytes(V.T(), V.W, lay2.lds_rows) : dp_lds_launch_bytes(V.Lpad(), wb);
        const int pf_off = (V.RM != 2 && getenv("SXG_POA_PREFETCH")) ? dp_pf_offset(V.Lpad(), wb, V.T()) : -1;
        if (pf_off >= 0) smem += dp_pf_bytes(V.Lpad(), wb, V.T());
        if (smem > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, V.T(), (size_t)smem) != hipSuccess || per_cu < 1) per_cu = 1;
        const uint64_t budget = arena_budget(h);
        int64_t n_slots = std::min<int64_t>((int64_t)pl.work.size(), (int64_t)h->num_cu * per_cu);
        n_slots = std::min<int64_t>(n_slots, (int64_t)(budget / lay2.total));
        if (n_slots < 1) { cleanup(); sxg_poa_align_free(out); return fail(SXG_E_NOMEM, "memory budget too small for one alignment arena"); }
        CK(d_arena.ensure((size_t)n_slots * lay2.total));
        HCK(hipMemcpy(d_work.p, pl.work.data(), 4 * pl.work.size(), hipMemcpyHostToDevice));
        HCK(hipMemset(d_queue.p, 0, 4));
        AlignArgs A;
        A.row_off = d_row_off.as<int64_t>(); A.row_code = d_code.as<uint8_t>(); A.row_sink = d_sink.as<uint8_t>();
        A.pred_off = d_pred_off.as<int64_t>(); A.preds = d_preds.as<int32_t>(); A.seq_off = d_seq_off.as<int64_t>();
        A.bases = d_bases.as<uint8_t>(); A.params = d_params.as<sxg_poa_params>(); A.per_problem_params = in->per_problem_params;
        A.work = d_work.as<int32_t>(); A.n_work = (int)pl.work.size(); A.queue = d_queue.as<int32_t>();
        A.arena = d_arena.as<uint8_t>(); A.lay = lay2;
        A.status = d_status.as<int32_t>(); A.score = d_score.as<int32_t>(); A.n_pairs = d_np.as<int32_t>();
        A.pair_row = d_pr.as<int32_t>(); A.pair_pos = d_pp.as<int32_t>();
        A.park_in_lds = (V.RM == 2 || dp_park_in_lds(V.Lpad(), wb)) ? 1 : 0;
        A.pf_off = pf_off;
        A.num_cu = std::max(h->num_cu, 1);
        hipLaunchKernelGGL(kern, dim3((unsigned)n_slots), dim3(V.T()), (size_t)smem, h->stream, A);
        HCK(hipGetLastError());
        HCK(hipStreamSynchronize(h->stream));
    }
    std::vector<int32_t> npairs(std::max(n, 1), 0), pr(outcap), pp(outcap);
    if (n) {
        HCK(hipMemcpy(o->status.data(), d_status.p, 4 * (size_t)n, hipMemcpyDeviceToHost));
        HCK(hipMemcpy(o->score.data(), d_score.p, 4 * (size_t)n, hipMemcpyDeviceToHost));
        HCK(hipMemcpy(npairs.data(), d_np.p, 4 * (size_t)n, hipMemcpyDeviceToHost));
        HCK(hipMemcpy(pr.data(), d_pr.p, 4 * outcap, hipMemcpyDeviceToHost));
        HCK(hipMemcpy(pp.data(), d_pp.p, 4 * outcap, hipMemcpyDeviceToHost));
    }
    for (int p = 0; p < n; ++p) o->pair_off[p + 1] = o->pair_off[p] + npairs[p];
    o->pair_row.resize((size_t)std::max<int64_t>(o->pair_off[n], 1)); o->pair_pos.resize(o->pair_row.size());
    for (int p = 0; p < n; ++p) {
        const int64_t s0 = in->row_off[p] + in->seq_off[p];
        std::copy(pr.begin() + s0, pr.begin() + s0 + npairs[p], o->pair_row.begin() + o->pair_off[p]);
        std::copy(pp.begin() + s0, pp.begin() + s0 + npairs[p], o->pair_pos.begin() + o->pair_off[p]);
    }
    cleanup();
#undef CK
#undef HCK
    out->status = o->status.data(); out->score = o->score.data(); out->pair_off = o->pair_off.data();
    out->pair_row = o->pair_row.data(); out->pair_pos = o->pair_pos.data();
    for (int p = 0; p < n; ++p)
        if (o->status[p] != ST_OK) return fail(SXG_E_BLOCK, "problem " + std::to_string(p) + " failed with status " + std::to_string(o->status[p]));
    return SXG_OK;
}

// ---------------------------------------------------------------------------------------
// XXH64 (published xxHash specification); dedup key of src/smooth.cpp:716.
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
extern "C" uint64_t sxg_xxh64(const void* data, uint64_t len, uint64_t seed) {
    const uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL,
                   P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
    auto rd64 = [](const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; };
    auto rd32 = [](const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; };
    auto round = [&](uint64_t acc, uint64_t in) { return rotl64(acc + in * P2, 31) * P1; };
    auto merge = [&](uint64_t hh, uint64_t v) { return (hh ^ round(0, v)) * P1 + P4; };
    const uint8_t *p = (const uint8_t*)data, *end = p + len;
    uint64_t hh;
    if (len >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t* lim = end - 32;
        do {
            v1 = round(v1, rd64(p)); v2 = round(v2, rd64(p + 8)); v3 = round(v3, rd64(p + 16)); v4 = round(v4, rd64(p + 24));
            p += 32;
        } while (p <= lim);
        hh = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        hh = merge(hh, v1); hh = merge(hh, v2); hh = merge(hh, v3); hh = merge(hh, v4);
    } else hh = seed + P5;
    hh += len;
    while (p + 8 <= end) { hh ^= round(0, rd64(p)); hh = rotl64(hh, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { hh ^= (uint64_t)rd32(p) * P1; hh = rotl64(hh, 23) * P2 + P3; p += 4; }
    while (p < end) { hh ^= (*p) * P5; hh = rotl64(hh, 11) * P1; ++p; }
    hh ^= hh >> 33; hh *= P2; hh ^= hh >> 29; hh *= P3; hh ^= hh >> 32;
    return hh;
}
