// poa_kernels.hip.h -- the persistent kernels of the engine and what they share with the host side (slot layout, launch
// arguments): one workgroup ("slot") owns one smoothxg block at a time and runs the whole sequential chain of
// src/smooth.cpp:760-769 on the device.  Included by sxg_poa.hip (host side of the C ABI) and by the kern_*.hip translation
// units that instantiate the kernel classes (compiled in parallel by smoothxg_amd/build.py).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>

#include "../../include/sxg_poa.h"
#include "poa_dp.hip.h"
#include "poa_dp16.hip.h"
#include "poa_band16.hip.h"
#include "poa_graph_dev.h"
#include "poa_bgraph_dev.h"

using namespace sxg;

// =======================================================================================
// device side
// =======================================================================================

struct SlotLayout {  // byte offsets inside one slot arena (all 16-byte aligned)
    size_t hdr, code, rank, order, order_tmp, leader, gmem, in_head, in_tail, out_head, out_tail, in_deg,
        out_deg, e_tail, e_head, e_next_in, e_next_out, e_w, posnode, target, newidx, nexta, preva, slotadd,
        kind, xpos, via, dfs_stack, dfs_rec, sp_rank, sp_first, sp_cnt, r_code, r_flags, r_pred_off, r_preds, r_slot, r_tbx, r_sseq, r_row_node, r_meta, tb, steps, pool,
        row0, park, cons_sc, cons_pr, pair_row, pair_pos, total;
    int nodes_cap, rows_cap, pool_slots, step_cap, scratch_len, Lpad, word_bytes, threads;
    int band_strips;  // packed sweep: strips per row in the traceback plane (0 = not the packed sweep)
    int lds_rows;     // packed sweep: stored rows the workgroup's LDS holds on chip (set by prepare_plan)
};

inline size_t lay(size_t& cur, size_t bytes) {
    size_t o = cur;
    cur += (bytes + 255) & ~(size_t)255;
    return o;
}

// band_strips > 0: packed sweep with strips of Lpad / (2 * threads) columns; cell_bytes: its plane cell format (2: delta
// codes, W + 1 halfwords per strip; 4: one dword per cell)
// spoa_scratch: the launch holds blocks that ask for spoa's depth-first order (S7'): stacks, records and the kept state of the
// re-sort, ~104 bytes per node of nodes_cap -- left out of every other arena
inline SlotLayout make_layout(int nodes_cap, int rows_cap, int pool_slots, int step_cap, int threads, int Lpad,
                              int word_bytes, bool pairs, int band_strips = 0, int cell_bytes = 4, bool spoa_scratch = false) {
    SlotLayout L;
    memset(&L, 0, sizeof(L));
    L.nodes_cap = nodes_cap; L.rows_cap = rows_cap; L.pool_slots = pool_slots;
    L.step_cap = step_cap; L.threads = threads; L.Lpad = Lpad; L.word_bytes = word_bytes;
    L.band_strips = band_strips;
    const bool packed = band_strips > 0;
    const size_t C = (size_t)nodes_cap + 4, S = (size_t)std::max(nodes_cap, Lpad) + 4, Rr = (size_t)rows_cap + 4;
    L.scratch_len = (int)S;
    size_t cur = 0;
    L.hdr = lay(cur, 512);
    L.code = lay(cur, C);
    L.rank = lay(cur, 4 * C); L.order = lay(cur, 4 * C); L.order_tmp = lay(cur, 4 * C); L.leader = lay(cur, 4 * C);
    L.gmem = lay(cur, 20 * C);
    L.in_head = lay(cur, 4 * C); L.in_tail = lay(cur, 4 * C); L.out_head = lay(cur, 4 * C);
    L.out_tail = lay(cur, 4 * C); L.in_deg = lay(cur, 4 * C); L.out_deg = lay(cur, 4 * C);
    L.e_tail = lay(cur, 4 * C); L.e_head = lay(cur, 4 * C); L.e_next_in = lay(cur, 4 * C);
    L.e_next_out = lay(cur, 4 * C); L.e_w = lay(cur, 4 * C);
    L.posnode = lay(cur, 4 * S); L.target = lay(cur, 4 * S); L.newidx = lay(cur, 4 * S);
    L.nexta = lay(cur, 4 * S); L.preva = lay(cur, 4 * S); L.slotadd = lay(cur, 4 * S); L.kind = lay(cur, S);
    L.xpos = lay(cur, 4 * C);
    L.via = lay(cur, 4 * C);
    L.dfs_stack = lay(cur, spoa_scratch ? 4 * (7 * C + 8) : 256);
    L.dfs_rec = lay(cur, spoa_scratch ? 64 * C : 256);
    L.sp_rank = lay(cur, spoa_scratch ? 4 * C : 256); L.sp_first = lay(cur, spoa_scratch ? 4 * C : 256); L.sp_cnt = lay(cur, spoa_scratch ? 4 * C : 256);
    L.r_code = lay(cur, Rr); L.r_flags = lay(cur, Rr); L.r_pred_off = lay(cur, 4 * Rr);
    L.r_preds = lay(cur, 4 * C); L.r_slot = lay(cur, 4 * Rr); L.r_tbx = lay(cur, 4 * Rr);
    L.r_sseq = lay(cur, 4 * Rr); L.r_row_node = lay(cur, 4 * Rr); L.r_meta = lay(cur, 32 * Rr);
    // traceback plane: one byte per cell, or (packed sweep) one dword per cell of the row's band of strips
    L.tb = lay(cur, ((size_t)rows_cap + 1) * (packed ? (size_t)band_strips * (size_t)p16_slot_dwords(Lpad / (2 * threads), cell_bytes) * 4 : (size_t)Lpad));
    L.steps = lay(cur, packed ? 256 : (size_t)std::max(step_cap, 1) * 3 * threads * 4);
    // (a stored row of the packed sweep ends with one more word per lane: the column left of the lane's strips)
    L.pool = lay(cur, (size_t)pool_slots * ((size_t)Lpad * word_bytes + (size_t)threads * 4));
    L.row0 = lay(cur, (size_t)Lpad * word_bytes + (size_t)threads * 4);
    L.park = lay(cur, (size_t)Lpad * word_bytes);
    L.cons_sc = lay(cur, 8 * C); L.cons_pr = lay(cur, 4 * C);
    if (pairs) { L.pair_row = lay(cur, 4 * (Rr + Lpad)); L.pair_pos = lay(cur, 4 * (Rr + Lpad)); }
    L.total = cur;
    return L;
}

struct SlotViews {
    GraphView G;
    RowsView R;
    DpBuffers B;
    int64_t* cons_sc;
    int32_t* cons_pr;
    int32_t *pair_row, *pair_pos;
};

__device__ static SlotViews slot_views(uint8_t* base, const SlotLayout& L) {
    SlotViews V;
    SXG_GP int32_t* hdr = (SXG_GP int32_t*)(base + L.hdr);
    V.G.n_nodes = hdr; V.G.n_edges = hdr + 1;
    V.G.code = (SXG_GP uint8_t*)(base + L.code);
#define P32(f) (SXG_GP int32_t*)(base + L.f)
    V.G.rank = P32(rank); V.G.order = P32(order); V.G.order_tmp = P32(order_tmp); V.G.leader = P32(leader);
    V.G.gmem = P32(gmem); V.G.in_head = P32(in_head); V.G.in_tail = P32(in_tail); V.G.out_head = P32(out_head);
    V.G.out_tail = P32(out_tail); V.G.in_deg = P32(in_deg); V.G.out_deg = P32(out_deg);
    V.G.e_tail = P32(e_tail); V.G.e_head = P32(e_head); V.G.e_next_in = P32(e_next_in);
    V.G.e_next_out = P32(e_next_out); V.G.e_w = (SXG_GP uint32_t*)(base + L.e_w);
    V.G.posnode = P32(posnode); V.G.target = P32(target); V.G.newidx = P32(newidx); V.G.nexta = P32(nexta);
    V.G.preva = P32(preva); V.G.slotadd = P32(slotadd); V.G.kind = (SXG_GP int8_t*)(base + L.kind);
    V.G.xpos = P32(xpos); V.G.via = P32(via); V.G.dfs_stack = P32(dfs_stack); V.G.dfs_rec = P32(dfs_rec);
    V.G.sp_rank = P32(sp_rank); V.G.sp_first = P32(sp_first); V.G.sp_cnt = P32(sp_cnt);
    V.R.code = (SXG_GP uint8_t*)(base + L.r_code); V.R.flags = (SXG_GP uint8_t*)(base + L.r_flags); V.R.pred_off = P32(r_pred_off);
    V.R.preds = P32(r_preds); V.R.slot = P32(r_slot); V.R.tbx = P32(r_tbx); V.R.sseq = P32(r_sseq);
    V.R.row_node = P32(r_row_node); V.R.meta = P32(r_meta);
    V.B.tb = base + L.tb; V.B.steps = (uint32_t*)(base + L.steps);
    V.B.pool = base + L.pool; V.B.row0 = base + L.row0; V.B.park = base + L.park;
    V.B.band_strips = L.band_strips; V.B.lds_rows = L.lds_rows;
    V.cons_sc = (int64_t*)(base + L.cons_sc); V.cons_pr = (int32_t*)(base + L.cons_pr);
    V.pair_row = (int32_t*)(base + L.pair_row); V.pair_pos = (int32_t*)(base + L.pair_pos);
#undef P32
    return V;
}

__host__ __device__ static inline Scoring normalise(const sxg_poa_params& p) {
    Scoring S;
    S.m = p.m; S.n = p.n; S.g = p.g; S.e = p.e; S.q = p.q; S.c = p.c;
    S.sw = (p.mode & 1) == SXG_MODE_LOCAL;
    S.convex = 0;
    if (S.g >= S.e) { S.e = S.g; S.q = S.g; S.c = S.g; }
    else if (S.g <= S.q || S.e >= S.c) { S.q = S.g; S.c = S.e; }
    else S.convex = 1;
    return S;
}

struct BlockArgs {
    // inputs (device)
    const int32_t* blk_off; const int64_t* seq_off; const uint8_t* bases; const uint32_t* weights;
    const sxg_poa_params* params; int per_block_params;
    // work list of this launch
    const int32_t* work; int n_work; int32_t* queue;
    // slots
    uint8_t* arena; SlotLayout lay;
    // outputs (device), worst-case layout: block b's nodes/edges start at seq_off[blk_off[b]]
    int32_t* status; int32_t* n_nodes; int32_t* n_edges; int32_t* n_cons;
    uint8_t* node_code; int32_t* node_rank; int32_t* node_group;
    int32_t* edge_tail; int32_t* edge_head; uint32_t* edge_weight;
    int32_t* paths; int32_t* score; unsigned long long* cells; int32_t* cons_nodes;
    int want_consensus;
    int park_in_lds;
    int pf_off;  // byte offset of the LDS prefetch area, -1 = off
    int num_cu;  // compute units of the device (co-residency of workgroups, sxg_rotate_prio)
    uint32_t* prio_board;              // [PRIO_BOARD_CUS][PRIO_BOARD_SLOTS] progress board, shared by the launches of a round
    int prio_base;                     // first board rank of this launch (the launches before it took the ranks below)
    const unsigned long long* est;     // [n_work] estimated cells of work item wi (host cost model)
    int lds_bytes;                     // dynamic LDS of the launch (S7': the re-sort keeps its per-node states behind the control words)
    unsigned long long* blk_cycles;    // [n_blocks] shader-clock cycles the slot spent on block b (sxg_poa_batch_out::block_cycles)
};

// Kernel classes <TMAX, W>: TMAX bounds blockDim.x (the actual T = 64 * strips is a run-time
// value), W = columns per lane.  The second launch-bound is the number of waves per SIMD the
// register allocator must leave room for: 8 columns/lane need ~100 VGPRs (4 waves), 16 need
// ~165 (3 waves); a 1024-thread workgroup is 4 waves per SIMD by itself.
// RM = row mode: 0 = 32-bit sweep with int16 row words, 1 = 32-bit sweep with int32 row words,
// 2 = packed-int16 sweep (poa_dp16.hip.h; two strips per lane, W <= 12),
// 3 = banded packed sweep (poa_band16.hip.h; one wave, a sliding window of 128 strips of W = 6, 8 or 11 columns).
__host__ __device__ constexpr int sxg_min_waves(int TMAX, int W, int RM) {
#ifdef SXG_DEV_WAVES
    return SXG_DEV_WAVES;
#endif
    // (packed sweep: 128 VGPRs hold up to 13 columns per strip since round 2 -- two 8-wave workgroups share a CU)
    return TMAX > 512 ? 4 : (RM == 2 ? 4 : (W <= 12 ? 4 : 3));
}

// CB: bytes per cell of the packed sweep's traceback plane (poa_dp16.hip.h: 2 = delta codes, 4 = H and the two distances)
#ifdef SXG_DEV_TFIX128
#define SXG_TFIX_OK(tm) true
#else
#define SXG_TFIX_OK(tm) ((tm) != 128)   /* (the two-wave class measured 1 % slower with a compile-time thread count) */
#endif
// DS: packed sweep compiled for smoothxg's default scores (see dp_fill_p16); the host launches it only for blocks that have them
template <int TMAX, int W, bool CVX, int RM, bool SW, int CB = 4, bool DS = false>
__global__ __launch_bounds__(TMAX, sxg_min_waves(TMAX, W, RM)) void poa_block_kernel(const BlockArgs A) {
    constexpr bool H16 = RM != 1;
    constexpr int CPL = RM >= 2 ? 2 * W : W;  // columns per lane
    const int T = (int)blockDim.x;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* lds = (int*)smem;
    int& s_work = lds[200];
    WgCtx ctx{(sxg_lds_int*)(size_t)((unsigned)__builtin_amdgcn_groupstaticsize() + 128u * 4u)};
    const int t = threadIdx.x;
    SlotViews V = slot_views(A.arena + (size_t)blockIdx.x * A.lay.total, A.lay);
    V.B.prio_rank = (int)(blockIdx.x / (unsigned)A.num_cu + (unsigned)A.prio_base) & (PRIO_BOARD_SLOTS - 1);
    V.B.prio_board = A.prio_board ? A.prio_board + (size_t)(__smid() & (PRIO_BOARD_CUS - 1)) * PRIO_BOARD_SLOTS : nullptr;
    const RowCaps caps{A.lay.rows_cap, A.lay.pool_slots, A.lay.step_cap, A.lay.lds_rows};
    // the first item of a slot is fixed (work[blockIdx]: the host orders the list by where the slot will
    // run, see launch_plan), the rest comes from the queue
    bool first = true;
    for (;;) {
        __syncthreads();
        if (t == 0) s_work = first ? (int)blockIdx.x : (int)gridDim.x + atomicAdd(A.queue, 1);
        first = false;
        __syncthreads();
        const int wi = s_work;
        if (wi >= A.n_work) break;
        const int b = A.work[wi];
        const unsigned long long est_total = A.est ? A.est[wi] : 0ull;
        unsigned long long done_cells = 0;
        const int s0 = A.blk_off[b], s1 = A.blk_off[b + 1];
        const int64_t base0 = A.seq_off[s0];
        const Scoring S = normalise(A.params[A.per_block_params ? b : 0]);
        if (t == 0) { *V.G.n_nodes = 0; *V.G.n_edges = 0; }
        __syncthreads();
        int status = ST_OK;
        unsigned long long* prof = (unsigned long long*)(A.arena + (size_t)blockIdx.x * A.lay.total + A.lay.hdr + 64);
        V.B.row_prof = prof + 28;
        if ((t & 63) == 0) {  // placement of every wave (debug dump): smid | raw HW_ID, and the slot's start
            prof[10 + (t >> 6)] = ((unsigned long long)__smid() << 32) | (unsigned long long)__builtin_amdgcn_s_getreg(GETREG_IMMED(31, 0, HW_ID));
            if (t == 0 && prof[8] == 0) { prof[8] = (unsigned long long)wall_clock64(); prof[6] = (unsigned long long)clock64(); }
        }
        unsigned long long tc0 = clock64(), tc1;
        const unsigned long long tblk0 = tc0;
#define PROF(k) do { if (t == 0) { tc1 = clock64(); prof[k] += tc1 - tc0; tc0 = tc1; } } while (0)
        for (int s = s0; s < s1 && status == ST_OK; ++s) {
            const int64_t so = A.seq_off[s];
            const int len = (int)(A.seq_off[s + 1] - so);
            const uint8_t* seq = A.bases + so;
            for (int i = t; i < len; i += T) V.G.posnode[i] = -1;
            __syncthreads();
            const int N = *V.G.n_nodes;
            int score = 0;
            if (RM != 3 && len + 1 > T * CPL) { status = ST_TOO_LONG; break; }
            if (N + len > A.lay.nodes_cap || *V.G.n_edges + len > A.lay.nodes_cap) { status = ST_NODES_OVERFLOW; break; }
            // (the packed full-matrix sweep clamps instead and lets the traceback decide: see P16_NWFLOOR)
            if (RM != 1 && RM != 2 && !S.sw &&
                -(sxg_gap_cost(S.g, S.e, S.q, S.c, N) + sxg_gap_cost(S.g, S.e, S.q, S.c, len)) >= (RM >= 2 ? 15800 : 30000)) {
                status = ST_RANGE_OVERFLOW;
                break;
            }
            if (N > 0 && len > 0) {
                PROF(0);
                const int band_mode = RM == 3 ? (int)A.params[A.per_block_params ? b : 0].banded : 0;   // 1 = B2, 2 = adaptive (B4)
                // (workgroups of one and two waves take more elements per thread and step: see WgCtxT)
                const int hinted_ = RM == 3 ? (band_mode == 2 ? 3 : 2) : (RM == 2 ? 1 : 0);
                if (TMAX == 64 || (TMAX <= 128 && T <= 64)) { WgCtxT<16> c16{ctx.lds}; status = prep_rows(c16, V.G, V.R, caps, hinted_); }
                else if (TMAX == 128) { WgCtxT<8> c8{ctx.lds}; status = prep_rows(c8, V.G, V.R, caps, hinted_); }
                else status = prep_rows(ctx, V.G, V.R, caps, hinted_);
                if (status != ST_OK) break;
                PROF(1);
                DpResult res;
                V.B.prio_rem0 = est_total > done_cells ? est_total - done_cells : 0ull;
                if constexpr (RM == 3) {
                    // banded sweep (decrees B1-B3): one wave, the band slides with the rows; out-of-band cells do not
                    // exist, so the traceback cannot leave the kept cells
                    V.B.band_w = band_half_width(len);
                    V.B.band_mode = band_mode;
                    res = dp_fill_band16<CVX, W, SW, SW ? CB : 4>(S, V.R, N, seq, len, V.B, smem, A.cells + s);
                    __syncthreads();
                    PROF(2);
                    if (t == 0) lds[TBM_FLAG] = 0;
                    __syncthreads();
                    if (res.bi >= 0)
                        traceback_p16<false, W, CVX, true, SW ? CB : 4>(V.R, V.B, S, seq, len, res.best, T, min(256, (len << 8) / max(N, 1)), res.bi, res.bj,
                                                                        V.G.posnode, nullptr, nullptr, smem);
                    __syncthreads();
                    PROF(3);
                    // (a cell outside a row's band does not exist for the banded walk, so it cannot miss: if it ever reports one,
                    //  that is a defect -- the block fails with its own status instead of being re-run on a sweep with other semantics)
                    if (lds[TBM_FLAG]) { status = ST_INTERNAL; break; }
                } else if constexpr (RM == 2) {
                    // The traceback derives the alignment from the band of cells the sweep kept around every
                    // row's hint.  If the walk needs a cell outside (a structural variant moved the alignment
                    // more than half a band away from the backbone coordinates), the hints of the rows not yet
                    // walked are shifted onto the walk and this sequence's sweep is repeated.
                    for (int att = 0;; ++att) {
#ifdef SXG_EXP
                        // (development: a sweep with parts switched off in front of the real one -- see dp_fill_p16's EXP)
                        if (att == 0) { res = dp_fill_p16<W, CVX, SW, CB, (RM == 2 && CB == 2 && TMAX <= 128) ? ((TMAX == 64 && W <= 11) ? 2 : 1) : 0, (RM == 2 && CB == 2 && TMAX <= 512 && SXG_TFIX_OK(TMAX)) ? TMAX : 0, DS, SXG_EXP>(S, V.R, N, seq, len, V.B, smem); __syncthreads(); if (res.best == 0x7fffffff) break; }
#endif
                        res = dp_fill_p16<W, CVX, SW, CB, (RM == 2 && CB == 2 && TMAX <= 128) ? ((TMAX == 64 && W <= 11) ? 2 : 1) : 0, (RM == 2 && CB == 2 && TMAX <= 512 && SXG_TFIX_OK(TMAX)) ? TMAX : 0, DS>(S, V.R, N, seq, len, V.B, smem);
                        __syncthreads();
                        PROF(2);
                        if (t == 0) { lds[TBM_FLAG] = 0; lds[TBM_RANGE] = 0; }
                        __syncthreads();
                        if (t < 64 && res.bi >= 0)
                            traceback_p16<false, W, CVX, false, CB>(V.R, V.B, S, seq, len, res.best, T, min(256, (len << 8) / max(N, 1)), res.bi, res.bj,
                                                                    V.G.posnode, nullptr, nullptr, smem);
                        __syncthreads();
                        PROF(3);
                        if (lds[TBM_RANGE]) { status = ST_RANGE_OVERFLOW; break; }
                        if (!lds[TBM_FLAG]) break;
                        if (att == 5) { status = ST_BAND_MISS; break; }
                        const int mrow = lds[TBM_ROW], mdelta = lds[TBM_DELTA];
                        __syncthreads();
                        for (int r2 = t; r2 < mrow; r2 += T) V.R.meta[8 * (size_t)r2 + 7] += mdelta;
                        for (int i2 = t; i2 < len; i2 += T) V.G.posnode[i2] = -1;
                        if (t == 0) prof[27] += 1;
                        __syncthreads();
                    }
                    if (status != ST_OK) break;
                } else {
                    dp_fill<W, CVX, H16, SW>(S, V.R, N, seq, len, V.B, smem, A.park_in_lds != 0, A.pf_off, res);
                    __syncthreads();
                    PROF(2);
                    if (t == 0 && res.bi >= 0) traceback<false>(V.R, V.B, T, W, S.sw, res.bi, res.bj, V.G.posnode, nullptr, nullptr);
                    __syncthreads();
                    PROF(3);
                }
                score = res.bi >= 0 ? res.best : 0;
            }
            if (t == 0) { A.score[s] = score; if (RM != 3 || N == 0 || len == 0) A.cells[s] = RM == 3 ? 0ull : (unsigned long long)N * (unsigned long long)len; }
            done_cells += (unsigned long long)N * (unsigned long long)len;
            // (spoa's order: the re-sort below rebuilds order and ranks; AddAlignment leaves its own bookkeeping of them out)
            const bool spoa_order = (A.params[A.per_block_params ? b : 0].mode & SXG_ORDER_SPOA) != 0;
            if (TMAX == 64 || (TMAX <= 128 && T <= 64)) { WgCtxT<16> c16{ctx.lds}; add_alignment(c16, V.G, seq, len, A.weights ? A.weights[s] : 1u, A.paths + so, !spoa_order); }
            else if (TMAX == 128) { WgCtxT<8> c8{ctx.lds}; add_alignment(c8, V.G, seq, len, A.weights ? A.weights[s] : 1u, A.paths + so, !spoa_order); }
            else add_alignment(ctx, V.G, seq, len, A.weights ? A.weights[s] : 1u, A.paths + so, !spoa_order);
            if (spoa_order) {   // S7': spoa's depth-first re-sort (every thread walks its roots: poa_graph_dev.h)
                typedef __attribute__((address_space(3))) uint8_t lds_u8;
                // (N nodes before this alignment; the block's first re-sort builds everything, the later ones what the alignment touched)
                spoa_resort(ctx, V.G, (lds_u8*)(size_t)((unsigned)__builtin_amdgcn_groupstaticsize() + (unsigned)LDS_CTL_BYTES), A.lds_bytes - LDS_CTL_BYTES,
                            (int)(A.seq_off[s0 + 1] - A.seq_off[s0]), len, N, s == s0, prof + 41);
                __syncthreads();
            }
            PROF(4);
        }
        PROF(0);
        __syncthreads();
        // results of the block
        const int N = *V.G.n_nodes, E = *V.G.n_edges;
        if (status == ST_OK) {
            for (int v = t; v < N; v += T) {
                A.node_code[base0 + v] = V.G.code[v];
                A.node_rank[base0 + v] = V.G.rank[v];
                A.node_group[base0 + v] = V.G.leader[v];
            }
            for (int e = t; e < E; e += T) {
                A.edge_tail[base0 + e] = V.G.e_tail[e];
                A.edge_head[base0 + e] = V.G.e_head[e];
                A.edge_weight[base0 + e] = V.G.e_w[e];
            }
            int nc = 0;
            if (A.want_consensus && t == 0) nc = consensus_serial(V.G, V.cons_sc, V.cons_pr, A.cons_nodes + base0);
            if (t == 0) { A.n_nodes[b] = N; A.n_edges[b] = E; A.n_cons[b] = nc; }
        } else if (t == 0) { A.n_nodes[b] = 0; A.n_edges[b] = 0; A.n_cons[b] = 0; }
        if (t == 0) { A.status[b] = status; if (A.blk_cycles) A.blk_cycles[b] = (unsigned long long)clock64() - tblk0; }
        if (t == 0 && V.B.prio_board) __hip_atomic_store(V.B.prio_board + V.B.prio_rank, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        PROF(5);
        if (t == 0) { prof[9] = (unsigned long long)wall_clock64(); prof[7] = (unsigned long long)clock64(); }
#undef PROF
    }
}

// ---------------------------------------------------------------------------------------
struct AlignArgs {
    const int64_t* row_off; const uint8_t* row_code; const uint8_t* row_sink; const int64_t* pred_off;
    const int32_t* preds; const int64_t* seq_off; const uint8_t* bases; const sxg_poa_params* params;
    int per_problem_params;
    const int32_t* work; int n_work; int32_t* queue;
    uint8_t* arena; SlotLayout lay;
    int32_t* status; int32_t* score; int32_t* n_pairs;
    int32_t* pair_row; int32_t* pair_pos;  // worst-case layout: problem p at row_off[p] + seq_off[p]
    int park_in_lds;
    int pf_off;
    int num_cu;
};

template <int TMAX, int W, bool CVX, int RM, bool SW>
__global__ __launch_bounds__(TMAX, sxg_min_waves(TMAX, W, RM)) void poa_align_kernel(const AlignArgs A) {
    constexpr bool H16 = RM != 1;
    constexpr int CPL = RM == 2 ? 2 * W : W;
    const int T = (int)blockDim.x;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* lds = (int*)smem;
    int& s_work = lds[200];
    WgCtx ctx{(sxg_lds_int*)(size_t)((unsigned)__builtin_amdgcn_groupstaticsize() + 128u * 4u)};
    const int t = threadIdx.x;
    SlotViews V = slot_views(A.arena + (size_t)blockIdx.x * A.lay.total, A.lay);
    V.B.prio_rank = (int)(blockIdx.x / (unsigned)A.num_cu) & (PRIO_BOARD_SLOTS - 1);
    V.B.prio_board = nullptr; V.B.prio_rem0 = 0; V.B.row_prof = nullptr;
    const RowCaps caps{A.lay.rows_cap, A.lay.pool_slots, A.lay.step_cap, A.lay.lds_rows};
    for (;;) {
        __syncthreads();
        if (t == 0) s_work = atomicAdd(A.queue, 1);
        __syncthreads();
        const int wi = s_work;
        if (wi >= A.n_work) break;
        const int p = A.work[wi];
        const int64_t r0 = A.row_off[p];
        const int N = (int)(A.row_off[p + 1] - r0);
        const int64_t so = A.seq_off[p];
        const int len = (int)(A.seq_off[p + 1] - so);
        const Scoring S = normalise(A.params[A.per_problem_params ? p : 0]);
        const int64_t e0 = A.pred_off[r0];
        int status = ST_OK, score = 0, npairs = 0;
        if (len + 1 > T * CPL) status = ST_TOO_LONG;
        else if (N > A.lay.rows_cap) status = ST_ROWS_OVERFLOW;
        else if (N > 0 && len > 0) {
            for (int r = t; r < N; r += T) {
                V.R.code[r] = A.row_code[r0 + r];
                V.R.pred_off[r] = (int)(A.pred_off[r0 + r] - e0);
                V.R.row_node[r] = r;
                V.R.slot[r] = r;
                V.R.tbx[r] = 0;  // (packed sweep: the align-only arenas keep every strip, hints are not used)
                int fl = A.row_sink[r0 + r] ? ROW_SINK : 0;
                for (int64_t k = A.pred_off[r0 + r]; k < A.pred_off[r0 + r + 1]; ++k)   // the previous rank is a predecessor:
                    if (A.preds[k] == r) fl |= ROW_REGPRED;                               // its values are still in registers
                V.R.flags[r] = (uint8_t)fl;
            }
            if (t == 0) V.R.pred_off[N] = (int)(A.pred_off[r0 + N] - e0);
            __syncthreads();
            const int E = V.R.pred_off[N];
            for (int k = t; k < E; k += T) V.R.preds[k] = A.preds[e0 + k];
            __syncthreads();
            for (int r = t; r < N; r += T)
                for (int k = V.R.pred_off[r]; k < V.R.pred_off[r + 1]; ++k) {
                    const int pr = V.R.preds[k];  // 1-based row of the predecessor
                    if (pr >= 1 && pr != r) {     // not the previous rank -> that row must be stored
                        atomicOr((unsigned*)(V.R.flags + ((pr - 1) & ~3)), (unsigned)ROW_STORE << (8 * ((pr - 1) & 3)));
                        atomicMax(&V.R.slot[pr - 1], r);
                    }
                }
            status = finish_rows(ctx, N, V.R, caps, RM == 2 ? 1 : 0);
            if (status == ST_OK) {
                DpResult res;
                if constexpr (RM == 2) res = dp_fill_p16<W, CVX, SW, 4>(S, V.R, N, A.bases + so, len, V.B, smem);
                else dp_fill<W, CVX, H16, SW>(S, V.R, N, A.bases + so, len, V.B, smem, A.park_in_lds != 0, A.pf_off, res);
                __syncthreads();
                if constexpr (RM == 2) {
                    if (t == 0) { lds[TBM_FLAG] = 0; lds[TBM_RANGE] = 0; }
                    __syncthreads();
                    // (STRICT: these alignments were admitted under the strict range rule -- no cell is clamped, and there is no wider
                    //  re-run behind this kernel: the walk must not stop at a legitimately low score; round 5 returned a truncated
                    //  pair list with status OK for global alignments scoring below -16 000 + m L)
                    if (t < 64 && res.bi >= 0) npairs = traceback_p16<true, W, CVX, false, 4, true>(V.R, V.B, S, A.bases + so, len, res.best, T, min(256, (len << 8) / max(N, 1)), res.bi, res.bj, nullptr, V.pair_row, V.pair_pos, smem);
                    __syncthreads();
                    if (lds[TBM_FLAG]) status = ST_BAND_MISS;  // (cannot happen: these arenas keep every strip)
                    else if (lds[TBM_RANGE]) status = ST_RANGE_OVERFLOW;  // (cannot happen under STRICT)
                } else if (t == 0 && res.bi >= 0) npairs = traceback<true>(V.R, V.B, T, W, S.sw, res.bi, res.bj, nullptr, V.pair_row, V.pair_pos);
                if (t == 0 && res.bi >= 0 && status == ST_OK) {
                    score = res.best;
                    const int64_t out0 = A.row_off[p] + A.seq_off[p];
                    for (int k = 0; k < npairs; ++k) {  // reverse into the output
                        A.pair_row[out0 + k] = V.pair_row[npairs - 1 - k] - 1;
                        A.pair_pos[out0 + k] = V.pair_pos[npairs - 1 - k];
                    }
                }
            }
        }
        if (t == 0) { A.status[p] = status; A.score[p] = score; A.n_pairs[p] = npairs; }
    }
}
