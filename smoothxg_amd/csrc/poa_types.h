// poa_types.h -- shared constants and flat views for the MI355X blocked-POA engine.
//
// Replaces what the reference keeps inside spoa::Graph / spoa::AlignmentEngine objects
// (call sites src/smooth.cpp:752-769) with flat, block-private arrays in HBM.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SXG_HD __host__ __device__ __forceinline__
// the graph phases stay OUT of line in the persistent kernel: inlined, their ~50 array
// descriptors stay live across the DP sweep and push it into scratch spills
#define SXG_HD_PHASE __host__ __device__ __noinline__ inline
#else
#define SXG_HD inline
#define SXG_HD_PHASE inline
#endif

namespace sxg {

constexpr int NEG = -(1 << 29);

// traceback byte: [2:0] source of H, then "came from EXTEND" flags of the four gap states
enum : int { SRC_STOP = 0, SRC_D = 1, SRC_F = 2, SRC_O = 3, SRC_E = 4, SRC_Q = 5 };
constexpr int TB_FEXT = 0x08, TB_OEXT = 0x10, TB_EEXT = 0x20, TB_QEXT = 0x40;

// per-row flags prepared for the DP
constexpr int ROW_STORE = 1;  // some successor is not rank+1 -> row goes to the row pool
constexpr int ROW_SINK = 2;   // no out-edge (NW end candidates)
constexpr int ROW_REGPRED = 4;  // the previous rank is one of the row's predecessors (its values are still in registers)

// status codes (per block); mirrored in include/sxg_poa.h
enum : int {
    ST_OK = 0,
    ST_ROWS_OVERFLOW = 1,   // graph grew past the traceback plane's row capacity
    ST_POOL_OVERFLOW = 2,   // live pred rows exceed the row pool
    ST_TBX_OVERFLOW = 3,    // too many multi-pred rows / in-degree beyond plane capacity
    ST_NODES_OVERFLOW = 4,
    ST_TOO_LONG = 5,
    ST_RANGE_OVERFLOW = 6,  // a global alignment outgrew the score range of the narrow sweep (re-run wider)
    ST_INTERNAL = 8,        // a state the kernels hold impossible (the banded traceback reporting a miss): final, never retried
    ST_BAND_MISS = 7,       // packed sweep: the traceback kept leaving the band of stored cells (re-run with a plane that keeps every strip)
};

// Lower bound of every reachable global-alignment score: the all-gap path (a gap down a chain of
// `rows` predecessors plus a gap over `cols` columns); H is a maximum over paths, so it is never below.
// ---- layout of the packed sweeps' traceback plane (poa_dp16.hip.h, poa_band16.hip.h): host + device, so that the CPU suite
// can check it.  A row holds BS slots (strip s in slot s mod BS) of W columns-in-strip, in GROUPS of four columns:
// [group][slot][column of the group]; the last group is W - 4 * (groups - 1) columns wide.  BS is a multiple of 4.
SXG_HD constexpr int plane_round4(int bs) { return (bs + 3) & ~3; }
SXG_HD constexpr int plane_group_width(int W, int gi) { return W - 4 * gi >= 4 ? 4 : W - 4 * gi; }
// dword of column-in-strip k of slot `slot` inside its row
SXG_HD constexpr int plane_cell_in_row(int W, int BS, int slot, int k) {
    return 4 * (k >> 2) * BS + slot * plane_group_width(W, k >> 2) + (k & 3);
}

SXG_HD long sxg_gap_cost(int g, int e, int q, int c, long k) {
    if (k <= 0) return 0;
    const long a = g + (k - 1) * (long)e, b = q + (k - 1) * (long)c;
    return a > b ? a : b;
}

struct Scoring {
    int m, n, g, e, q, c;  // normalised (linear: e=q=c=g; affine: q=g,c=e)
    int sw;                // 1 = local
    int convex;            // 0 = O/Q states are copies of F/E and are skipped
};

// Pointer members of the views are global-address-space pointers in device code: read back from a
// struct, a plain pointer is "flat" to the compiler, and flat accesses count on lgkmcnt as well as
// vmcnt, so every LDS wait of the block-wide scans also waited for HBM.
#if defined(__HIP_DEVICE_COMPILE__)
#define SXG_GP __attribute__((address_space(1)))
#else
#define SXG_GP
#endif

// Graph of ONE block, slot-private working set.  Capacities: nodes/edges <= sum of the
// block's sequence lengths.
struct GraphView {
    SXG_GP int32_t *n_nodes, *n_edges;  // scalars in global memory (slot header)
    SXG_GP uint8_t *code;
    SXG_GP int32_t *rank, *order, *order_tmp, *leader, *gmem;
    SXG_GP int32_t *in_head, *in_tail, *out_head, *out_tail, *in_deg, *out_deg;
    SXG_GP int32_t *e_tail, *e_head, *e_next_in, *e_next_out;
    SXG_GP uint32_t *e_w;
    // scratch for add_alignment (length >= max sequence length / max nodes + 1)
    SXG_GP int32_t *posnode, *target, *newidx, *nexta, *preva, *slotadd;
    SXG_GP int8_t *kind;
    // backbone coordinate of every node: the DP column at which it is expected to align (first
    // sequence: its own position; later nodes inherit / interpolate from the nodes they were aligned
    // next to).  Centres the band of cells the packed sweep keeps for the traceback; never affects results.
    SXG_GP int32_t *xpos;
    // S7' (spoa's depth-first re-sort, sxg_poa_params::mode | SXG_ORDER_SPOA): the node a node was created aligned to (-1:
    // created unaligned) -- it fixes the order of spoa's aligned-node lists --, and the scratch of the re-sort
    SXG_GP int32_t *via;
    SXG_GP int32_t *dfs_stack;   // [n_edges + 6 n_nodes + 8]: the walks' stacks, one piece per root
    SXG_GP int32_t *dfs_rec;     // [16 n_nodes]: per-node records of the walk (poa_graph_dev.h::spoa_resort); kept from re-sort to re-sort
    // what the previous re-sort of this block left for the next one (round 6: only what an alignment touched is sorted again)
    SXG_GP int32_t *sp_rank;     // [n_nodes] spoa rank of every node
    SXG_GP int32_t *sp_first;    // [n_nodes] first() of every node: the root whose walk finished it
    SXG_GP int32_t *sp_cnt;      // [n_nodes] nodes the walk from root s finished (at roots)
};

// Row structures of the current graph in rank space, rebuilt before every alignment.
struct RowsView {
    SXG_GP uint8_t *code;       // [N]
    SXG_GP uint8_t *flags;      // [N]
    SXG_GP int32_t *pred_off;   // [N+1]
    SXG_GP int32_t *preds;      // [E] row indices (rank+1), in-edge insertion order
    SXG_GP int32_t *slot;       // [N] row-pool slot of stored rows
    SXG_GP int32_t *tbx;        // [N] 32-bit sweep: first fold step of a multi-pred row in the step-mask plane
                         //      (row with np preds owns np-1 steps); -1: single-pred row.
                         //      packed sweep: band hint = expected DP column of the row's node
    SXG_GP int32_t *sseq;       // [N+1] exclusive count of stored rows (scratch)
    SXG_GP int32_t *row_node;   // [N] node id at rank
    SXG_GP int32_t *meta;       // [N*8] per-row DP descriptor, see RowMeta
};

// What the DP needs to start a row, gathered into 32 bytes so the sweep reads it from an
// LDS-staged chunk instead of chasing four dependent global loads per row.
struct RowMeta {
    int pb;        // offset of the predecessor list in RowsView::preds
    int info;      // np | code << 16 | flags << 24
    int p0, s0;    // first predecessor row (0 = virtual source) and its pool slot (-1: none/regs)
    int p1, s1;    // second predecessor row / slot (np >= 2)
    int slot;      // own pool slot (-1: row not stored)
    int tbx;       // as RowsView::tbx (fold step / band hint)
};

// The views reach the out-of-line graph phases by reference to the kernel's private memory: read through that reference, every
// array base is a FLAT load from scratch in front of the access it serves (72 of them in prep_rows alone, round 5's ISA) and a
// 64-bit VGPR address behind it.  A phase therefore starts by copying the view it was handed into one whose pointers went
// through the scalar unit once (readfirstlane: they are the same for every thread of the workgroup): the accesses become
// "scalar base + lane offset", the pointers live in SGPRs.  (Host builds -- the CPU emulation harness -- copy plainly.)
#if defined(__HIP_DEVICE_COMPILE__)
template <class Tp> __device__ __forceinline__ Tp sxg_scalar_ptr(Tp p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (Tp)(((unsigned long long)hi << 32) | lo);
}
#else
template <class Tp> inline Tp sxg_scalar_ptr(Tp p) { return p; }
#endif
SXG_HD GraphView sxg_scalar_view(const GraphView& G) {
    GraphView U;
#define SXG_U(f) U.f = sxg_scalar_ptr(G.f)
    SXG_U(n_nodes); SXG_U(n_edges); SXG_U(code); SXG_U(rank); SXG_U(order); SXG_U(order_tmp); SXG_U(leader); SXG_U(gmem);
    SXG_U(in_head); SXG_U(in_tail); SXG_U(out_head); SXG_U(out_tail); SXG_U(in_deg); SXG_U(out_deg);
    SXG_U(e_tail); SXG_U(e_head); SXG_U(e_next_in); SXG_U(e_next_out); SXG_U(e_w);
    SXG_U(posnode); SXG_U(target); SXG_U(newidx); SXG_U(nexta); SXG_U(preva); SXG_U(slotadd); SXG_U(kind);
    SXG_U(xpos); SXG_U(via); SXG_U(dfs_stack); SXG_U(dfs_rec); SXG_U(sp_rank); SXG_U(sp_first); SXG_U(sp_cnt);
#undef SXG_U
    return U;
}
SXG_HD RowsView sxg_scalar_view(const RowsView& R) {
    RowsView U;
#define SXG_U(f) U.f = sxg_scalar_ptr(R.f)
    SXG_U(code); SXG_U(flags); SXG_U(pred_off); SXG_U(preds); SXG_U(slot); SXG_U(tbx); SXG_U(sseq); SXG_U(row_node); SXG_U(meta);
#undef SXG_U
    return U;
}

}  // namespace sxg
