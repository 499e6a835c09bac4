// poa_types.h -- shared constants and flat views for the MI355X blocked-POA engine.
//
// Replaces what the reference keeps inside spoa::Graph / spoa::AlignmentEngine objects
// (call sites src/smooth.cpp:752-769) with flat, block-private arrays in HBM.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SXG_HD __host__ __device__ __forceinline__
#else
#define SXG_HD inline
#endif

namespace sxg {

constexpr int NEG = -(1 << 29);

// traceback byte: [2:0] source of H, then "came from EXTEND" flags of the four gap states
enum : int { SRC_STOP = 0, SRC_D = 1, SRC_F = 2, SRC_O = 3, SRC_E = 4, SRC_Q = 5 };
constexpr int TB_FEXT = 0x08, TB_OEXT = 0x10, TB_EEXT = 0x20, TB_QEXT = 0x40;

// per-row flags prepared for the DP
constexpr int ROW_STORE = 1;  // some successor is not rank+1 -> row goes to the row pool
constexpr int ROW_SINK = 2;   // no out-edge (NW end candidates)

// status codes (per block); mirrored in include/sxg_poa.h
enum : int {
    ST_OK = 0,
    ST_ROWS_OVERFLOW = 1,   // graph grew past the traceback plane's row capacity
    ST_POOL_OVERFLOW = 2,   // live pred rows exceed the row pool
    ST_TBX_OVERFLOW = 3,    // too many multi-pred rows / in-degree beyond plane capacity
    ST_NODES_OVERFLOW = 4,
    ST_TOO_LONG = 5,
};

struct Scoring {
    int m, n, g, e, q, c;  // normalised (linear: e=q=c=g; affine: q=g,c=e)
    int sw;                // 1 = local
    int convex;            // 0 = O/Q states are copies of F/E and are skipped
};

// Graph of ONE block, slot-private working set.  Capacities: nodes/edges <= sum of the
// block's sequence lengths.
struct GraphView {
    int32_t *n_nodes, *n_edges;  // scalars in global memory (slot header)
    uint8_t *code;
    int32_t *rank, *order, *order_tmp, *leader, *gmem;
    int32_t *in_head, *in_tail, *out_head, *out_tail, *in_deg, *out_deg;
    int32_t *e_tail, *e_head, *e_next_in, *e_next_out;
    uint32_t *e_w;
    // scratch for add_alignment (length >= max sequence length / max nodes + 1)
    int32_t *posnode, *target, *newidx, *nexta, *preva, *slotadd;
    int8_t *kind;
};

// Row structures of the current graph in rank space, rebuilt before every alignment.
struct RowsView {
    uint8_t *code;       // [N]
    uint8_t *flags;      // [N]
    int32_t *pred_off;   // [N+1]
    int32_t *preds;      // [E] row indices (rank+1), in-edge insertion order
    int32_t *slot;       // [N] row-pool slot of stored rows
    int32_t *tbx;        // [N] >=0: row index in the u16 ordinal plane; <=-2: -(idx+2) in the
                         //      u32 plane; -1: single-pred row
    int32_t *sseq;       // [N+1] exclusive count of stored rows (scratch)
    int32_t *row_node;   // [N] node id at rank
};

}  // namespace sxg
