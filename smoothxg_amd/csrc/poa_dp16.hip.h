// poa_dp16.hip.h -- packed-int16 DP sweep (gfx950 v_pk_add_i16 / v_pk_max_i16 / v_pk_sub_i16) and the
// traceback that DERIVES the alignment from stored values.
//
// Same semantics (S1-S5) and the same row-uniform structure as poa_dp.hip.h, but every VGPR
// holds TWO cells: lane t owns strip "lo" = columns [t*W, (t+1)*W) and strip "hi" = columns
// [T*W + t*W, T*W + (t+1)*W); the low / high 16 bits of a register are the two strips, so every
// add/max serves two cells.  Measured on MI355X (profiles/ubench/valu_rate.hip): a wave64
// integer VALU instruction issues every 4 cycles whether it is 32-bit or packed 16-bit, and the
// 32-bit sweep is bound by exactly that issue rate -- packing is the lever.
//
// Round 2: the sweep records NO decisions.  No packed compare exists on gfx950, so every "which
// candidate won" bit used to be the sign of a packed difference shifted into a mask word (3 VALU
// instructions per decision, 8 decisions per packed column: a third of the row) and the mask plane was
// 40 % of the kernel's HBM writes -- for a walk that visits ~10^4 of 2.7*10^7 cells.  Now:
//  * rows carry OUTGOING gap candidates oF = max(H+g, F+e), oO = max(H+q, O+c) instead of F and O:
//    computed once at the end of a row, consumed for free by a register successor and with an unpack
//    by a stored one; a row word is packed H plus the two 8-bit distances H-oF, H-oO;
//  * every row additionally writes those words for a BAND of strips around the column where its node
//    is expected to align (the node's backbone coordinate, kept by add_alignment) to the traceback
//    plane, one dword per cell: H int16 | dF << 16 | dO << 24;
//  * the traceback re-applies the tie rules S3 to the candidates of each visited cell (executable
//    specification: oracle/poa_vtb.c, checked against the recorded-choice oracle): D from the
//    predecessors' H, F / O from their outgoing candidates, E / Q from a leftward scan of the row's own
//    H; the walk carries the value of the state it is in, so OPEN-versus-EXTEND is one equality test.
//    A cell outside the band is a BAND MISS: the hints of the rows above are shifted and the alignment's
//    sweep is repeated (rare: the band is ~1100 columns wide; see poa_block_kernel).
//  * the end cell of a local alignment is found with a packed running maximum per row and a
//    wave-uniform search of the column only in rows that improve it.
// Applicability: every reachable score and intermediate fits +-15800 (host check); otherwise the
// 32-bit sweep runs.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "poa_dp.hip.h"

namespace sxg {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

constexpr int NEGP = -16384;  // "minus infinity" of the packed sweep
// GLOBAL alignments whose all-gap corner scores leave int16 (affine 1,4,6,2 at 5 kbp: -20 006) still run the packed sweep
// (round 5): H is clamped at P16_NWFLOOR from below -- the same one instruction per cell that clamps a local alignment at
// 0 -- which bounds every gap state too (each is some H plus one opening at least).  A clamped cell is larger than its true
// value, and whatever it feeds gains at most m per column, so every cell with H > P16_NWFLOOR + m * L + m is EXACT, and so is
// every decision the traceback takes in such a cell (a candidate out of the clamped region cannot equal its value).  The
// traceback checks that of every cell it visits; a walk that meets a smaller H -- two sequences with little in common --
// reports ST_RANGE_OVERFLOW and the block is re-run on the 32-bit sweep, as before.  (Rounds 1-4 sent every global
// alignment whose corner did not fit down that ladder up front: 130 blocks/s instead of the packed sweep's rate.)
constexpr int P16_NWFLOOR = -16000;
// smoothxg's default scores in the engine's (spoa's) sign convention: the score set the DS kernel classes are compiled for
constexpr int P16_DEF_M = 1, P16_DEF_N = -4, P16_DEF_G = -6, P16_DEF_E = -2, P16_DEF_Q = -26, P16_DEF_C = -1;
__host__ __device__ inline bool p16_default_scores(const Scoring& S) {
    return S.convex && S.m == P16_DEF_M && S.n == P16_DEF_N && S.g == P16_DEF_G && S.e == P16_DEF_E && S.q == P16_DEF_Q && S.c == P16_DEF_C;
}
// LOCAL alignments run the packed sweep on BIASED fields: a register half holds score + P16_BIAS, always inside
// [P16_FLOOR, 32767].  Every reachable value of a local alignment is >= min(g, q) >= -120 (H >= 0, every gap state is
// some H plus at most one opening) and the few "minus infinity" inputs (the column left of column 0, a carry that enters
// strip 0) are replaced by P16_FLOOR - P16_BIAS = -512, which never wins a maximum.  Fields that never leave
// [0, 65535] make a PLAIN 32-bit add/subtract of a constant -- or of any register whose fields do not underflow --
// correct on both halves at once (no borrow ever crosses bit 16), and on gfx950 v_add_u32 / v_sub_u32 issue every 2
// cycles per wave where every VOP3P instruction (v_pk_add_i16, v_pk_max_i16, ...) takes 4
// (profiles/r03/op_rate.txt): 37 % of the row loop's VALU instructions were packed adds and subtracts.
constexpr int P16_BIAS = 1024, P16_FLOOR = 512;

__device__ __forceinline__ int pk_add(int a, int b) { return __builtin_bit_cast(int, (s16x2)(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ int pk_sub(int a, int b) { return __builtin_bit_cast(int, (s16x2)(__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ int pk_max(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ int pk_mad(int a, int b, int c) { return __builtin_bit_cast(int, (s16x2)(__builtin_bit_cast(s16x2, a) * __builtin_bit_cast(s16x2, b) + __builtin_bit_cast(s16x2, c))); }
__device__ __forceinline__ int pk_lshr(int a, int sh) { return __builtin_bit_cast(int, (u16x2)(__builtin_bit_cast(u16x2, a) >> __builtin_bit_cast(u16x2, sh))); }
__device__ __forceinline__ int pk2(int lo, int hi) { return (lo & 0xffff) | (hi << 16); }
__device__ __forceinline__ int pk_lo(int v) { return (int)(short)(v & 0xffff); }
__device__ __forceinline__ int pk_hi(int v) { return v >> 16; }

// Row words of the packed ring, per lane and column: packed H, and the distances H - oF, H - oO
// to the row's OUTGOING gap candidates oF = max(H + g, F + e), oO = max(H + q, O + c) -- what every
// successor takes as its F / O.  H >= F and g <= e < 0 bound the distances to [-e, -g] and [-c, -q]:
// one byte each, never clamped (host check: |g|, |q| <= 120).
// (BIASED: the fields are biased and h >= of, oo field by field, so the differences are plain 32-bit subtractions)
template <bool CVX, bool BIASED = false>
__device__ __forceinline__ u32x2 p16_pack_row(int h, int of, int oo) {
    const int df = BIASED ? (int)((unsigned)h - (unsigned)of) : pk_sub(h, of);
    const int dq = BIASED ? (int)((unsigned)h - (unsigned)oo) : pk_sub(h, oo);
    return u32x2{(unsigned)h, (unsigned)(CVX ? (df | (dq << 8)) : df)};
}
template <bool BIASED = false>
__device__ __forceinline__ void p16_unpack_row(u32x2 w, int& h, int& of, int& oo) {
    h = (int)w.x;
    const int df = (int)(w.y & 0x00ff00ffu), dq = (int)((w.y >> 8) & 0x00ff00ffu);
    of = BIASED ? (int)((unsigned)h - (unsigned)df) : pk_sub(h, df);
    oo = BIASED ? (int)((unsigned)h - (unsigned)dq) : pk_sub(h, dq);
}

// ---- 2-byte plane cells (CB = 2, round 5) --------------------------------------------------------------------------
// The band stores of the traceback plane were the largest single cost of the packed sweep (cost map of round 4: 24 % of the
// headline launch; 45 % of the sweep of 8000 blocks of 16 x 1 kbp, where the launch sits at the HBM write roof).  A plane cell
// held H (16 bits) and the two distances H - oF, H - oO (8 bits each).  H need not be stored in full: along a row it moves in
// small steps --
//     g <= H[i][j] - H[i][j-1] <= m - g        (g = the cheapest gap opening, normalised scores; proof in DESIGN.md section 3.1:
//     the lower bound is the in-row gap E >= H[j-1] + g, the upper one follows by induction over the ranks from
//     H[i][j-1] >= H[p][j-1] + g for the predecessor p a diagonal step into (i, j) came from)
// -- and the distances lie in [-e, -g] and [-c, -q].  A cell is therefore the 16-bit code
//     (H[j] - H[j-1] - g)  |  (H - oF + e) << bH  |  (H - oO + c) << (bH + bF),      bH + bF + bO <= 16,
// and a strip of a row is W + 1 halfwords: the H of the column LEFT of the strip (the value the sweep hands from lane to
// lane anyway; strip 0: H of column 0, whose own step is written as 0), then the W codes.  12 bits for the default scores
// 1,4,6,2,26,1; 16 for pggb's asm10 set; a score set that needs more (asm5: 1,19,39,3,81,1 -- 20 bits) takes the 4-byte
// cells (CB = 4: the round-4 format, every kernel class exists in both).  The traceback rebuilds H by summing the steps of
// a strip from its left end -- at most W additions for a cell it visits.
struct P16Delta {
    int bH, bF, bO;      // field widths
    int g, eabs, cabs;   // what the fields are offset by: dH - g, dF - |e|, dO - |c| are >= 0
};
__host__ __device__ inline int p16_bits_for(int n_values) { int b = 0; while ((1 << b) < n_values) ++b; return b; }
__host__ __device__ inline P16Delta p16_delta_of(const Scoring& S) {
    P16Delta D;
    D.g = S.g; D.eabs = -S.e; D.cabs = -S.c;
    D.bH = p16_bits_for(S.m - 2 * S.g + 1);
    D.bF = p16_bits_for(S.e - S.g + 1);
    D.bO = S.convex ? p16_bits_for(S.c - S.q + 1) : 0;
    return D;
}
__host__ __device__ inline bool p16_delta_fits(const Scoring& S) { const P16Delta D = p16_delta_of(S); return D.bH + D.bF + D.bO <= 16; }
// dwords of one strip of a row: W + 1 halfwords
__host__ __device__ constexpr int p16_slot_dwords(int W, int CB) { return CB == 2 ? (W + 2) / 2 : W; }

// Buffer addressing for the sweep's rows: address = descriptor base (one row of the ring / of the plane,
// built per row with a few SALU instructions) + SGPR offset (column k of the strip) + ONE loop-invariant VGPR
// offset (the lane).  With global_load/store the compiler formed 64-bit VGPR addresses per access
// (v_lshl_add_u64) and kept the per-column offsets in SGPR pairs, which it then spilled to VGPR lanes
// (v_readlane per use): ~25 % of the row's VALU instructions were address bookkeeping.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t p16_rsrc(const void* base, const int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// ---- the traceback plane of the packed sweep: a band of strips per row -------------------------
// Strip s = columns [s*W, (s+1)*W) (s < T: lo strips, s >= T: hi strips).  Row r keeps the BS strips
// starting at band_first_strip(hint of r), strip s in slot s mod BS.  A row's cells are laid out in GROUPS of four
// columns-in-strip: [group][slot][column of the group], i.e. cell (r, column j), k = j % W, slot = (j / W) % BS is the dword
//     plane[(r * W + 4 * (k / 4)) * BS + slot * gw(k / 4) + k % 4]  =  H int16 | (H - oF) << 16 | (H - oO) << 24,
// gw(group) = min(4, W - 4 * group).  A lane then writes (reads) its strip with ceil(W / 4) 16-byte instructions, each of
// which covers 1 KB of consecutive memory per wave (round 4; rounds 1-3 had [column][slot]: W four-byte instructions of
// 256 B each -- the packed sweep's band stores were 24 % of its launch, most of it instruction issue and the vmcnt they hold).
// BS is a multiple of 4: rows and groups start on 16-byte boundaries.
// (geometries of up to 1408 columns keep every strip: the band would cover most of the row anyway, and a plane that holds
//  whole rows doubles as the row ring -- see ring_plane in dp_fill_p16)
__host__ __device__ constexpr int p16_band_strips(int T, int W) {
    return 2 * T * W <= 1408 ? 2 * T : (plane_round4((1100 + W - 1) / W) < 2 * T ? plane_round4((1100 + W - 1) / W) : 2 * T);
}
template <int W>
__device__ __forceinline__ size_t plane_cell(const size_t row, const int BS, const int slot, const int k) {
    return row * (size_t)(W * BS) + (size_t)plane_cell_in_row(W, BS, slot, k);   // (poa_types.h)
}
// byte offset of cell k of slot `slot` inside its row (for buffer accesses through a per-row descriptor)
template <int W>
__device__ __forceinline__ unsigned plane_cell_byte(const int BS, const unsigned slot, const int k) {
    return (unsigned)plane_cell_in_row(W, BS, (int)slot, k) * 4u;
}
// Cache policy of the plane stores: NON-TEMPORAL (aux bit 1 = nt).  The plane is written once per row and read back by the
// traceback for ~1 % of its cells (the banded sweep: by 40 % of the rows, a few rows later); streamed past the L2 it stops evicting
// the ring rows and descriptors that ARE re-read.  Same box, one gpurun call: headline 1 987 -> 1 922 ms, c3b 1 400 -> 1 514
// blocks/s, c2 +5 %.  (nt on the ring stores as well: +-0 on the headline; nt on the banded sweep's plane LOADS: c3b -2 %.)
#ifndef SXG_PLANE_AUX
#define SXG_PLANE_AUX 2
#endif
// a slot no row has: slot * (bytes of a group) lies beyond every row's buffer descriptor and stays below 2^32 (see P16_STORES)
#define P16_SLOT_OOB 0x04000000u
// one strip of a row: cell(k) -> the dword of column k; rs = the row's descriptor
template <int W, int GI, class F>
__device__ __forceinline__ void plane_store_group(const __amdgpu_buffer_rsrc_t rs, const unsigned slot, const int BS, F& cell) {
    constexpr int k = 4 * GI, gw = W - k >= 4 ? 4 : W - k;
    // The group's offset travels in the VGPR offset, the SGPR offset field stays 0: a store of more than 64 bits reads its
    // upper data registers after it has issued, and the compiler only keeps the next VALU write away from them (the "VMEM
    // store data" hazard) when the SGPR offset field is NOT a register -- with `s_offset` = GI * BS * 16 it scheduled the next
    // group's v_perm right behind the store, and on gfx950 the stored cells were then the NEXT group's (measured: band misses
    // on every deep block; the same layout with dword stores was exact).
    const unsigned vo = slot * (unsigned)(gw * 4) + (unsigned)(GI * BS * 16);
#ifdef SXG_PLANE_NARROW
    for (int x = 0; x < gw; ++x) __builtin_amdgcn_raw_buffer_store_b32(cell(k + x), rs, vo, 4 * x, 0);
    return;
#endif
    if constexpr (gw == 4)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{cell(k), cell(k + 1), cell(k + 2), cell(k + 3)}, rs, vo, 0, SXG_PLANE_AUX);
    else if constexpr (gw == 3)
        __builtin_amdgcn_raw_buffer_store_b96(u32x3{cell(k), cell(k + 1), cell(k + 2)}, rs, vo, 0, SXG_PLANE_AUX);
    else if constexpr (gw == 2)
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{cell(k), cell(k + 1)}, rs, vo, 0, SXG_PLANE_AUX);
    else
        __builtin_amdgcn_raw_buffer_store_b32(cell(k), rs, vo, 0, SXG_PLANE_AUX);
}
template <int W, class F, int... GI>
__device__ __forceinline__ void plane_store_groups(const __amdgpu_buffer_rsrc_t rs, const unsigned slot, const int BS, F& cell, std::integer_sequence<int, GI...>) {
    (plane_store_group<W, GI>(rs, slot, BS, cell), ...);
}
template <int W, class F>
__device__ __forceinline__ void plane_store_strip(const __amdgpu_buffer_rsrc_t rs, const unsigned slot, const int BS, F cell) {
    plane_store_groups<W>(rs, slot, BS, cell, std::make_integer_sequence<int, (W + 3) / 4>{});
}
template <int W, int GI>
__device__ __forceinline__ void plane_load_group(const __amdgpu_buffer_rsrc_t rs, const unsigned slot, const int BS, unsigned (&out)[W]) {
    constexpr int k = 4 * GI, gw = W - k >= 4 ? 4 : W - k;
    if constexpr (gw == 4) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, slot << 4, GI * BS * 16, 0);
        out[k] = v.x; out[k + 1] = v.y; out[k + 2] = v.z; out[k + 3] = v.w;
    } else if constexpr (gw == 3) {
        const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, slot * 12u, GI * BS * 16, 0);
        out[k] = v.x; out[k + 1] = v.y; out[k + 2] = v.z;
    } else if constexpr (gw == 2) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, slot << 3, GI * BS * 16, 0);
        out[k] = v.x; out[k + 1] = v.y;
    } else
        out[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, slot << 2, GI * BS * 16, 0);
}
template <int W, int... GI>
__device__ __forceinline__ void plane_load_groups(const __amdgpu_buffer_rsrc_t rs, const unsigned slot, const int BS, unsigned (&out)[W], std::integer_sequence<int, GI...>) {
    (plane_load_group<W, GI>(rs, slot, BS, out), ...);
}
template <int W>
__device__ __forceinline__ void plane_load_strip(const __amdgpu_buffer_rsrc_t rs, const unsigned slot, const int BS, unsigned (&out)[W]) {
    plane_load_groups<W>(rs, slot, BS, out, std::make_integer_sequence<int, (W + 3) / 4>{});
}
// one strip of a row through a plain pointer (the traceback: every lane its own row): group by group, as wide as the group
template <int SDW>
__device__ __forceinline__ void plane_load_slot(SXG_GLOBAL const uint32_t* const row, const int BS, const int slot, unsigned (&out)[SDW]) {
#pragma unroll
    for (int gi = 0; gi < (SDW + 3) / 4; ++gi) {
        const int gw = SDW - 4 * gi >= 4 ? 4 : SDW - 4 * gi;
        SXG_GLOBAL const uint32_t* const q = row + 4 * gi * BS + slot * gw;
        if (gw == 4) { const u32x4 v = *(SXG_GLOBAL const u32x4*)q; out[4 * gi] = v.x; out[4 * gi + 1] = v.y; out[4 * gi + 2] = v.z; out[4 * gi + 3] = v.w; }
        else if (gw == 3) { typedef u32x3 __attribute__((aligned(4))) u32x3_a4; const u32x3 v = *(SXG_GLOBAL const u32x3_a4*)q; out[4 * gi] = v.x; out[4 * gi + 1] = v.y; out[4 * gi + 2] = v.z; }
        else if (gw == 2) { const u32x2 v = *(SXG_GLOBAL const u32x2*)q; out[4 * gi] = v.x; out[4 * gi + 1] = v.y; }
        else out[4 * gi] = *q;
    }
}
__device__ __forceinline__ int band_first_strip(const int hint_col, const int W, const int BS, const int T) {
    const int c = hint_col / W - (BS >> 1);
    return min(max(c, 0), 2 * T - BS);
}

#ifndef SXG_P16_INLINE
#define SXG_P16_INLINE __noinline__
#endif
// (out of line, views by value: inlined into the persistent kernel the sweep shared ~100 SGPRs with the
// kernel's own state and reloaded its loop-invariant scalars from VGPR lanes -- v_readlane + hazard nops --
// all over the row loop)
// EXP (development, profiles/tools/cost_map.sh): bits switch parts of the row OFF in a sweep whose results are thrown away
// (the kernel runs it in front of the real one), so that the difference between two builds prices a part of the row:
// 1 = no band stores, 2 = no ring stores, 4 = no end-cell bookkeeping, 8 = no carry scans, 16 = no mailbox exchange,
// 32 = every row takes the register-predecessor path (no fetch, no fold), 64 = no pass 2, 128 = no pass 1.
// CB: bytes per plane cell (2: delta codes, see P16Delta; 4: H | H - oF | H - oO).
// End cell of a LOCAL alignment (round 5): every lane keeps, per strip, ONE 32-bit key -- the strip's greatest H so far in the
// upper half, 0xffff minus the row that first reached it in the lower -- updated with a max per row (5 instructions; rounds 2-4
// compared, voted and searched the strip's columns in every row that improved any lane, i.e. nearly every row of the waves the
// best diagonal runs through: 15 % of the row on 1 kbp blocks).  Rows are numbered within epochs of 65536; at an epoch's end the
// wave folds its keys into a scalar 64-bit best.  The sweep returns the STRIP of the end cell (bj = -(strip + 1)); the
// traceback reads the column out of the plane row it starts from.
// RP: the class may run with a plane that keeps every strip and read stored rows back from it (ring_plane below) -- the
// one- and two-wave classes (TMAX = 128); the wider classes are compiled without that path (its three fetch sites and their
// scalars cost the four-wave headline class SGPR spills in the row loop).
// (RP: 0 = never; 1 = when the launch's plane keeps every strip, a run-time test; 2 = ALWAYS -- round 6: the one-wave classes up to W = 11
//  cover at most 1 408 columns, their plane keeps every strip by construction (p16_band_strips), and compiled for that alone the fold has
//  ONE form: with both, every fold ended in 30 v_mov that merged the two forms' registers.  The host sends a one-wave launch whose plane
//  was narrowed -- the test knob SXG_POA_BAND_COLS -- to the two-wave class run at 64 threads.)
template <int W, bool CVX, bool SW, int CB = 4, int RP = 0, int TFIX = 0, bool DS = false, int EXP = 0>
__device__ SXG_P16_INLINE DpResult dp_fill_p16(const Scoring S, const RowsView R, const int N_,
                                             const uint8_t* seq, const int L_, const DpBuffers B,
                                             char* smem) {
    DpResult res;
    const int N = __builtin_amdgcn_readfirstlane(N_), L = __builtin_amdgcn_readfirstlane(L_);
    static_assert(W >= 4 && W <= 15, "strip width");
    // (TFIX: the class runs at exactly this many threads -- everything derived from T folds into immediates: 1.2 % on the headline)
    const int T = TFIX ? TFIX : (int)blockDim.x;
    constexpr bool ONEW = TFIX == 64;
    const int NW = T >> 6;
    const int TW = T * W;           // columns of one half
    const int MB = dp16_meta_bytes(T);   // (bytes of the LDS area the traceback window uses; the sweep keeps its mailboxes there)
    int* lds = (int*)smem;
    // (TFIX = 64: a one-wave class -- no left or right neighbour, the mailbox code folds away: 1.7 % on 8000 x 16 x 1 kbp)
    const int t = threadIdx.x, lane = t & 63, wv = ONEW ? 0 : __builtin_amdgcn_readfirstlane(t >> 6);
    // Wave w owns the 128 strips [128 w, 128 w + 128): lane l its strips 128 w + l (low halves) and 128 w + 64 + l (high
    // halves).  A wave's columns are contiguous, so the ONLY thing that crosses a wave boundary inside a row is what crosses
    // one column boundary: the gap states entering the wave's first column and the diagonal's source left of it.
    const int s_lo = wv * 128 + lane, s_hi = s_lo + 64;
    const int j0 = s_lo * W, j0h = s_hi * W;   // first columns of my two strips
    // scoring values are block-uniform: keep them (and everything derived) in SGPRs
    // DS: the class is compiled FOR smoothxg's default scores 1,4,6,2,26,1 (src/main.cpp:322-327) -- the six values and everything
    // derived from them are immediates instead of ~16 loop-invariant scalar registers, which the row loop otherwise spills to
    // VGPR lanes and reads back per use (headline, same box: 1 910 -> 1 848 ms).  Other score sets take the generic class.
    const int g = DS ? P16_DEF_G : __builtin_amdgcn_readfirstlane(S.g), e = DS ? P16_DEF_E : __builtin_amdgcn_readfirstlane(S.e);
    const int q = DS ? P16_DEF_Q : __builtin_amdgcn_readfirstlane(S.q), c = DS ? P16_DEF_C : __builtin_amdgcn_readfirstlane(S.c);
    const int sm = DS ? P16_DEF_M : __builtin_amdgcn_readfirstlane(S.m), sn = DS ? P16_DEF_N : __builtin_amdgcn_readfirstlane(S.n);
    const int G2 = pk2(g, g), E2 = pk2(e, e), Q2 = pk2(q, q), C2 = pk2(c, c);
    // local alignment: biased fields (see P16_BIAS); "x + K" for a penalty K <= 0 is then x - |K| * 0x10001 in 32 bits
    constexpr int BIAS = SW ? P16_BIAS : 0, FLOORV = SW ? P16_FLOOR : NEGP;
    const int NEG2 = pk2(FLOORV, FLOORV), B2 = pk2(BIAS, BIAS), NWF2 = pk2(P16_NWFLOOR, P16_NWFLOOR);
    const unsigned Gm = (unsigned)(-g) * 0x10001u, Em = (unsigned)(-e) * 0x10001u, Qm = (unsigned)(-q) * 0x10001u, Cm = (unsigned)(-c) * 0x10001u;
#define P16_DEC(x, Km, K2) (SW ? (int)((unsigned)(x) - (Km)) : pk_add((x), (K2)))   /* x + K */
#define P16_INC(x, Km, K2) (SW ? (int)((unsigned)(x) + (Km)) : pk_sub((x), (K2)))   /* x - K */
    // substitution scores come out of a per-row byte table (see pass 1): all-mismatch rows of it, and match ^ mismatch
    const unsigned SC_N4 = (unsigned)(sn & 0xff) * 0x01010101u, SC_MX = (unsigned)((sm ^ sn) & 0xff);
    const int We = W * e, Wc = W * c;
    const int BS = __builtin_amdgcn_readfirstlane(B.band_strips);
    constexpr int SD = p16_slot_dwords(W, CB);   // dwords of one strip in a plane row
    // (CB = 2) the multipliers that shift a cell's two distances into their fields of the code
    Scoring SD_ = S;
    if (DS) { SD_.m = P16_DEF_M; SD_.n = P16_DEF_N; SD_.g = P16_DEF_G; SD_.e = P16_DEF_E; SD_.q = P16_DEF_Q; SD_.c = P16_DEF_C; SD_.convex = 1; }
    const P16Delta DF = p16_delta_of(SD_);
    const int KF2 = pk2(1 << __builtin_amdgcn_readfirstlane(DF.bH), 1 << __builtin_amdgcn_readfirstlane(DF.bH));
    const int KO2 = pk2(1 << __builtin_amdgcn_readfirstlane(DF.bH + DF.bF), 1 << __builtin_amdgcn_readfirstlane(DF.bH + DF.bF));
    // ... and what reading a row back out of the plane needs (full-width planes only, see ring_plane below)
    const int dbH_ = __builtin_amdgcn_readfirstlane(DF.bH), dbF_ = __builtin_amdgcn_readfirstlane(DF.bF);
    const int D_SH1 = pk2(dbH_, dbH_), D_SH2 = pk2(dbH_ + dbF_, dbH_ + dbF_);
    const int D_MH = pk2((1 << dbH_) - 1, (1 << dbH_) - 1), D_MF = pk2((1 << dbF_) - 1, (1 << dbF_) - 1);
    const int d_cst_ = g + ((-e) << dbH_) + ((CVX ? -c : 0) << (dbH_ + dbF_));
    const int D_CST = pk2(d_cst_, d_cst_);
    const int D_EA = pk2(-e, -e), D_CA = pk2(-c, -c);
    // A plane that keeps EVERY strip of every row (sequences up to ~1.4 kbp -- the blocks smoothxg's default -l 700 ... 1100
    // produces --, the every-strip plane of a band-miss re-run) already holds what the row ring would: stored rows are not
    // written a second time, a stored predecessor is read back from its plane row and decoded.  (Cost map of 8000 x 16 x 1 kbp,
    // round 5: ring stores 14 % of the sweep, at the HBM write roof.)
    const bool ring_plane = RP == 2 ? true : (RP == 1 && CB == 2 && BS == 2 * T);
    // ---- wave pipeline.  The waves of a workgroup do NOT meet inside the row loop.  Wave w sweeps row i as soon as wave
    // w-1 has handed over, through a ring of P16_MBOX mailboxes in LDS, the three values that cross its left edge in row i:
    // E and Q entering its first column and H of the column left of it (one 16-byte word {E, Q, H, row}, written and read
    // by single LDS instructions; the row number is the "full" flag).  Everything else a wave reads it wrote itself: rows
    // in the ring are lane-private, and a stored row carries the left-neighbour column of every lane as one more word.
    // Waves drift apart by up to P16_MBOX rows (a writer checks every P16_MBOX / 2 rows that its reader has freed the
    // slots it is about to reuse), so a wave that waits -- for a stored predecessor row, for its SIMD -- delays its
    // successors only once that slack is used up: the per-row cost is the AVERAGE over the waves, not the maximum that
    // two s_barrier per row made it (47 % of the wave time was parked in round 2).
    typedef __attribute__((address_space(3))) i32x4 lds_i32x4;
    typedef __attribute__((address_space(3))) int lds_i32;
    constexpr int P16_MBOX = 16;
    const unsigned lds0 = (unsigned)__builtin_amdgcn_groupstaticsize();
    volatile lds_i32x4* const mb_in = (volatile lds_i32x4*)(size_t)(lds0 + (unsigned)(LDS_CTL_BYTES + (max(wv, 1) - 1) * P16_MBOX * 16));
    volatile lds_i32x4* const mb_out = (volatile lds_i32x4*)(size_t)(lds0 + (unsigned)(LDS_CTL_BYTES + wv * P16_MBOX * 16));
    volatile lds_i32* const prog = (volatile lds_i32*)(size_t)(lds0 + 64u * 4u);   // [16] rows consumed by wave w
    if (lane < P16_MBOX) mb_out[lane] = i32x4{0, 0, 0, 0};
    if (lane == 0) prog[wv] = 0;
    __syncthreads();

    // query letters, one byte per (strip, column), as SELECTORS into the row's score table: A,C,G,T,N = 0..4, "no letter"
    // (padding columns, never a match) = 5.  Register k2 holds the bytes (lo_k+1, lo_k, hi_k+1, hi_k) of columns
    // k = 2 k2 and k + 1 -- the order in which one v_perm turns them into four scores with column k's pair on the odd
    // bytes, where a second v_perm can sign-extend them into a packed pair.
    constexpr int NL = (W + 1) / 2;
    unsigned let[NL];
#pragma unroll
    for (int k2 = 0; k2 < NL; ++k2) {
        unsigned v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int kc = 2 * k2 + ((b & 1) ? 0 : 1);  // strip-local column (W odd: one spare pair)
            const int j = ((b >> 1) ? j0h : j0) + kc;
            const bool valid = kc < W && j >= 1 && j <= L;
            const unsigned ch = valid ? (unsigned)seq[j - 1] : 5u;
            v |= (valid && ch > 4u ? 4u : ch) << (8 * b);
        }
        let[k2] = v;
    }
    // The letters live in LDS, not in registers: six loop-invariant VGPRs less, which is what the
    // allocator otherwise evicts to scratch around the multi-predecessor path -- and a scratch reload
    // is an in-order vmcnt wait behind the stores just issued.  Lane-major, read back with
    // two wide LDS loads per row.
    // (raw LDS byte offset: the dynamic LDS starts right behind the kernel's static LDS)
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    const int LDS_ROWS = __builtin_amdgcn_readfirstlane(B.lds_rows);
    const unsigned llet_off = (unsigned)__builtin_amdgcn_groupstaticsize() + (unsigned)(LDS_CTL_BYTES + MB + LDS_ROWS * (TW * 8 + T * 4) + t * NL * 4);
#pragma unroll
    for (int k2 = 0; k2 < NL; ++k2) ((lds_u32*)(size_t)llet_off)[k2] = let[k2];

    // Rows in HBM (ring, row 0) are laid out [column-in-strip][lane] and the plane
    // [row][column-in-strip][strip of the band]: a load/store instruction then covers consecutive words of
    // a wave instead of 64 different cache lines.  Every access is "uniform pointer"[lane offset]: scalar
    // base, one loop-invariant lane offset register, no per-access address arithmetic.
    const unsigned ut8 = (unsigned)t * 8u;   // my byte offset inside a [column][lane] row of 8-byte words
    SXG_GLOBAL u32x2* const g_row0 = sxg_uniform(sxg_global((u32x2*)B.row0));
    SXG_GLOBAL u32x2* const g_pool = sxg_uniform(sxg_global((u32x2*)B.pool));
    SXG_GLOBAL uint32_t* const g_tb = sxg_uniform(sxg_global((uint32_t*)B.tb));
    SXG_GLOBAL const int32_t* const g_meta = sxg_uniform(sxg_global((const int32_t*)R.meta));
    SXG_GLOBAL const int32_t* const g_preds = sxg_uniform(sxg_global((const int32_t*)R.preds));
    SXG_GLOBAL const int32_t* const g_slot = sxg_uniform(sxg_global((const int32_t*)R.slot));
    int Hp[W], Fp[W], Op[W], Hleft;
    // virtual row 0
#pragma unroll
    for (int k = 0; k < W; ++k) {
        int h2[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int j = (hf ? j0h : j0) + k;
            int h = BIAS;
            if (!SW && j > 0) { const int a = g + (j - 1) * e, b = q + (j - 1) * c; h = max(a > b ? a : b, P16_NWFLOOR); }
            h2[hf] = h;
        }
        Hp[k] = pk2(h2[0], h2[1]);
        Fp[k] = P16_DEC(Hp[k], Gm, G2);             // nothing to extend in row 0: both candidates open
        Op[k] = CVX ? P16_DEC(Hp[k], Qm, Q2) : NEG2;
    }
    {
        int h2[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int j = (hf ? j0h : j0) - 1;
            int h = BIAS;
            if (!SW && j > 0) { const int a = g + (j - 1) * e, b = q + (j - 1) * c; h = max(a > b ? a : b, P16_NWFLOOR); }
            h2[hf] = j < 0 ? FLOORV : h;
        }
        Hleft = pk2(h2[0], h2[1]);
    }
    // a stored row: W columns of 8-byte words [column][lane], then the column LEFT of every lane's strips (4 bytes per lane).
    // (Pairs of columns per 16-byte access, as in the plane, were measured in round 4: 90 instead of 131 memory instructions in
    // the row loop, 2 010 against 1 996 ms on the headline, c3 -1 %, c4 +1 %: the ring's cost is its volume, not its instruction
    // count -- kept at 8 bytes.)
    const int RB = TW * 8 + T * 4;
    {
        const __amdgpu_buffer_rsrc_t rs0 = p16_rsrc((const void*)g_row0, RB);
#pragma unroll
        for (int k = 0; k < W; ++k)
            __builtin_amdgcn_raw_buffer_store_b64(p16_pack_row<CVX, SW>(Hp[k], Fp[k], Op[k]), rs0, ut8, k * T * 8, 0);
        __builtin_amdgcn_raw_buffer_store_b32((unsigned)Hleft, rs0, (unsigned)t * 4u, TW * 8, 0);
    }
    int best_lo = NEGP * 2, best_hi = best_lo, bi_lo = -1, bi_hi = -1, bk_lo = 0, bk_hi = 0;   // (global alignment only)
    const int kL_lo = L - j0, kL_hi = L - j0h;  // strip-local index of the end column L
    unsigned key_lo = 0u, key_hi = 0u;          // local alignment: (greatest H of my strip) << 16 | 0xffff - (row & 0xffff)
    unsigned long long ekey = 0ull;             // wave-uniform: best of the finished epochs (value, row, strip -- see fold_keys)
    // per-lane keys of the epoch that ends in front of row `i_end` -> ekey.  value << 32 | (0xFFFFF - row) << 12 | (0xFFF - strip):
    // greatest score, then smallest row, then smallest strip (strips are disjoint column ranges in ascending order)
    auto fold_keys = [&](const int i_end) {
        const unsigned ep = (unsigned)(i_end - 1) >> 16;
        auto k64 = [&](const unsigned k, const int strip) -> unsigned long long {
            if (k == 0u) return 0ull;
            const unsigned row = (ep << 16) | (0xffffu - (k & 0xffffu));
            return ((unsigned long long)(k >> 16) << 32) | ((unsigned long long)(0xFFFFFu - row) << 12) | (unsigned long long)(0xFFFu - (unsigned)strip);
        };
        unsigned long long kk = k64(key_lo, s_lo);
        const unsigned long long k2 = k64(key_hi, s_hi);
        kk = k2 > kk ? k2 : kk;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long o = __shfl_xor(kk, d);
            kk = o > kk ? o : kk;
        }
        const unsigned hi_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(kk >> 32)), lo_ = (unsigned)__builtin_amdgcn_readfirstlane((int)kk);
        const unsigned long long ku = ((unsigned long long)hi_ << 32) | lo_;
        ekey = ku > ekey ? ku : ekey;
        key_lo = 0u; key_hi = 0u;
    };

    // (An L2 warm-up for the next row's stored predecessor -- one load per wave touching its 44 cache lines a row ahead,
    // retired before the stores -- measured 0-5 % SLOWER in round 2: returns are in order, so whatever is in front of the
    // demand loads holds them back, and a wait behind it stalls.)
    // (A register/LDS prefetch of the next row's stored predecessor, requested after pass 2 and collected before
    // this row's stores, was measured again in round 2 with the spills gone: 2.81 s against 2.68 s.  At this VALU
    // occupancy the co-resident workgroups already hide the round trip; its ~60 extra instructions do not pay.)
    typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
    // on-chip copies of stored rows (finish_rows: a stored row whose last reader comes before LDS_ROWS more rows are stored
    // lives in LDS only and never travels to HBM): behind the control words and the mailbox / traceback-window area
    const unsigned lds_rows0 = lds0 + (unsigned)(LDS_CTL_BYTES + MB);
// my words of on-chip row copy b_: W 8-byte words [wave][column][lane] (a wave's 64 lanes side by side: every LDS access of
// the sweep is conflict-free, and column k is the immediate offset k * 512), then the left-neighbour word [lane].  The
// addresses are rebuilt from an opaque copy of the thread index: no loop-invariant address registers.
#define P16_LDS_ROW(words_, left_, b_)                                                                      \
    int tq_ = t;                                                                                            \
    asm volatile("" : "+v"(tq_));                                                                           \
    const unsigned lb_ = lds_rows0 + (unsigned)(b_) * (unsigned)RB;                                         \
    lds_u32x2* const words_ = (lds_u32x2*)(size_t)(lb_ + (unsigned)(wv * (512 * W - 512)) + (unsigned)tq_ * 8u); \
    lds_u32* const left_ = (lds_u32*)(size_t)(lb_ + (unsigned)(TW * 8) + (unsigned)tq_ * 4u)
    // (B lives in the kernel's private memory: testing B.prio_board per row was a scratch load plus an
    // in-order vmcnt(0) -- a wait for every store of the previous row -- at the top of EVERY row)
    const bool has_board = __builtin_amdgcn_readfirstlane((int)(B.prio_board != nullptr)) != 0;
    // slots of my two strips in a plane row (strip s -> slot s mod BS: the address of a cell does not depend
    // on where the row's band starts, so the traceback fetches cells and row descriptors in ONE round trip)
    const unsigned soff = (unsigned)(s_lo % BS) | ((unsigned)(s_hi % BS) << 16);
    const int prio_rank = __builtin_amdgcn_readfirstlane(B.prio_rank);
#ifdef SXG_ROW_PROF
// profiling builds: wait for the fetched row right away and book the time as segment 6 ("stored row fetch")
#define P16_PROF_FETCH() do { RP_MARK(0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); RP_MARK(6); } while (0)
#else
#define P16_PROF_FETCH() do { } while (0)
#endif
#ifdef SXG_ROW_PROF
    unsigned long long racc[8] = {0};  // per row segment; scalar registers (s_memtime deltas)
#define RP_MARK(seg) do { const unsigned long long tn_ = __builtin_readcyclecounter(); racc[seg] += tn_ - rt_; rt_ = tn_; } while (0)
#else
#define RP_MARK(seg) do { } while (0)
#endif
    // Row descriptors (RowMeta, 32 bytes) come through the SCALAR data path, one s_load_dwordx8 per row issued a row ahead:
    // every wave is at its own row, so there is no chunk of descriptors the workgroup could stage together.  The scalar
    // cache is not coherent with the vector stores that wrote the descriptors (prep_rows, the band-miss hint shift): it is
    // invalidated once per sweep, after the workgroup barrier that follows those stores.
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(4))) const i32x8 const_i32x8;
    const_i32x8* const cmeta = (const_i32x8*)(unsigned long long)g_meta;
    __builtin_amdgcn_s_dcache_inv();
    i32x8 dnext = cmeta[0];
    for (int i = 1; i <= N; ++i) {
        if (has_board) { if ((i & 127) == 1) sxg_balance_prio(B, (unsigned long long)i * (unsigned long long)L); }
        else if ((i & 63) == 1) sxg_rotate_prio(prio_rank);
        const i32x8 dsc = dnext;
        dnext = cmeta[min(i, N - 1)];
        const int pb = dsc[0], info = dsc[1], p0 = dsc[2], s0 = dsc[3], p1 = dsc[4], s1 = dsc[5], myslot = dsc[6], hint = dsc[7];
        const int np = info & 0xffff, code = (info >> 16) & 0xff, flags = (info >> 24) & 0xff;
        // score table of this row: byte c = score(node letter, query letter c); c = 5 (no letter) never matches
        const unsigned SC_T0 = SC_N4 ^ (code < 4 ? SC_MX << (8 * code) : 0u), SC_T1 = SC_N4 ^ (code == 4 ? SC_MX : 0u);

#ifdef SXG_ROW_PROF
        unsigned long long rt_ = __builtin_readcyclecounter();
#endif
        int Hc[W];
        // Fp/Op arrive holding the previous row's OUTGOING candidates.  (Rounds 1-5 had a sibling rule: a row whose successor has
        // the same single predecessor -- an alternative allele, 15 % of the rows -- kept its own F/O for it and the successor fetched
        // only the diagonal.  Round 6 measured it as a LOSS: "update in place or keep" was a merge the register allocator resolved
        // with a second set of 2 W registers and 22 v_mov at the top of four rows in five, and the second store path doubled the
        // loop's code.  Without it a sibling unpacks F/O from its predecessor's stored row like any other row: headline
        // 1 859 -> 1 786 ms on one box, 1 417 -> 1 025 static VALU instructions and 132 -> 42 v_mov in the loop.)
// words of the stored row of predecessor p_ (slot sl_) and the column to their left (stored with the row: every word of
// a stored row was written by the lane that reads it).  sl_ <= -2: the row is one of the LDS_ROWS on-chip copies.
#define P16_FETCH(p_, sl_, wr_, hl_)                                                                        \
    do {                                                                                                    \
        if ((sl_) <= -2) {                                                                                  \
            P16_LDS_ROW(la_, lf_, -2 - (sl_));                                                              \
            _Pragma("unroll") for (int k = 0; k < W; ++k) wr_[k] = la_[k * 64];                             \
            hl_ = (int)*lf_;                                                                                \
        } else {                                                                                            \
            const __amdgpu_buffer_rsrc_t rs_ = p16_rsrc(((p_) == 0) ? (const void*)g_row0 : (const void*)((SXG_GLOBAL const char*)g_pool + (size_t)(sl_) * (size_t)RB), RB); \
            _Pragma("unroll") for (int k = 0; k < W; ++k) wr_[k] = __builtin_amdgcn_raw_buffer_load_b64(rs_, ut8, k * T * 8, 0); \
            P16_PROF_FETCH();                                                                               \
            hl_ = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_, ut8 >> 1, TW * 8, 0);                       \
            P16_DRAIN();                                                                                    \
        }                                                                                                   \
    } while (0)

// ... or, with ring_plane, the row's strips out of its plane row: H of the column left of the strips, then per column the
// code's three fields; HS_/FS_/OS_[k] receive the row's H and outgoing candidates (FO_ = false: H only, for a sibling)
#define P16_FETCH_PLANE(p_, FO_, HS_, FS_, OS_, hl_)                                                        \
    do {                                                                                                    \
        const __amdgpu_buffer_rsrc_t rsp_ = p16_rsrc((const void*)(g_tb + (size_t)(p_) * (size_t)(SD * BS)), SD * BS * 4); \
        unsigned dl_[SD], dh_[SD];                                                                          \
        plane_load_strip<SD>(rsp_, soff & 0xffffu, BS, dl_);                                                \
        plane_load_strip<SD>(rsp_, soff >> 16, BS, dh_);                                                    \
        P16_PROF_FETCH();                                                                                   \
        P16_DRAIN();                                                                                        \
        int run_ = (int)__builtin_amdgcn_perm(dh_[0], dl_[0], 0x05040100u);                                 \
        hl_ = t == 0 ? (int)(((unsigned)run_ & 0xffff0000u) | ((unsigned)FLOORV & 0xffffu)) : run_;         \
        _Pragma("unroll") for (int k = 0; k < W; ++k) {                                                     \
            const int cn_ = pk_sub((int)__builtin_amdgcn_perm(dh_[(k + 1) >> 1], dl_[(k + 1) >> 1], ((k + 1) & 1) ? 0x07060302u : 0x05040100u), D_CST); \
            run_ = pk_add(pk_add(run_, cn_ & D_MH), G2);                                                    \
            HS_[k] = run_;                                                                                  \
            if (FO_) {                                                                                      \
                FS_[FO_ ? k : 0] = pk_sub(pk_sub(run_, pk_lshr(cn_, D_SH1) & D_MF), D_EA);                  \
                if (CVX) OS_[FO_ ? k : 0] = pk_sub(pk_sub(run_, pk_lshr(cn_, D_SH2)), D_CA);                \
            }                                                                                               \
        }                                                                                                   \
    } while (0)

// Every branch that loaded a stored row from HBM ends by draining its loads itself.  Otherwise the compiler, which cannot
// know at the join which branch ran, waits for vmcnt(0) at the top of pass 1 of EVERY row (a loaded register that a branch
// did not consume is reused there) -- and on gfx9 vmcnt also counts stores, so rows that read nothing waited for the
// write acknowledgements of the previous row's ring and band stores.
#define P16_DRAIN() __builtin_amdgcn_s_waitcnt(0x0F70) /* vmcnt(0), expcnt / lgkmcnt untouched */
        // The diagonal sources are left WHERE THEY ARE -- Hp[k] = max over the predecessors of their H in column k,
        // Hleft the column left of the strip -- and pass 1 walks the strip right to left, so that column k's new value
        // can take the register of Hp[k] (dead once column k+1 has used it): the shifted copy Hc[k] = Hp[k-1] that an
        // ascending pass needs cost 11-23 v_mov per row.
        if ((EXP & 32) || (np <= 1 && p0 == i - 1)) {
            // register predecessor: its outgoing candidates ARE this row's F and O
        } else {
            // Several predecessors: D, F and O are plain maxima over them (which predecessor won is re-derived by the
            // traceback), so the fold order is free: when the previous row is one of them (ROW_REGPRED) the registers are
            // the running maxima as they stand, otherwise the first predecessor's stored row is unpacked into them; every
            // other predecessor is folded in.
            const bool regbase = (flags & ROW_REGPRED) != 0;
            if (!regbase) {
                int hl;
                if (ring_plane && s0 >= 0 && p0 != 0) P16_FETCH_PLANE(p0, true, Hp, Fp, Op, hl);
                else {
                    u32x2 wr[W];
                    P16_FETCH(p0, s0, wr, hl);
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        p16_unpack_row<SW>(wr[k], Hp[k], Fp[k], Op[k]);
                        SXG_PIN("+v"(Hp[k]), "+v"(Fp[k]), "+v"(Op[k]));
                    }
                }
                Hleft = hl;
            }
// fold one more predecessor row (p_, slot sl_) into the running maxima
#define P16_FOLD(p_, sl_)                                                                                   \
    do {                                                                                                    \
        int hl;                                                                                             \
        if (ring_plane && (sl_) >= 0 && (p_) != 0) {                                                        \
            int hs_[W], fs_[W], os_[W];                                                                     \
            P16_FETCH_PLANE(p_, true, hs_, fs_, os_, hl);                                                   \
            _Pragma("unroll") for (int k = 0; k < W; ++k) {                                                 \
                Fp[k] = pk_max(Fp[k], fs_[k]);                                                              \
                if (CVX) Op[k] = pk_max(Op[k], os_[k]);                                                     \
                Hp[k] = pk_max(Hp[k], hs_[k]);                                                              \
                SXG_PIN("+v"(Hp[k]), "+v"(Fp[k]), "+v"(Op[k]));                                             \
            }                                                                                               \
        } else {                                                                                            \
        u32x2 wr[W];                                                                                        \
        P16_FETCH(p_, sl_, wr, hl);                                                                         \
        _Pragma("unroll") for (int k = 0; k < W; ++k) {                                                     \
            int hs, fs, os;                                                                                 \
            p16_unpack_row<SW>(wr[k], hs, fs, os);                                                          \
            Fp[k] = pk_max(Fp[k], fs);                                                                      \
            if (CVX) Op[k] = pk_max(Op[k], os);                                                             \
            Hp[k] = pk_max(Hp[k], hs);                                                                      \
            SXG_PIN("+v"(Hp[k]), "+v"(Fp[k]), "+v"(Op[k]));                                                 \
        }                                                                                                   \
        }                                                                                                   \
        Hleft = pk_max(Hleft, hl);                                                                          \
    } while (0)
            // (the first two predecessors in straight-line code: two-predecessor rows -- the closing node of every
            // bubble -- are a third of all rows; the loop form made the allocator spill around them)
            if (regbase && p0 != i - 1) P16_FOLD(p0, s0);
            if (np >= 2 && p1 != i - 1) P16_FOLD(p1, s1);
            for (int x = 2; x < np; ++x) {
                const int p = __builtin_amdgcn_readfirstlane(g_preds[pb + x]);
                if (p == i - 1) continue;
                const int sl = p >= 1 ? __builtin_amdgcn_readfirstlane(g_slot[p - 1]) : -1;
                P16_FOLD(p, sl);
            }
#undef P16_FOLD
        }
#undef P16_DRAIN
#undef P16_FETCH
        if (!CVX) {
#pragma unroll
            for (int k = 0; k < W; ++k) Op[k] = NEG2;
        }

        RP_MARK(0);  // predecessor rows read, F/O/diagonal set up
        {   // this row's copy of my query letters (an opaque address keeps the loads inside the loop)
            unsigned lo_ = llet_off;
            asm volatile("" : "+v"(lo_));
#pragma unroll
            for (int k2 = 0; k2 < NL; ++k2) let[k2] = ((lds_u32*)(size_t)lo_)[k2];
        }
        // ---- pass 1 (right to left): H before the in-row gaps, strip-local carries
        int a = NEG2, b = NEG2;
        unsigned sc4 = 0;
#pragma unroll
        for (int k = W - 1; k >= 0; --k) {
            if (EXP & 128) { Hc[k] = Fp[k]; continue; }
            // four scores per letter register in one table look-up; the odd bytes -- column k's pair, after a byte shift
            // column k+1's -- are sign-extended into a packed pair by a second permute.  (2 + 1/2 instructions and an
            // add per column; the compare-free form before -- xor, extract, min with 1, multiply-add, add m -- took 4 1/2)
            if ((k & 1) || k == W - 1) sc4 = __builtin_amdgcn_perm(SC_T1, SC_T0, let[k >> 1]);
            const int sc = (int)__builtin_amdgcn_perm(0u, (k & 1) ? (sc4 << 8) : sc4, 0x09030801u);
            int h = pk_add(k ? Hp[k - 1] : Hleft, sc);    // diagonal + (match ? m : n)
            h = pk_max(h, Fp[k]);
            if (CVX) h = pk_max(h, Op[k]);
            Hc[k] = h;
            // a = max_k' (h_k' - (k'-k) e) over the columns k' >= k done so far: "the gap is e longer, or restarts here";
            // shifted by (W-1) e below it is max_k (h_k + (W-1-k) e), the carry the strip hands on.  The opening cost
            // and the local-alignment clamp (whose best term is k = W-1) are applied once per row
            a = pk_max(P16_INC(a, Em, E2), h);
            if (CVX) b = pk_max(P16_INC(b, Cm, C2), h);
            SXG_PIN("+v"(Hc[k]), "+v"(a), "+v"(b));
        }
        a = P16_DEC(a, (unsigned)(W - 1) * Em, pk2((W - 1) * e, (W - 1) * e));
        if (CVX) b = P16_DEC(b, (unsigned)(W - 1) * Cm, pk2((W - 1) * c, (W - 1) * c));
        if (SW) { a = pk_max(a, B2); if (CVX) b = pk_max(b, B2); }
        a = P16_DEC(a, Gm, G2);
        if (CVX) b = P16_DEC(b, Qm, Q2);
        // ---- carries (32-bit), inside the wave.  Wave-local strip index u: lo strips 0..63, then hi strips 64..127; what
        // came in through the mailbox is strip u = -1.  y_u = a_u - u*W*e;  E entering strip u = max_{u'<u} y_u' + (u-1)*W*e
        // (the lane's offsets are rebuilt from an opaque copy of the lane every row: hoisted out of the loop
        // they are six more loop-invariant VGPRs, which the allocator spills and reloads per row --
        // and a scratch reload is an in-order vmcnt wait behind every store still in flight)
        int tt = lane;
        asm volatile("" : "+v"(tt));
        const int tWe = __mul24(tt, We), tWc = __mul24(tt, Wc);
        int ya_lo = pk_lo(a) - tWe, ya_hi = pk_hi(a) - tWe - 64 * We;
        int yb_lo = CVX ? pk_lo(b) - tWc : NEG, yb_hi = CVX ? pk_hi(b) - tWc - 64 * Wc : NEG;
        if (!(EXP & 8)) {
        ya_lo = sxg_wave_incl_max(ya_lo); ya_hi = sxg_wave_incl_max(ya_hi);
        if (CVX) { yb_lo = sxg_wave_incl_max(yb_lo); yb_hi = sxg_wave_incl_max(yb_hi); }
        }
        RP_MARK(1);  // pass 1 + in-wave scan
        // what crosses my left edge in this row: {E, Q entering my first column, H of the column left of it, row}
        int in_e = NEG * 2, in_q = NEG * 2, in_h = FLOORV;
        if (wv > 0 && !(EXP & 16)) {
            i32x4 m = mb_in[i & (P16_MBOX - 1)];
            while (__builtin_amdgcn_readfirstlane(m.w) != i) { __builtin_amdgcn_s_sleep(2); m = mb_in[i & (P16_MBOX - 1)]; }
            in_e = __builtin_amdgcn_readfirstlane(m.x) + We;   // (as y of strip u = -1)
            in_q = __builtin_amdgcn_readfirstlane(m.y) + Wc;
            in_h = __builtin_amdgcn_readfirstlane(m.z);
            if (lane == 0) prog[wv] = i;
        }
        RP_MARK(2);  // waiting for the left neighbour
        {
            const int ta = max(__builtin_amdgcn_readlane(ya_lo, 63), in_e), tb = max(__builtin_amdgcn_readlane(yb_lo, 63), in_q);
            ya_lo = max(sxg_wave_shr1(ya_lo, NEG * 2), in_e); ya_hi = max(sxg_wave_shr1(ya_hi, NEG * 2), ta);
            yb_lo = max(sxg_wave_shr1(yb_lo, NEG * 2), in_q); yb_hi = max(sxg_wave_shr1(yb_hi, NEG * 2), tb);
        }
        const int Ein_lo = max(ya_lo + tWe - We, FLOORV);
        const int Ein_hi = max(ya_hi + tWe + 63 * We, FLOORV);
        const int Qin_lo = !CVX ? FLOORV : max(yb_lo + tWc - Wc, FLOORV);
        const int Qin_hi = !CVX ? FLOORV : max(yb_hi + tWc + 63 * Wc, FLOORV);
        int E = pk2(Ein_lo, Ein_hi), Q = pk2(Qin_lo, Qin_hi);

        // ---- pass 2: final H
        int rowmax = SW ? 0 : NEG2;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            if (EXP & 64) { rowmax = pk_max(rowmax, Hc[k] ^ E ^ Q); continue; }
            int h = pk_max(Hc[k], E);
            if (CVX) h = pk_max(h, Q);
            h = pk_max(h, SW ? B2 : NWF2);   // (local: 0; global: P16_NWFLOOR)
            Hc[k] = h;
            if constexpr (W >= 12 || !SW) rowmax = pk_max(rowmax, h);
            E = pk_max(P16_DEC(h, Gm, G2), P16_DEC(E, Em, E2));
            if (CVX) Q = pk_max(P16_DEC(h, Qm, Q2), P16_DEC(Q, Cm, C2));
            // (round 6: no register pin behind a column below W = 12.  The empty asm made the hazard recogniser pad every column
            //  with three s_nop and kept the compiler from reusing h + g and h + q -- computed here for E and Q -- as the opening
            //  halves of the row's outgoing candidates: 15 + 21 VALU instructions and 18 s_nop per row of the W = 11 headline
            //  class, 124 VGPRs instead of 120, no scratch.  The 16-wave classes W = 12, 13 spill without it.)
            if constexpr (W >= 12) SXG_PIN("+v"(Hc[k]), "+v"(E), "+v"(Q), "+v"(rowmax));
        }
        if constexpr (W < 12 && SW) {
            // (round 6) the row's greatest H as a TREE over the columns: accumulated column by column the compiler sank the chain of
            // W dependent packed maxima to the loop's end, one s_nop between each (local alignment: every H is >= the bias, no floor needed)
            if (!(EXP & 64)) {
                int t_[W];
#pragma unroll
                for (int k = 0; k < W; ++k) t_[k] = Hc[k];
#pragma unroll
                for (int n = W; n > 1; n = (n + 1) / 2)
#pragma unroll
                    for (int k = 0; k < n / 2; ++k) t_[k] = pk_max(t_[k], t_[n - 1 - k]);
                rowmax = t_[0];
            }
        }
        // hand my last column to the right neighbour: inside the wave by a lane shift (lane 0's hi strip begins where lane
        // 63's lo strip ends; the column left of its lo strip came in through the mailbox); lane 63's hi strip is the wave's
        // right edge -- E, Q after its last column and its H go into the next wave's mailbox of this row
        const int xh = Hc[W - 1];
        int lh = sxg_wave_shr1(xh, 0);
        {
            const int x63 = __builtin_amdgcn_readlane(xh, 63);
            if (lane == 0) lh = pk2(in_h, pk_lo(x63));
        }
        RP_MARK(3);  // carry combine + pass 2
        if (wv + 1 < NW && !(EXP & 16)) {
            if ((i & (P16_MBOX / 2 - 1)) == 0)   // the slots of the next P16_MBOX / 2 rows: has my reader freed them?
                while (__builtin_amdgcn_readfirstlane(prog[wv + 1]) < i - P16_MBOX / 2) __builtin_amdgcn_s_sleep(2);
            if (lane == 63) mb_out[i & (P16_MBOX - 1)] = i32x4{pk_hi(E), pk_hi(Q), pk_hi(xh), i};
        }
        RP_MARK(4);  // waiting for the right neighbour's progress

        // ---- end cell bookkeeping
        if (EXP & 4) { key_lo ^= (unsigned)rowmax & 1u; }
        else if (SW) {
            // (biased fields: 512 <= field <= 32767, so the keys order as unsigned numbers; a row that merely equals the
            //  strip's best has a smaller lower half and changes nothing: first strictly greatest, decree S4)
            if ((i & 0xffff) == 0) fold_keys(i);
            unsigned inv = 0xffffu - ((unsigned)i & 0xffffu);
            asm volatile("" : "+v"(inv));   // (one VGPR copy: v_perm / v_and_or take one scalar operand each)
            key_lo = max(key_lo, __builtin_amdgcn_perm((unsigned)rowmax, inv, 0x05040100u));
            key_hi = max(key_hi, ((unsigned)rowmax & 0xffff0000u) | inv);
        } else if (flags & ROW_SINK) {
#pragma unroll
            for (int k = 0; k < W; ++k) {
                if (k == kL_lo && (bi_lo < 0 || pk_lo(Hc[k]) > best_lo)) { best_lo = pk_lo(Hc[k]); bi_lo = i; bk_lo = k; }
                if (k == kL_hi && (bi_hi < 0 || pk_hi(Hc[k]) > best_hi)) { best_hi = pk_hi(Hc[k]); bi_hi = i; bk_hi = k; }
            }
        }

        RP_MARK(5);  // hand-over + end-cell bookkeeping
        // ---- outgoing candidates (see p16_pack_row), ring store, band store
        // band of this row: strips [bs0, bs0 + BS); my wave covers lo strips [128 wv, 128 wv + 64) and the
        // hi strips 64 further on.  (wave-uniform tests; the lane test is ONE exec mask around all W stores)
        const int bs0 = band_first_strip(hint, W, BS, T);
        const int w0 = wv << 7;
        const bool band_lo = !(EXP & 1) && (w0 + 63 >= bs0) && (w0 < bs0 + BS);
        const bool band_hi = !(EXP & 1) && (w0 + 127 >= bs0) && (w0 + 64 < bs0 + BS);
        const bool ring = !(EXP & 2) && (flags & ROW_STORE) != 0;
        const __amdgpu_buffer_rsrc_t rs_ring = p16_rsrc((const void*)((SXG_GLOBAL const char*)g_pool + (size_t)(ring && myslot >= 0 ? myslot : 0) * (size_t)RB), RB);
        const __amdgpu_buffer_rsrc_t rs_plane = p16_rsrc((const void*)(g_tb + (size_t)i * (size_t)(SD * BS)), SD * BS * 4);
        const bool in_lo = (unsigned)(w0 + tt - bs0) < (unsigned)BS, in_hi = (unsigned)(w0 + 64 + tt - bs0) < (unsigned)BS;
        const unsigned sl_lo = soff & 0xffffu, sl_hi = soff >> 16;   // strip s lives in slot s mod BS of its row
        // (CB = 2) the H left of my strips as the plane row keeps it: strip 0 has no left neighbour -- its own first column,
        // whose step is then 0
        int lhs = lh;
        if (CB == 2 && t == 0) lhs = (int)(((unsigned)lh & 0xffff0000u) | ((unsigned)Hc[0] & 0x0000ffffu));
// (round 6) a lane whose strip lies outside the row's band stores to a slot beyond the row's buffer descriptor -- the hardware
// drops the access -- instead of sitting out a lane-divergent branch around the stores: the structurised branch made the
// compiler keep two sets of the 2 W gap-state registers and copy between them on every row (22 to 66 v_mov per row).
// ring row + band cells of this row; CF(k) / CO(k) = the row's outgoing candidates of column k
#define P16_STORES(CF, CO)                                                                                  \
    do {                                                                                                    \
        if (ring) {                                                                                         \
            if (myslot <= -2) {                                                                             \
                P16_LDS_ROW(la_, lf_, -2 - myslot);                                                         \
                _Pragma("unroll") for (int k = 0; k < W; ++k) la_[k * 64] = p16_pack_row<CVX, SW>(Hc[k], CF, CO); \
                *lf_ = (unsigned)lh;                                                                        \
            } else if (!ring_plane) {                                                                       \
                _Pragma("unroll") for (int k = 0; k < W; ++k)                                               \
                    __builtin_amdgcn_raw_buffer_store_b64(p16_pack_row<CVX, SW>(Hc[k], CF, CO), rs_ring, ut8, k * T * 8, 0); \
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)lh, rs_ring, ut8 >> 1, TW * 8, 0);          \
            }                                                                                               \
        }                                                                                                   \
        if (CB == 2) {                                                                                      \
            if (band_lo || band_hi) {                                                                       \
                /* delta codes of my two strips (see P16Delta): both halves of a register at once */       \
                int code_[W], prev_ = lhs;                                                                  \
                _Pragma("unroll") for (int k = 0; k < W; ++k) {                                             \
                    const int h_ = Hc[k], of_ = (CF), oo_ = (CO);                                           \
                    const int d2_ = SW ? (int)((unsigned)h_ - (unsigned)of_) : pk_sub(h_, of_);             \
                    int cd_ = pk_mad(d2_, KF2, pk_sub(h_, prev_));                                          \
                    if (CVX) cd_ = pk_mad(SW ? (int)((unsigned)h_ - (unsigned)oo_) : pk_sub(h_, oo_), KO2, cd_); \
                    code_[k] = cd_;                                                                         \
                    prev_ = h_;                                                                             \
                }                                                                                           \
                /* dword x of a strip: halfwords 2x, 2x + 1 of (left H, code 0, ..., code W-1) */            \
                if (band_lo)                                                                                \
                    plane_store_strip<SD>(rs_plane, in_lo ? sl_lo : P16_SLOT_OOB, BS, [&](const int x) -> unsigned { \
                        return __builtin_amdgcn_perm((unsigned)(2 * x < W ? code_[2 * x < W ? 2 * x : 0] : 0), (unsigned)(x ? code_[x ? 2 * x - 1 : 0] : lhs), 0x05040100u); }); \
                if (band_hi)                                                                                \
                    plane_store_strip<SD>(rs_plane, in_hi ? sl_hi : P16_SLOT_OOB, BS, [&](const int x) -> unsigned { \
                        return __builtin_amdgcn_perm((unsigned)(2 * x < W ? code_[2 * x < W ? 2 * x : 0] : 0), (unsigned)(x ? code_[x ? 2 * x - 1 : 0] : lhs), 0x07060302u); }); \
            }                                                                                               \
        } else {                                                                                            \
        if (band_lo) {                                                                                      \
                plane_store_strip<W>(rs_plane, in_lo ? sl_lo : P16_SLOT_OOB, BS, [&](const int k) -> unsigned { \
                    const u32x2 w = p16_pack_row<CVX, SW>(Hc[k], CF, CO);                                   \
                    return __builtin_amdgcn_perm(w.y, w.x, 0x05040100u); });                                \
        }                                                                                                   \
        if (band_hi) {                                                                                      \
                plane_store_strip<W>(rs_plane, in_hi ? sl_hi : P16_SLOT_OOB, BS, [&](const int k) -> unsigned { \
                    const u32x2 w = p16_pack_row<CVX, SW>(Hc[k], CF, CO);                                   \
                    return __builtin_amdgcn_perm(w.y, w.x, 0x07060302u); });                                \
        }                                                                                                   \
        }                                                                                                   \
    } while (0)
#pragma unroll
        for (int k = 0; k < W; ++k) {
            Fp[k] = pk_max(P16_DEC(Hc[k], Gm, G2), P16_DEC(Fp[k], Em, E2));
            if (CVX) Op[k] = pk_max(P16_DEC(Hc[k], Qm, Q2), P16_DEC(Op[k], Cm, C2));
            SXG_PIN("+v"(Fp[k]), "+v"(Op[k]));
        }
        P16_STORES(Fp[k], Op[k]);
#undef P16_STORES
#undef P16_LDS_ROW
#pragma unroll
        for (int k = 0; k < W; ++k) Hp[k] = Hc[k];
        Hleft = lh;
        RP_MARK(7);  // outgoing candidates + stores
    }
#ifdef SXG_ROW_PROF
    if (t == 0 && B.row_prof)
        for (int k = 0; k < 8; ++k) B.row_prof[k] += racc[k];
#endif
#undef RP_MARK
#undef P16_DEC
#undef P16_INC

    if (SW) {
        // ---- end cell of a local alignment: the wave's keys, then the workgroup's; the column is found by the traceback
        fold_keys(N + 1);
        __syncthreads();
        unsigned long long* kl = (unsigned long long*)lds;
        if (lane == 0) kl[wv] = ekey;
        __syncthreads();
        unsigned long long key = kl[0];
        for (int x = 1; x < NW; ++x) key = kl[x] > key ? kl[x] : key;
        __syncthreads();
        const int val = (int)(unsigned)(key >> 32) - BIAS;
        if (key == 0ull || val <= 0) { res.best = 0; res.bi = -1; res.bj = -1; }
        else {
            res.best = val;
            res.bi = (int)(0xFFFFFu - (unsigned)((key >> 12) & 0xFFFFFu));
            res.bj = -1 - (int)(0xFFFu - (unsigned)(key & 0xFFFu));   // -(strip + 1): see traceback_p16
        }
        return res;
    }
    // ---- end cell: greatest score, then smallest row, then smallest column (two candidates per lane)
    unsigned long long key = 0;
    if (bi_lo >= 0)
        key = ((unsigned long long)(unsigned)(best_lo + (1 << 27)) << 35) |
              ((unsigned long long)(0xFFFFFu - (unsigned)bi_lo) << 15) | (unsigned long long)(0x7FFFu - (unsigned)(j0 + bk_lo));
    if (bi_hi >= 0) {
        const unsigned long long k2 = ((unsigned long long)(unsigned)(best_hi + (1 << 27)) << 35) |
                                      ((unsigned long long)(0xFFFFFu - (unsigned)bi_hi) << 15) |
                                      (unsigned long long)(0x7FFFu - (unsigned)(j0h + bk_hi));
        key = k2 > key ? k2 : key;
    }
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
        res.best = (int)(unsigned)(key >> 35) - (1 << 27) - BIAS;
        res.bi = (int)(0xFFFFFu - (unsigned)((key >> 15) & 0xFFFFFu));
        res.bj = (int)(0x7FFFu - (unsigned)(key & 0x7FFFu));
    }
    return res;
}

// ---------------------------------------------------------------------------------------------------
// Traceback of the packed sweep: the alignment is DERIVED from the plane's values (rules: header of this
// file and oracle/poa_vtb.c).  The walk is a chain of dependent reads, so the whole of wave 0 takes part:
// lane l fetches, in two round trips, what a step through row (top - l) can need -- the row descriptor,
// the node id and the plane cells of 8 columns around the column where the alignment is expected to cross
// that row (L/N columns per graph row) -- into an LDS window of 64 rows; the walk itself runs out of LDS on
// the scalar unit (every lane executes it redundantly, lane 0 writes) and returns to HBM only for a cell
// the window does not hold (an indel run, a predecessor far up the order) and for the leftward scans that
// resolve a gap in the graph (E/Q), which all 64 lanes do together, 64 columns per round trip.
constexpr int TBW_ROWS = 64;
constexpr int TBW_STRIDE = 13;  // dwords per window row (odd: conflict-free fills)
constexpr int TBW_COLS = 8;     // plane cells per window row
// window row: TBW_COLS cells, then
enum : int { EO_PB = 8, EO_INFO = 9, EO_Q0 = 10, EO_Q1 = 11, EO_NODE = 12 /* node | valid-cell mask << 24 */ };
constexpr int TBW_LET = 128;    // query letters of columns jtop, jtop-1, ... kept beside the window
static_assert((TBW_ROWS * TBW_STRIDE + TBW_LET) * 4 <= LDS_META_BYTES / 2, "traceback window lives in the descriptor area");

// words of LDS (control area) through which the walk reports a band miss to the workgroup
enum : int { TBM_FLAG = 208, TBM_ROW = 209, TBM_DELTA = 210, TBM_RANGE = 211 /* a clamped cell on the walk: see P16_NWFLOOR */ };

// BANDED (poa_band16.hip.h): a row keeps exactly the strips of its band, max(0, hint - w) / W .. (hint + w) / W, and a
// cell outside the band does not exist (it reads as -inf and is never a miss).
// CB = 2: the plane holds delta codes (P16Delta); the helpers below rebuild the cells the walk asks for -- H by summing a
// strip's steps from its left end -- in the round-4 word format (H | H - oF | H - oO), so the walk itself is the same code.
// j < 0 on entry: the sweep of a local alignment names the STRIP of the end cell, -(strip + 1); the column is the first
// of that strip's cells in row i that holds the best score.
// STRICT: the caller admitted the alignment under the strict range rule (every cell above P16_NWFLOOR: nothing is ever clamped,
// sxg_poa.hip::p16_safe without clamp_ok -- the align-only kernel, which has no wider re-run): the walk takes every cell at its word.
template <bool PAIRS, int W, bool CVX, bool BANDED = false, int CB = 4, bool STRICT = false>
// (views by value: a reference to the kernel's private copy trips an AMDGPU back-end assertion on
// the private-aperture null check for some strip widths)
__device__ __noinline__ int traceback_p16(const RowsView R, const DpBuffers B, const Scoring S_, const uint8_t* seq, const int L_,
                                          const int best_, const int T_, const int slope_, int i, int j, int32_t* posnode,
                                          int32_t* pair_row, int32_t* pair_pos, char* smem) {
    // arguments arrive in vector registers: make the uniform ones scalar again
    const int T = __builtin_amdgcn_readfirstlane(T_), best = __builtin_amdgcn_readfirstlane(best_);
    const int L = __builtin_amdgcn_readfirstlane(L_);
    // columns the alignment advances per graph row, in 1/256: a graph of N rows against L letters is
    // walked at about L/N columns per row (rows of other branches are skipped), which is where the
    // window is laid
    const int slope = __builtin_amdgcn_readfirstlane(slope_);
    i = __builtin_amdgcn_readfirstlane(i); j = __builtin_amdgcn_readfirstlane(j);
    const int sm = __builtin_amdgcn_readfirstlane(S_.m), sn = __builtin_amdgcn_readfirstlane(S_.n);
    const int g = __builtin_amdgcn_readfirstlane(S_.g), e = __builtin_amdgcn_readfirstlane(S_.e);
    const int q = __builtin_amdgcn_readfirstlane(S_.q), c = __builtin_amdgcn_readfirstlane(S_.c);
    const int sw = __builtin_amdgcn_readfirstlane(S_.sw);
    // the full-matrix sweep of a local alignment stores biased H (P16_BIAS): "zero" is the bias
    const int bias = (!BANDED && sw) ? P16_BIAS : 0;
    const int BS = __builtin_amdgcn_readfirstlane(B.band_strips);
    const int bw = __builtin_amdgcn_readfirstlane(B.band_w), last_strip = L / W;
    // adaptive band (B4): the sweep left every row's band -- first | last strip << 16 -- in word 6 of its descriptor
    const bool ada = BANDED && __builtin_amdgcn_readfirstlane(B.band_mode) == 2;
    const int HW = ada ? 6 : 7;   // descriptor word that says which strips the row kept
    // The adaptive sweep rewrote words 6, 7 of every row descriptor with device-scope (sc1) stores; the chunk the sweep
    // staged earlier may still sit in this CU's L1, and __syncthreads does not invalidate it: drop the L1 copies once per
    // walk (buffer_inv sc1) so that the plain loads below see the band words the sweep left.
    if (ada) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // is strip s of a row with band hint `hint` kept in the plane?
    auto kept = [&](int hint, int s) -> bool {
        if (BANDED) {
            if (ada) return s >= (hint & 0xffff) && s <= (int)((unsigned)hint >> 16);
            return s >= max(hint - bw, 0) / W && s <= min((hint + bw) / W, last_strip);
        }
        return (unsigned)(s - band_first_strip(hint, W, BS, T)) < (unsigned)BS;
    };
    // (BANDED with CB = 2 -- round 6, local alignment: every strip starts with the H of its OWN first column and a step of 0, the
    //  packed sweep's "strip 0" form, so the decoders below need nothing of a strip's left neighbour; strips are ABSOLUTE there)
    constexpr int SD = p16_slot_dwords(W, CB);     // dwords of one strip in a plane row
    // (CB = 2) fields of a cell's code
    const P16Delta DF = p16_delta_of(S_);
    const int dbH = __builtin_amdgcn_readfirstlane(DF.bH), dbF = __builtin_amdgcn_readfirstlane(DF.bF);
    const unsigned dmH = (1u << dbH) - 1u, dmF = (1u << dbF) - 1u;
    const int dG = __builtin_amdgcn_readfirstlane(DF.g), dE = __builtin_amdgcn_readfirstlane(DF.eabs), dC = __builtin_amdgcn_readfirstlane(DF.cabs);
    // The sweep adds the raw step and distances into the code; taking their least values off first (mod 2^16) leaves three
    // non-negative fields side by side.
    const unsigned dCst = (unsigned)(dG + (dE << dbH) + ((CVX ? dC : 0) << (dbH + dbF))) & 0xffffu;
    auto d_norm = [&](const unsigned raw16) -> unsigned { return (raw16 - dCst) & 0xffffu; };
    // normalised code -> step of H; the cell word given H
    auto d_step = [&](const unsigned code) -> int { return (int)(code & dmH) + dG; };
    auto d_word = [&](const int h, const unsigned code) -> uint32_t {
        return ((uint32_t)h & 0xffffu) | ((((code >> dbH) & dmF) + (unsigned)dE) << 16) | (((code >> (dbH + dbF)) + (unsigned)(CVX ? dC : 0)) << 24);
    };
    // halfword hw (0 = the H left of the strip, 1 + k = code of column k) of a strip whose dwords are d[0..SD)
    auto d_half = [&](const unsigned (&d)[SD], const int hw) -> unsigned { return (hw & 1) ? d[hw >> 1] >> 16 : d[hw >> 1] & 0xffffu; };
    // outputs and letters through global pointers: a FLAT store also counts on lgkmcnt, and the walk
    // waits on lgkmcnt for its LDS reads every step -- i.e. it would wait for the previous step's store to
    // reach HBM
    SXG_GLOBAL int32_t* const g_posnode = sxg_global(posnode);
    SXG_GLOBAL int32_t* const g_pair_row = sxg_global(pair_row);
    SXG_GLOBAL int32_t* const g_pair_pos = sxg_global(pair_pos);
    SXG_GLOBAL const uint8_t* const g_seq = sxg_global(seq);
    SXG_GLOBAL const uint32_t* const g_plane = sxg_global((const uint32_t*)B.tb);
    SXG_GLOBAL const int32_t* const g_meta = sxg_global((const int32_t*)R.meta);
    SXG_GLOBAL const int32_t* const g_preds = sxg_global((const int32_t*)R.preds);
    SXG_GLOBAL const int32_t* const g_row_node = sxg_global((const int32_t*)R.row_node);
    const int lane = threadIdx.x & 63;
    int* ctl = (int*)smem;
    // the block's only serial phase (the other waves wait for this one): a dependent-read chain that
    // rarely has an instruction ready, so top priority costs the co-residents next to nothing
    __builtin_amdgcn_s_setprio(3);
    uint32_t* win = (uint32_t*)(smem + LDS_CTL_BYTES);
    const int WR = TBW_ROWS;  // window rows
    uint32_t* wlet = win + TBW_ROWS * TBW_STRIDE;  // [TBW_LET] query letters of columns jtop, jtop-1, ...
    // Diagonal steps are precomputed for the whole window by all 64 lanes (lane l: the 8 cells of row wtop - l): a word
    // per cell saying "from here the walk takes a D step to window cell (l', x')" -- or 0 when the cell needs the
    // general code below (a gap, three or more predecessors, a cell the window does not hold, the end of a local
    // alignment).  The serial part of a run of matches/mismatches is then one LDS read per step instead of ~150
    // scalar instructions, and its outputs are written by 64 lanes at once.  Lives in the parked-row area of the
    // sweep (free during the walk).
    uint32_t* trans = (uint32_t*)(smem + LDS_CTL_BYTES + dp16_meta_bytes(T));   // [TBW_ROWS][TBW_COLS]
    uint32_t* chain = trans + TBW_ROWS * TBW_COLS;                               // [TBW_ROWS] cells of the current run
    int* whint = (int*)(chain + TBW_ROWS);                                        // [TBW_ROWS] backbone hint of every window row (the last 256 of the area's 2 560 bytes)
    const int kmax_e = CVX ? 1 + (g - q) / (c - e) : 0x7fffffff;  // longest gap the first piece can win (poa_vtb.c)
#define TBU(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
    // H of the virtual row 0
    auto h_row0 = [&](int col) -> int {
        if (sw || col <= 0) return bias;
        const int a = g + (col - 1) * e, b = q + (col - 1) * c;
        return max(a > b ? a : b, BANDED ? NEGP : P16_NWFLOOR);
    };
    // global alignment on the full matrix: cells at or below `thr` may be clamped ones (see P16_NWFLOOR); the walk must not
    // take a decision in one
    const int thr = (!BANDED && !STRICT && !sw) ? P16_NWFLOOR + sm * L + sm : -0x40000000;
    bool range = false;
    int wtop = -1, wj = 0;
    bool miss = false;
    int miss_row = 0, miss_delta = 0;
    // plane cell (row p >= 1, column col) straight from HBM; a cell outside the row's band is a miss
    auto gcell = [&](int p, int col) -> uint32_t {
        const int hint = (int)TBU(g_meta[8 * (size_t)(p - 1) + HW]);
        const int s = col / W, k = col - s * W;
        if (!kept(hint, s)) {
            if (BANDED) return 0x0000C000u;   // (NEGCELL: H = -inf)
            if (!miss) { miss = true; miss_row = p; miss_delta = col - hint; }
            return 0u;
        }
        if constexpr (CB == 2) {
            // lanes 0 .. SD-1 fetch the strip's dwords; the steps up to column k are summed on the scalar unit
            const unsigned dw = lane < SD ? g_plane[(size_t)p * (size_t)(SD * BS) + (size_t)plane_cell_in_row(SD, BS, s % BS, min(lane, SD - 1))] : 0u;
            int h = (int)(short)(TBU(__builtin_amdgcn_readlane((int)dw, 0)) & 0xffffu);
            unsigned cd = 0;
            for (int t2 = 0; t2 <= k; ++t2) {
                const unsigned d = TBU(__builtin_amdgcn_readlane((int)dw, (t2 + 1) >> 1));
                cd = d_norm(((t2 + 1) & 1) ? d >> 16 : d & 0xffffu);
                h += d_step(cd);
            }
            return d_word(h, cd);
        }
        return TBU(g_plane[plane_cell<W>((size_t)p, BS, s % BS, k)]);
    };
    int wgeo = slope;   // slope the current window was laid out with
    auto wcol0 = [&](int l) -> int { return wj - ((l * wgeo) >> 8) - (TBW_COLS - 3); };  // first column the window holds of row wtop-l
    auto cell = [&](int p, int col) -> uint32_t {
        const int l = wtop - p;
        if (wtop >= 0 && (unsigned)l < (unsigned)WR) {
            const int x = col - wcol0(l);
            if ((unsigned)x < (unsigned)TBW_COLS && ((TBU(win[l * TBW_STRIDE + EO_NODE]) >> (24 + x)) & 1u))
                return TBU(win[l * TBW_STRIDE + x]);
        }
        return gcell(p, col);
    };
    auto sext = [](uint32_t w) -> int { return (int)(short)(w & 0xffffu); };
    // ---- window fills in two halves (round 5): win_issue sends a window's loads (lane l: row top - l -- descriptor, node id, the
    // plane cells around the column where the walk is expected to cross that row, and two query letters), win_commit decodes
    // them into the LDS window and precomputes the diagonal runs.  The NEXT window is issued as soon as the current one is
    // committed, at the place the running slope predicts, and committed without a round trip when the walk enters it there
    // (rounds 2-4: every window was two dependent round trips, half of the walk's time on 1 kbp blocks).
    constexpr int NSW = CB == 2 ? (W >= 8 ? 2 : 3) : 1;            // strips a lane fetches (CB = 2: whole strips)
    constexpr int NRAW = CB == 2 ? NSW * SD : TBW_COLS;
    struct WinRaw { i32x4 d0, d1; int node; unsigned cells[NRAW]; uint32_t let0, let1; };
    auto win_issue = [&](const int top, const int jj, const int slp, WinRaw& R_) {
        const int row = top - lane;
        R_.d0 = i32x4{0, 0, 0, 0}; R_.d1 = R_.d0; R_.node = 0;
#pragma unroll
        for (int x2 = 0; x2 < NRAW; ++x2) R_.cells[x2] = 0u;
        if (row >= 1) {
            SXG_GLOBAL const i32x4* dm = (SXG_GLOBAL const i32x4*)(g_meta + 8 * (size_t)(row - 1));
            R_.d0 = dm[0]; R_.d1 = dm[1];
            R_.node = g_row_node[row - 1];
            const int c0 = jj - ((lane * slp) >> 8) - (TBW_COLS - 3);
            // (a cell's address does not depend on the row's band start: cells and descriptor travel together)
            if constexpr (CB == 2) {
                const int s0w = max(c0, 0) / W;
#pragma unroll
                for (int si = 0; si < NSW; ++si) {
                    unsigned tmp[SD];
                    plane_load_slot<SD>(g_plane + (size_t)row * (size_t)(SD * BS), BS, min(s0w + si, BANDED ? last_strip : 2 * T - 1) % BS, tmp);
#pragma unroll
                    for (int x2 = 0; x2 < SD; ++x2) R_.cells[si * SD + x2] = tmp[x2];
                }
            } else {
#pragma unroll
                for (int x2 = 0; x2 < TBW_COLS; ++x2) {
                    const int col = min(max(c0 + x2, 0), L);
                    const int s = col / W, k = col - s * W;
                    R_.cells[x2] = g_plane[plane_cell<W>((size_t)row, BS, s % BS, k)];
                }
            }
        }
        // (letters: one per column; the walk never reads them from HBM -- a global load inside a step
        // makes the compiler wait for vmcnt(0) there, i.e. for the previous step's output store)
        R_.let0 = (jj - lane >= 1) ? (uint32_t)g_seq[jj - lane - 1] : 255u;
        R_.let1 = (jj - 64 - lane >= 1) ? (uint32_t)g_seq[jj - 64 - lane - 1] : 255u;
    };
    // (the window's geometry -- wtop, wj, wgeo -- is set by the caller before the commit)
    auto win_commit = [&](const WinRaw& R_) {
        const int row = wtop - lane;
        if (row >= 1 && lane < WR) {
            const i32x4 d0 = R_.d0, d1 = R_.d1;
            const int c0 = wcol0(lane);
            uint32_t v[TBW_COLS], valid = 0;
            uint32_t* en = win + lane * TBW_STRIDE;
            if constexpr (CB == 2) {
                // whole strips of the row, from the one that holds the window's first column: every strip is decoded from
                // its left end and each cell that falls into the window goes straight to its place in my window row
                const int s0w = max(c0, 0) / W;
#pragma unroll
                for (int x2 = 0; x2 < TBW_COLS; ++x2) v[x2] = 0u;
#pragma unroll
                for (int si = 0; si < NSW; ++si) {
                    int h = (int)(short)(R_.cells[si * SD] & 0xffffu);
                    const int xb = (s0w + si) * W - c0;   // window place of the strip's first column
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        const unsigned dwk = R_.cells[si * SD + ((1 + k) >> 1)];
                        const unsigned cd = d_norm(((1 + k) & 1) ? dwk >> 16 : dwk & 0xffffu);
                        h += d_step(cd);
                        if ((unsigned)(xb + k) < (unsigned)TBW_COLS && s0w + si <= (BANDED ? last_strip : 2 * T - 1)) en[xb + k] = d_word(h, cd);
                    }
                }
            } else {
#pragma unroll
                for (int x2 = 0; x2 < TBW_COLS; ++x2) v[x2] = R_.cells[x2];
            }
#pragma unroll
            for (int x2 = 0; x2 < TBW_COLS; ++x2) {
                const int col = c0 + x2;
                if (col >= 0 && col <= L) {
                    if (kept(ada ? d1.z : d1.w, col / W)) valid |= 1u << x2;
                    else if (BANDED) { if constexpr (CB == 2) en[x2] = 0x0000C000u; else v[x2] = 0x0000C000u; valid |= 1u << x2; }
                }
            }
            if constexpr (CB != 2) {
#pragma unroll
                for (int x2 = 0; x2 < TBW_COLS; ++x2) en[x2] = v[x2];
            }
            en[EO_PB] = (uint32_t)d0.x; en[EO_INFO] = (uint32_t)d0.y; en[EO_Q0] = (uint32_t)d0.z; en[EO_Q1] = (uint32_t)d1.x;
            en[EO_NODE] = (uint32_t)R_.node | (valid << 24);
            whint[lane] = d1.w;
        }
        wlet[lane] = R_.let0;
        wlet[64 + lane] = R_.let1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {   // D transitions of my row's cells (rule D of poa_vtb.c, rows with one or two predecessors)
            const int row2 = wtop - lane;
            uint32_t tw[TBW_COLS];
#pragma unroll
            for (int x2 = 0; x2 < TBW_COLS; ++x2) tw[x2] = 0u;
            if (row2 >= 1 && lane < WR) {
                const uint32_t* me = win + lane * TBW_STRIDE;
                const uint32_t nv = me[EO_NODE];
                const int inf2 = (int)me[EO_INFO], np2 = inf2 & 0xffff, code2 = (inf2 >> 16) & 0xff;
                const int pa = (int)me[EO_Q0], pbb = (int)me[EO_Q1];
                const int c0 = wcol0(lane);
                if (np2 >= 1 && np2 <= 2 && pa >= 1 && (np2 == 1 || pbb >= 1)) {
                    const int la = wtop - pa, lb = np2 == 2 ? wtop - pbb : la;
                    if ((unsigned)la < (unsigned)WR && (unsigned)lb < (unsigned)WR) {
                        const int ca = wcol0(la), cb = wcol0(lb);
                        const uint32_t nva = win[la * TBW_STRIDE + EO_NODE], nvb = win[lb * TBW_STRIDE + EO_NODE];
#pragma unroll
                        for (int x2 = 0; x2 < TBW_COLS; ++x2) {
                            const int col = c0 + x2;
                            const int xa = col - 1 - ca, xb = col - 1 - cb, lo = wj - col;
                            if (!((nv >> (24 + x2)) & 1u) || col < 1 || (unsigned)xa >= (unsigned)TBW_COLS || (unsigned)xb >= (unsigned)TBW_COLS ||
                                (unsigned)lo >= (unsigned)TBW_LET) continue;
                            if (!((nva >> (24 + xa)) & 1u) || !((nvb >> (24 + xb)) & 1u)) continue;
                            const int hcell = sext(me[x2]);
                            if ((sw && hcell == bias) || hcell <= thr) continue;
                            const int ha = sext(win[la * TBW_STRIDE + xa]), hb = np2 == 2 ? sext(win[lb * TBW_STRIDE + xb]) : ha;
                            const bool second = np2 == 2 && hb > ha;    // (first predecessor in list order on ties)
                            const int best2 = second ? hb : ha;
                            if (best2 + ((int)wlet[lo] == code2 ? sm : sn) == hcell)
                                tw[x2] = 0x80000000u | (uint32_t)(second ? lb : la) | ((uint32_t)(second ? xb : xa) << 6);
                        }
                    }
                }
            }
            if (lane < WR) {
#pragma unroll
                for (int x2 = 0; x2 < TBW_COLS; ++x2) trans[lane * TBW_COLS + x2] = tw[x2];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    WinRaw pf;                     // the window asked for ahead of the walk
    int pf_top = -1, pf_j = 0, pf_slope = 0;
    int wslope = slope;            // running estimate of the columns the walk advances per row (1/256)
    int n = 0;
    int st = SRC_STOP;           // SRC_STOP = "in H", SRC_F / SRC_O = walking up a gap in the sequence
    int hv = best + bias, gv = 0;       // H of the current cell / value of the gap state being walked
#ifdef SXG_ROW_PROF
    unsigned long long tb_steps = 0, tb_loads = 0, tb_t0 = __builtin_readcyclecounter(), tb_ld = 0, tb_hits = 0;
#endif
    if (j < 0) {
        // The sweep of a local alignment found the end cell's row and STRIP (dp_fill_p16): its column is the first cell of
        // that strip, in row i, that holds the score.  A strip the row did not keep is a band miss like any other.
        const int s = -1 - j;
        const int hint = (int)TBU(g_meta[8 * (size_t)(i - 1) + HW]);
        j = 0;
        if (!kept(hint, s)) { miss = true; miss_row = i; miss_delta = s * W + W / 2 - hint; }
        else {
            int kf = -1;
            if constexpr (CB == 2) {
                const unsigned dw = lane < SD ? g_plane[(size_t)i * (size_t)(SD * BS) + (size_t)plane_cell_in_row(SD, BS, s % BS, min(lane, SD - 1))] : 0u;
                int h = (int)(short)(TBU(__builtin_amdgcn_readlane((int)dw, 0)) & 0xffffu);
                for (int t2 = 0; t2 < W && kf < 0; ++t2) {
                    const unsigned d = TBU(__builtin_amdgcn_readlane((int)dw, (t2 + 1) >> 1));
                    h += d_step(d_norm(((t2 + 1) & 1) ? d >> 16 : d & 0xffffu));
                    if (h == hv) kf = t2;
                }
            } else {
                const uint32_t cw = lane < W ? g_plane[plane_cell<W>((size_t)i, BS, s % BS, min(lane, W - 1))] : 0u;
                const unsigned long long meq = __ballot(lane < W && sext(cw) == hv);
                if (meq) kf = (int)__builtin_ctzll(meq);
            }
            if (kf < 0) { miss = true; miss_row = i; miss_delta = 0; }   // (cannot happen: the strip's key came from these cells)
            else j = s * W + kf;
        }
    }
    while (!miss) {
#ifdef SXG_ROW_PROF
        ++tb_steps;
#endif
        if (i == 0) {
            if (j == 0 || sw) break;
            if (PAIRS && lane == 0) { g_pair_row[n] = 0; g_pair_pos[n] = j - 1; }
            ++n; --j;
            continue;
        }
        if (sw && st == SRC_STOP && hv == bias) break;
        if (st == SRC_STOP && hv <= thr) { range = true; break; }
        {   // (re)fill the window when row i or column j leave it
            const int l = wtop - i;
            const int x = j - wcol0(l);
            if (wtop < 0 || l < 0 || l > WR - 3 || x < 2 || x >= TBW_COLS) {
#ifdef SXG_ROW_PROF
                ++tb_loads;
                const unsigned long long tl0 = __builtin_readcyclecounter();
#endif
                // the slope the walk really had across the window it leaves (rows of other branches are skipped, so it differs from
                // L / N locally), blended into the running estimate
                if (wtop >= 0 && wtop > i) {
                    const int est = ((wj - j) << 8) / (wtop - i);
                    wslope = min(max((wslope + est) >> 1, 16), 512);
                }
                // the window asked for while the walk was still in the last one, if the walk enters it where it was expected
                bool hit = false;
                if (pf_top >= 1) {
                    const int l2 = pf_top - i;
                    const int x2 = j - (pf_j - ((l2 * pf_slope) >> 8) - (TBW_COLS - 3));
                    hit = l2 >= 0 && l2 <= WR / 2 && x2 >= 3 && x2 < TBW_COLS - 1;
                }
#ifdef SXG_ROW_PROF
                if (hit) ++tb_hits;
#endif
                if (hit) { wtop = pf_top; wj = pf_j; wgeo = pf_slope; win_commit(pf); }
                else {
                    wtop = i; wj = j; wgeo = wslope;
                    WinRaw cur;
                    win_issue(wtop, wj, wgeo, cur);
                    win_commit(cur);
                }
                // ... and ask for the next one right away: its loads travel while the walk crosses this window
                // Where will the walk enter it?  Every row's descriptor carries the DP column its node is expected to align at (the
                // backbone hint that centres the plane's band): the walk runs at a nearly constant offset from those -- it changes
                // with this sequence's own indels only --, and the next window's top row is a row of this window.  (The adaptive
                // band keeps another quantity in that word: there the running slope predicts.)
                pf_top = wtop - (WR - 6); pf_slope = wslope; pf_j = max(wj - (((WR - 6) * wslope) >> 8), 0);
                if (!ada && pf_top >= 1 && wtop - i >= 0 && wtop - i < WR)
                    pf_j = min(max((int)TBU(whint[WR - 6]) + (j - (int)TBU(whint[wtop - i])), 0), L);
#ifdef SXG_TB_NO_PREFETCH
                pf_top = -1;   // (A/B builds: every window is fetched when the walk needs it, as in rounds 2-4)
#endif
                // (Measured, round 5: the walk enters the predicted window in only 16-22 % of the cases -- rows of other branches make
                //  its path through rank space too irregular for a straight line eight columns wide -- which pays on one- and
                //  two-wave workgroups, where nothing else hides the round trips (16 x 1 kbp: 43.2 -> 42.1 ms), and costs 1.6 % on the
                //  four-wave headline class, whose other waves do: those keep fetching on demand.)
#ifndef SXG_TB_PREFETCH_ALL
                if (T > 128) pf_top = -1;
#endif
                if (pf_top >= 1) win_issue(pf_top, pf_j, pf_slope, pf); else pf_top = -1;
#ifdef SXG_ROW_PROF
                tb_ld += __builtin_readcyclecounter() - tl0;
#endif
            }
        }
        if (st == SRC_STOP) {   // a run of precomputed diagonal steps
            int cl = wtop - i, cx = j - wcol0(cl), ns = 0;
            for (;;) {
                const uint32_t tr = TBU(trans[cl * TBW_COLS + cx]);
                if (!(tr >> 31)) break;
                if (lane == 0) chain[ns] = (uint32_t)cl | ((uint32_t)cx << 6);
                ++ns;
                cl = (int)(tr & 63u); cx = (int)((tr >> 6) & 7u);
            }
            if (ns) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < ns) {   // the run's outputs, one lane per step
                    const uint32_t ce = chain[lane];
                    const int l2 = (int)(ce & 63u), col = wcol0(l2) + (int)(ce >> 6);
                    if (PAIRS) { g_pair_row[n + lane] = wtop - l2; g_pair_pos[n + lane] = col - 1; }
                    if (posnode) g_posnode[col - 1] = (int)(win[l2 * TBW_STRIDE + EO_NODE] & 0x00ffffffu);
                }
                n += ns;
                i = wtop - cl; j = wcol0(cl) + cx;
                hv = sext(TBU(win[cl * TBW_STRIDE + cx]));
#ifdef SXG_ROW_PROF
                tb_steps += (unsigned long long)(ns - 1);
#endif
                continue;
            }
        }
        // Everything the walk reads is the same for all lanes; saying so (readfirstlane) keeps its state
        // in scalar registers and its control flow on the scalar unit instead of 64-wide selects and
        // exec-mask juggling.
        const uint32_t* en = win + (wtop - i) * TBW_STRIDE;
        const int pb = (int)TBU(en[EO_PB]), info = (int)TBU(en[EO_INFO]), q0 = (int)TBU(en[EO_Q0]), q1 = (int)TBU(en[EO_Q1]);
        const int node = (int)(TBU(en[EO_NODE]) & 0x00ffffffu);
        const int np = info & 0xffff, npe = np > 0 ? np : 1;
        auto pred_at = [&](int x) -> int {
            if (np == 0) return 0;
            return x == 0 ? q0 : (x == 1 ? q1 : (int)TBU(g_preds[pb + x]));
        };
        if (st == SRC_STOP) {
            bool done = false;
            if (j >= 1) {   // D: the first predecessor (list order) with the greatest H[p][j-1]
                int dmax = -0x40000000, dp = 0;
                for (int x = 0; x < npe; ++x) {
                    const int p = pred_at(x);
                    const int hp = p == 0 ? h_row0(j - 1) : sext(cell(p, j - 1));
                    if (hp > dmax) { dmax = hp; dp = p; }
                }
                const int lo = wj - j;
                const int letter = (unsigned)lo < (unsigned)TBW_LET ? (int)TBU(wlet[lo]) : (int)TBU(g_seq[j - 1]);
                if (!miss && dmax + (letter == ((info >> 16) & 0xff) ? sm : sn) == hv) {
                    if (lane == 0) {
                        if (PAIRS) { g_pair_row[n] = i; g_pair_pos[n] = j - 1; }
                        if (posnode) g_posnode[j - 1] = node;
                    }
                    ++n;
                    i = dp; --j; hv = dmax;
                    done = true;
                }
            }
            if (!done && !miss) {   // F, then O: a predecessor's outgoing candidate equals H
                int fmax = -0x40000000, omax = -0x40000000;
                for (int x = 0; x < npe; ++x) {
                    const int p = pred_at(x);
                    int of, oo;
                    if (p == 0) { const int h0 = h_row0(j); of = h0 + g; oo = h0 + q; }
                    else { const uint32_t w = cell(p, j); const int h = sext(w); of = h - (int)((w >> 16) & 0xffu); oo = h - (int)(w >> 24); }
                    fmax = max(fmax, of); omax = max(omax, oo);
                }
                if (!miss) {
                    if (fmax == hv) { st = SRC_F; gv = hv; done = true; }
                    else if (CVX && omax == hv) { st = SRC_O; gv = hv; done = true; }
                }
            }
            if (!done && !miss) {
                // a gap in the graph.  E: smallest k with H[i][j-k] + g + (k-1) e == hv (k <= kmax_e), else Q
                // likewise with q, c; 64 columns per round trip, straight from the plane.
                const int hint = (int)TBU(g_meta[8 * (size_t)(i - 1) + HW]);
                int kk = 0, hnew = 0;
                for (int piece = 0; piece < (CVX ? 2 : 1) && !kk && !miss; ++piece) {
                    const int go = piece ? q : g, ge = piece ? c : e;
                    const int kcap = piece ? j : min(j, kmax_e);
                    for (int base = 0; base < kcap && !kk && !miss; base += 64) {
                        const int x = base + lane + 1;
                        const bool act = x <= kcap;
                        const int col = j - x;
                        const int s = col / W, k = col - s * W;
                        const bool inb = act && kept(hint, s);
                        int hval = 0;
                        if constexpr (CB == 2) {
                            if (act && inb) {
                                unsigned raw[SD];
                                plane_load_slot<SD>(g_plane + (size_t)i * (size_t)(SD * BS), BS, s % BS, raw);
                                int h = (int)(short)(raw[0] & 0xffffu);
#pragma unroll
                                for (int t2 = 0; t2 < W; ++t2) if (t2 <= k) h += d_step(d_norm(d_half(raw, 1 + t2)));
                                hval = h;
                            }
                        } else
                        if (act && inb) hval = sext(g_plane[plane_cell<W>((size_t)i, BS, s % BS, k)]);
                        const unsigned long long meq = __ballot(act && inb && hval + go + (x - 1) * ge == hv);
                        const unsigned long long moob = BANDED ? 0ull : __ballot(act && !inb);
                        const int feq = meq ? (int)__builtin_ctzll(meq) : 64, foob = moob ? (int)__builtin_ctzll(moob) : 64;
                        if (foob < feq) { miss = true; miss_row = i; miss_delta = (j - (base + foob + 1)) - hint; }
                        else if (meq) { kk = base + feq + 1; hnew = __builtin_amdgcn_readlane(hval, feq); }
                    }
                }
                if (!miss) {
                    if (!kk) { miss = true; miss_row = i; miss_delta = 0; }   // (cannot happen: some candidate equals H)
                    else {
                        if (PAIRS)
                            for (int x = lane; x < kk; x += 64) { g_pair_row[n + x] = 0; g_pair_pos[n + x] = j - 1 - x; }
                        n += kk; j -= kk; hv = hnew;
                    }
                }
            }
        } else {
            // walking up a gap in the sequence: the first predecessor whose outgoing candidate carries gv;
            // it OPENED the gap iff its H + g (q) is that value, else the walk continues in it with gv - e (c)
            const bool isf = st == SRC_F;
            const int go = isf ? g : q, ge = isf ? e : c;
            int pp = -1, hp = 0;
            for (int x = 0; x < npe && pp < 0; ++x) {
                const int p = pred_at(x);
                int h, oc;
                if (p == 0) { h = h_row0(j); oc = h + go; }
                else { const uint32_t w = cell(p, j); h = sext(w); oc = h - (int)(isf ? ((w >> 16) & 0xffu) : (w >> 24)); }
                if (oc == gv) { pp = p; hp = h; }
            }
            if (!miss) {
                if (pp < 0) { miss = true; miss_row = i; miss_delta = 0; }   // (cannot happen)
                else {
                    if (PAIRS && lane == 0) { g_pair_row[n] = i; g_pair_pos[n] = -1; }
                    ++n;
                    i = pp;
                    if (hp + go == gv) { st = SRC_STOP; hv = hp; } else gv -= ge;
                }
            }
        }
    }
#undef TBU
    if (lane == 0) { ctl[TBM_FLAG] = miss ? 1 : 0; ctl[TBM_ROW] = miss_row; ctl[TBM_DELTA] = miss_delta; ctl[TBM_RANGE] = range ? 1 : 0; }
#ifdef SXG_ROW_PROF
    if (lane == 0 && B.row_prof) {
        B.row_prof[8] += tb_steps; B.row_prof[9] += tb_loads;
        B.row_prof[10] += __builtin_readcyclecounter() - tb_t0; B.row_prof[11] += tb_ld; B.row_prof[12] += tb_hits;
    }
#endif
    __builtin_amdgcn_s_setprio(0);
    return n;
}

}  // namespace sxg
