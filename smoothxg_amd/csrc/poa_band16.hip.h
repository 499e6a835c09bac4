// poa_band16.hip.h -- BANDED packed-int16 sweep: the reference's abPOA path (A11, src/smooth.cpp:133-627,
// band wb=311 / wf=0.03 at :266-271) re-designed for one wavefront.
//
// Semantics: decrees B1-B3 of oracle/poa_oracle.c.  The band of a row is a whole number of W-column strips (W = 6, 8
// or 11: the narrowest with which the band of the block's longest sequence spans at most 128 strips)
// centred on the backbone coordinate of its node -- known BEFORE the row is computed -- so the sweep does not
// mask lanes of the 5632-column geometry of poa_dp16.hip.h: ONE wave owns a window of 128 strips (128 W columns,
// lane t: strips origin+t and origin+64+t in the low / high halves of its registers) that follows the band
// down the graph.  Consequences:
//   * no barriers and no LDS exchange: the in-row gap scan is a DPP scan of one wave, the hand-over of the
//     last column a wave_shr;
//   * a band of ~925 columns (5 kbp) costs a quarter of the full row, and sixteen one-wave workgroups share a
//     CU instead of four four-wave ones;
//   * there is no row ring: every row writes its band to the plane (slot = strip mod BS, the layout the traceback of
//     poa_dp16.hip.h reads) and predecessor rows that are not in registers are fetched from there by ABSOLUTE strip, so
//     a predecessor may have had any window origin;
//   * CB = 4: one dword per cell (H | H - oF | H - oO).  CB = 2 (round 6, local alignment): the 2-byte delta codes of the
//     packed sweep -- a strip is W + 1 halfwords: the H of its OWN first column, then the W codes, the first with step 0
//     (the packed sweep's "strip 0" convention, here for every strip: a band that starts at a strip boundary can make the
//     step INTO a strip arbitrarily large -- the row to the left may not exist in a predecessor -- while inside a strip
//     the bounds of DESIGN.md 3.1 hold, because bands are whole strips).  The sweep sits at the HBM roof (0.50-0.58 of
//     the 8 TB/s peak = 80-90 % of what a copy kernel reaches on the box): 5 instead of 8 dwords per strip of 8 columns
//     is the lever.  A reader gets the column LEFT of a strip from its left neighbour's decoded last column (lane shift);
//   * the window moves rarely (its 128 strips hold a band of ~86 with slack on both sides): when the band
//     would leave it, the origin is re-centred, the letters of the new strips are loaded and the previous row,
//     if needed, is fetched from the plane like any stored row;
//   * lanes outside the band compute values nobody reads; they are forced to -inf only where the band is about
//     to change (their strips may enter the next row's band) and kept out of the gap scan, the hand-over, the
//     end-cell keys and the stores (which go to a slot beyond the row's descriptor).
// SW = false: GLOBAL alignment (smooth_abpoa sets its band for both modes, src/smooth.cpp:259-271), adaptive band only.
// Nothing is clamped at 0 then; a cell none of whose sources exists holds "minus infinity give or take a few penalties"
// (NEGP +- ...), which no real score -- all within +-15 800, checked per alignment -- ever equals or falls below, and
// such cells only occur in the strip by which a band moved LEFT (to the right of it the row's own gaps reach every cell
// from a real one), so the drift stays a few steps.  The virtual row is not banded: a row whose predecessor it is takes
// the gap costs of its columns.  End cell: column L of the sink rows (its strip is in their band: remain() = 0).
//
// ADAPTIVE band (params.banded = 2, decree B4 -- abPOA's rule): the band of a row follows the columns of the greatest H of
// its predecessor rows and the node's distance to the end of the graph.  Everything it depends on belongs to EARLIER rows,
// so it is still known before the row starts and the one-wave window applies unchanged: a finished row leaves its band
// (strips) and its leftmost / rightmost best column in words 6 and 7 of its descriptor (16 bits each), the previous row's
// stay in scalar registers, a stored predecessor's come back with one 8-byte load issued before its cells are fetched.
// Per row this adds a wave-wide maximum, two ballots and a scalar search of one lane's columns.
#pragma once
#include <hip/hip_runtime.h>
#include "poa_dp16.hip.h"

namespace sxg {

constexpr int BAND_WB = 311;        // abpt->wb, src/smooth.cpp:269
constexpr double BAND_WF = 0.03;    // abpt->wf, src/smooth.cpp:271
constexpr int BAND_W_MAX = 11;      // widest strip (decree B2; poa_band_strip_width of the oracle)
constexpr int BAND_WIN = 128;       // strips of the window (two per lane)
constexpr int BAND_WMAX = 693;      // decree B1: cap of the half-width -- a band never exceeds the window (2 * 693 / 11 + 2 = 128 strips)
// decree B2: strip width of a block whose longest sequence has maxlen letters
__host__ __device__ inline int band_strip_width(int maxlen) {
    const int w0 = BAND_WB + (int)(BAND_WF * maxlen), w = w0 < BAND_WMAX ? w0 : BAND_WMAX;
    return 2 * w <= 756 ? 6 : (2 * w <= 1008 ? 8 : 11);
}
__host__ __device__ inline int band_half_width(int L) { const int w = BAND_WB + (int)(BAND_WF * L); return w < BAND_WMAX ? w : BAND_WMAX; }
// strips the plane keeps per row: the widest band of a block whose longest sequence has maxlen letters
__host__ __device__ inline int band_plane_strips(int maxlen, int W) { return plane_round4(2 * band_half_width(maxlen) / W + 2); }
__host__ __device__ constexpr int band_lds_bytes(int W) { return LDS_CTL_BYTES + LDS_META_BYTES / 2 + 64 * W * 8; }

constexpr unsigned NEGCELL = 0x0000C000u;   // plane cell of a cell that does not exist: H = NEGP, distances 0

__device__ __forceinline__ void band_strips_of(const int hint, const int w, const int W, const int last_strip, int& bl, int& bh) {
    bl = max(hint - w, 0) / W;
    bh = min((hint + w) / W, last_strip);
}

template <bool CVX, int W, bool SW = true, int CB = 4>
__device__ __noinline__ DpResult dp_fill_band16(const Scoring S, const RowsView R, const int N_, const uint8_t* seq, const int L_,
                                               const DpBuffers B, char* smem, unsigned long long* cells_out) {
    DpResult res;
    const int N = __builtin_amdgcn_readfirstlane(N_), L = __builtin_amdgcn_readfirstlane(L_);
    const int MB = LDS_META_BYTES / 2, CH = MB / 32;
    const i32x4* lmeta = (const i32x4*)(smem + LDS_CTL_BYTES);
    const int lane = threadIdx.x;
    const int g = __builtin_amdgcn_readfirstlane(S.g), e = __builtin_amdgcn_readfirstlane(S.e);
    const int q = __builtin_amdgcn_readfirstlane(S.q), c = __builtin_amdgcn_readfirstlane(S.c);
    const int G2 = pk2(g, g), E2 = pk2(e, e), Q2 = pk2(q, q), C2 = pk2(c, c);
    const int sm = __builtin_amdgcn_readfirstlane(S.m), sn = __builtin_amdgcn_readfirstlane(S.n);
    const int NEG2 = pk2(NEGP, NEGP);
    const unsigned SC_N4 = (unsigned)(sn & 0xff) * 0x01010101u, SC_MX = (unsigned)((sm ^ sn) & 0xff);   // score table, see poa_dp16.hip.h
    const int We = W * e, Wc = W * c;
    const int BS = __builtin_amdgcn_readfirstlane(B.band_strips), bw = __builtin_amdgcn_readfirstlane(B.band_w);
    const int last_strip = L / W;   // the strip that holds column L
    static_assert(CB == 4 || SW, "2-byte band cells: local alignment only (every cell of a band is a real score)");
    constexpr int SD = p16_slot_dwords(W, CB);   // dwords of one strip in a plane row
    // (CB = 2) the delta code's constants, as in dp_fill_p16
    const P16Delta DF = p16_delta_of(S);
    const int dbH_ = __builtin_amdgcn_readfirstlane(DF.bH), dbF_ = __builtin_amdgcn_readfirstlane(DF.bF);
    const int KF2 = pk2(1 << dbH_, 1 << dbH_), KO2 = pk2(1 << (dbH_ + dbF_), 1 << (dbH_ + dbF_));
    const int D_SH1 = pk2(dbH_, dbH_), D_SH2 = pk2(dbH_ + dbF_, dbH_ + dbF_);
    const int D_MH = pk2((1 << dbH_) - 1, (1 << dbH_) - 1), D_MF = pk2((1 << dbF_) - 1, (1 << dbF_) - 1);
    const int d_cst_ = g + ((-e) << dbH_) + ((CVX ? -c : 0) << (dbH_ + dbF_));
    const int D_CST = pk2(d_cst_, d_cst_);
    const int D_EA = pk2(-e, -e), D_CA = pk2(-c, -c);
    SXG_GLOBAL uint32_t* const g_tb = sxg_uniform(sxg_global((uint32_t*)B.tb));
    SXG_GLOBAL const int32_t* const g_meta = sxg_uniform(sxg_global((const int32_t*)R.meta));
    SXG_GLOBAL const int32_t* const g_preds = sxg_uniform(sxg_global((const int32_t*)R.preds));
    SXG_GLOBAL const int32_t* const g_hint = sxg_uniform(sxg_global((const int32_t*)R.tbx));
    SXG_GLOBAL const uint8_t* const g_seq = sxg_global(seq);
    const bool has_board = __builtin_amdgcn_readfirstlane((int)(B.prio_board != nullptr)) != 0;
    const int prio_rank = __builtin_amdgcn_readfirstlane(B.prio_rank);
    constexpr int NL = (W + 1) / 2;
    typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
    // adaptive band (B4): per-row records {band word = first | last strip << 16, best-cell word = leftmost | rightmost column << 16}
    // in words 6, 7 of the row descriptors, written and read with vector instructions only (this wave wrote them)
    const bool ada = __builtin_amdgcn_readfirstlane(B.band_mode) == 2;
    // (device-scope accesses, sc1: the descriptor lines were read into the CU's vector L1 when their chunk was staged, and a
    // later store does not refresh that copy -- a plain load of the record would still see the staged words)
    constexpr int REC_AUX = 0x10;
    const __amdgpu_buffer_rsrc_t rs_meta = p16_rsrc((const void*)g_meta, N * 32);
    int prev_bw = 0, prev_lr = 0;   // record of row i-1
    auto load_rec = [&](const int p) -> u32x2 {   // record of a stored row p >= 1 (uniform address: one request for the wave)
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_meta, 0u, (p - 1) * 32 + 24, REC_AUX);
        return u32x2{(unsigned)__builtin_amdgcn_readfirstlane((int)v.x), (unsigned)__builtin_amdgcn_readfirstlane((int)v.y)};
    };
    auto ada_band = [&](const int p_l, const int p_r, const int rem, int& bl_, int& bh_) {
        const int c_ = L - rem;
        const int b0 = max(min(p_l, c_) - bw, 0), e0 = min(max(p_r, c_) + bw, L);
        int sb = b0 / W, se = e0 / W;
        if (se - sb + 1 > BAND_WIN) {   // (the ambiguous first rows of a local alignment: keep the window's worth around c)
            const int sc = min(max(c_, 0), L) / W;
            sb = max(sb, min(sc - BAND_WIN / 2, se - (BAND_WIN - 1)));
            se = sb + BAND_WIN - 1;
        }
        bl_ = sb; bh_ = se;
    };

    // H of the virtual row (global mode): the cost of the gap that precedes column j
    auto row0_h = [&](const int j) -> int {
        if (j < 0 || j > L) return NEGP;
        if (j == 0) return 0;
        const int a_ = g + (j - 1) * e, b_ = q + (j - 1) * c;
        return max(a_ > b_ ? a_ : b_, NEGP);
    };
    int s0 = -1000000;          // window origin (strip); far away: the first row re-centres
    unsigned let[NL];           // letters of my two strips, one byte per (strip, column): (lo_k+1, lo_k, hi_k+1, hi_k)
    unsigned so_lo = 0, so_hi = 0;   // slots of my strips in a plane row
    int Hp[W], Fp[W], Op[W], Hleft = NEG2;
#pragma unroll
    for (int k = 0; k < W; ++k) { Hp[k] = NEG2; Fp[k] = NEG2; Op[k] = NEG2; }
    int pbl = 0, pbh = -1;      // band strips of the row held in Hp/Fp/Op (row i-1); empty: nothing valid in registers
    bool regs_ok = false;       // Hp/Fp/Op hold row i-1 aligned to the current window
    bool next_sib = false;
    int best_lo = SW ? 0 : NEGP * 2, best_hi = best_lo, bi_lo = -1, bi_hi = -1, bj_lo = 0, bj_hi = 0;   // (bj: ABSOLUTE column -- the window moves)
    unsigned long long cells = 0;
    // End cell of a LOCAL alignment (round 6: the packed sweep's keys instead of "compare, vote, search the columns" in every row that
    // improves a lane -- nearly every row of the wave the diagonal runs through): per strip (greatest H so far) << 16 | 0xffff - row,
    // one unsigned max per row; folded into a scalar (value, row, strip) at every 65 536th row, whenever the window moves (a lane's
    // strips change) and at the end.  The sweep names the STRIP of the end cell (bj = -(strip + 1)), the traceback reads the column.
    unsigned key_lo = 0u, key_hi = 0u;
    unsigned long long ekey = 0ull;
    auto fold_keys = [&](const int i_end, const int s0_) {
        const unsigned ep = (unsigned)(i_end - 1) >> 16;
        auto k64 = [&](const unsigned k, const int strip) -> unsigned long long {
            if ((k >> 16) == 0u) return 0ull;   // (no positive score in this strip)
            const unsigned row = (ep << 16) | (0xffffu - (k & 0xffffu));
            return ((unsigned long long)(k >> 16) << 32) | ((unsigned long long)(0xFFFFFu - row) << 12) | (unsigned long long)(0xFFFu - (unsigned)strip);
        };
        unsigned long long kk = k64(key_lo, s0_ + lane);
        const unsigned long long k2 = k64(key_hi, s0_ + 64 + lane);
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

#ifdef SXG_ROW_PROF
    unsigned long long racc[8] = {0};
#define BP_MARK(seg) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long tn_ = __builtin_readcyclecounter(); racc[seg] += tn_ - rt_; rt_ = tn_; } while (0)
#else
#define BP_MARK(seg) do { } while (0)
#endif
    for (int i = 1; i <= N; ++i) {
#ifdef SXG_ROW_PROF
        unsigned long long rt_ = __builtin_readcyclecounter();
#endif
        if (has_board) { if ((i & 127) == 1) sxg_balance_prio(B, (unsigned long long)i * (unsigned long long)(2 * bw)); }
        else if ((i & 63) == 1) sxg_rotate_prio(prio_rank);
        const int r = i - 1;
        if ((r & (CH - 1)) == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            SXG_GLOBAL const i32x4* gm = (SXG_GLOBAL const i32x4*)(g_meta + 8 * (size_t)r);
            i32x4* lm = (i32x4*)(smem + LDS_CTL_BYTES);
            const int nrow = min(CH, N - r);
            for (int x = lane; x < 2 * nrow; x += 64) lm[x] = gm[x];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const i32x4 m0 = lmeta[2 * (r & (CH - 1))], m1 = lmeta[2 * (r & (CH - 1)) + 1];
        const int pb = __builtin_amdgcn_readfirstlane(m0.x);
        const int info = __builtin_amdgcn_readfirstlane(m0.y);
        const int p0 = __builtin_amdgcn_readfirstlane(m0.z);
        const int h0 = __builtin_amdgcn_readfirstlane(m0.w);    // hint of predecessor #0
        const int p1 = __builtin_amdgcn_readfirstlane(m1.x);
        const int h1 = __builtin_amdgcn_readfirstlane(m1.y);    // hint of predecessor #1
        const int hint = __builtin_amdgcn_readfirstlane(m1.w);
        const int np = info & 0xffff, code = (info >> 16) & 0xff, flags = (info >> 24) & 0xff;
        const unsigned SC_T0 = SC_N4 ^ (code < 4 ? SC_MX << (8 * code) : 0u), SC_T1 = SC_N4 ^ (code == 4 ? SC_MX : 0u);

        int bl, bh;
        int hp0 = h0, hp1 = h1;     // what BAND_FETCH needs of predecessors #0, #1: their hint (B2) or their band word (B4)
        int my_pl = 0, my_pr = 0;   // B4: P_l, P_r of this row (a sibling successor has the same)
        if (!ada) band_strips_of(hint, bw, W, last_strip, bl, bh);
        else {
            const bool st0 = p0 >= 1 && p0 != i - 1, st1 = np >= 2 && p1 >= 1 && p1 != i - 1;
            u32x2 r0 = u32x2{0u, 0u}, r1 = u32x2{0u, 0u};
            if (st0 && st1) {   // (both loads in flight together)
                const u32x2 v0 = __builtin_amdgcn_raw_buffer_load_b64(rs_meta, 0u, (p0 - 1) * 32 + 24, REC_AUX);
                const u32x2 v1 = __builtin_amdgcn_raw_buffer_load_b64(rs_meta, 0u, (p1 - 1) * 32 + 24, REC_AUX);
                r0 = u32x2{(unsigned)__builtin_amdgcn_readfirstlane((int)v0.x), (unsigned)__builtin_amdgcn_readfirstlane((int)v0.y)};
                r1 = u32x2{(unsigned)__builtin_amdgcn_readfirstlane((int)v1.x), (unsigned)__builtin_amdgcn_readfirstlane((int)v1.y)};
            } else if (st0) r0 = load_rec(p0);
            else if (st1) r1 = load_rec(p1);
            if (p0 == i - 1 && p0 >= 1) r0 = u32x2{(unsigned)prev_bw, (unsigned)prev_lr};
            if (np >= 2 && p1 == i - 1) r1 = u32x2{(unsigned)prev_bw, (unsigned)prev_lr};
            int p_l = (int)(r0.y & 0xffffu) + 1, p_r = (int)(r0.y >> 16) + 1;   // (np = 0: the virtual row, ml = mr = 0)
            if (np >= 2) { p_l = min(p_l, (int)(r1.y & 0xffffu) + 1); p_r = max(p_r, (int)(r1.y >> 16) + 1); }
            for (int x = 2; x < np; ++x) {
                const int p = __builtin_amdgcn_readfirstlane(g_preds[pb + x]);
                const u32x2 rx = p == i - 1 ? u32x2{(unsigned)prev_bw, (unsigned)prev_lr} : (p >= 1 ? load_rec(p) : u32x2{0u, 0u});
                p_l = min(p_l, (int)(rx.y & 0xffffu) + 1); p_r = max(p_r, (int)(rx.y >> 16) + 1);
            }
            ada_band(p_l, p_r, hint, bl, bh);
            hp0 = (int)r0.x; hp1 = (int)r1.x;
            my_pl = p_l; my_pr = p_r;
        }
        BP_MARK(0);   // descriptor, (B4) predecessor records, band
        if (bh >= bl) cells += (unsigned long long)(min(L, bh * W + W - 1) - bl * W + 1);
        if (bh >= bl && (bl < s0 || bh >= s0 + BAND_WIN)) {
            // re-centre the window on this band; registers of the previous row no longer line up with it
            if (SW && s0 > -1000000) fold_keys(i, s0);   // (the keys belong to the strips the lanes had so far)
            s0 = max(0, bl - (BAND_WIN - (bh - bl + 1)) / 2);
            regs_ok = false;
#pragma unroll
            for (int k2 = 0; k2 < NL; ++k2) {
                unsigned v = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {   // bytes (lo_k+1, lo_k, hi_k+1, hi_k): selectors into the row's score table
                    const int kc = 2 * k2 + ((b & 1) ? 0 : 1);
                    const int j = (s0 + lane + ((b >> 1) ? 64 : 0)) * W + kc;
                    const bool valid = kc < W && j >= 1 && j <= L;
                    const unsigned ch = valid ? (unsigned)g_seq[j - 1] : 5u;
                    v |= (valid && ch > 4u ? 4u : ch) << (8 * b);
                }
                let[k2] = v;
            }
            so_lo = (unsigned)((s0 + lane) % BS);
            so_hi = (unsigned)((s0 + 64 + lane) % BS);
        }
        const int st_lo = s0 + lane, st_hi = s0 + 64 + lane;   // my strips
        const bool in_lo = st_lo >= bl && st_lo <= bh, in_hi = st_hi >= bl && st_hi <= bh;
        const int m2 = (in_lo ? 0x0000ffff : 0) | (in_hi ? (int)0xffff0000 : 0);

// cells (packed words as in the ring of poa_dp16.hip.h) of predecessor row p_ (band hint hp_) for my strips, and the
// column left of them; cells outside that row's band do not exist (NEGCELL); row 0 is not stored: local mode, H = 0
#define BAND_FETCH(p_, hp_, wr_, hl_)                                                                       \
    do {                                                                                                    \
        if ((p_) == 0) {                                                                                    \
            if (SW) {                                                                                       \
                const int z_ = pk2(st_lo * W <= L ? 0 : NEGP, st_hi * W <= L ? 0 : NEGP);                    \
                _Pragma("unroll") for (int k = 0; k < W; ++k) wr_[k] = p16_pack_row<CVX>(z_, pk_add(z_, G2), CVX ? pk_add(z_, Q2) : NEG2); \
                hl_ = pk2(st_lo == 0 ? NEGP : 0, 0);                                                        \
            } else {                                                                                        \
                _Pragma("unroll") for (int k = 0; k < W; ++k) {                                             \
                    const int z_ = pk2(row0_h(st_lo * W + k), row0_h(st_hi * W + k));                       \
                    wr_[k] = p16_pack_row<CVX>(z_, pk_add(z_, G2), CVX ? pk_add(z_, Q2) : NEG2);            \
                }                                                                                           \
                hl_ = pk2(row0_h(st_lo * W - 1), row0_h(st_hi * W - 1));                                    \
            }                                                                                               \
        } else {                                                                                            \
            int fl_, fh_;                                                                                   \
            if (ada) { fl_ = (hp_) & 0xffff; fh_ = (int)((unsigned)(hp_) >> 16); }                          \
            else band_strips_of(hp_, bw, W, last_strip, fl_, fh_);                                          \
            const __amdgpu_buffer_rsrc_t rs_ = p16_rsrc((const void*)(g_tb + (size_t)(p_) * (size_t)(W * BS)), W * BS * 4); \
            const bool a_ = st_lo >= fl_ && st_lo <= fh_, b_ = st_hi >= fl_ && st_hi <= fh_;                 \
            const bool la_ = st_lo - 1 >= fl_ && st_lo - 1 <= fh_, lb_ = st_hi - 1 >= fl_ && st_hi - 1 <= fh_; \
            unsigned cl_[W], chh_[W];                                                                       \
            plane_load_strip<W>(rs_, so_lo, BS, cl_);                                                       \
            plane_load_strip<W>(rs_, so_hi, BS, chh_);                                                      \
            const unsigned ll_ = __builtin_amdgcn_raw_buffer_load_b32(rs_, plane_cell_byte<W>(BS, (unsigned)((st_lo + BS - 1) % BS), W - 1), 0, 0); \
            const unsigned lh_ = __builtin_amdgcn_raw_buffer_load_b32(rs_, plane_cell_byte<W>(BS, (unsigned)((st_hi + BS - 1) % BS), W - 1), 0, 0); \
            _Pragma("unroll") for (int k = 0; k < W; ++k) {                                                 \
                const unsigned x_ = a_ ? cl_[k] : NEGCELL, y_ = b_ ? chh_[k] : NEGCELL;                     \
                wr_[k] = u32x2{__builtin_amdgcn_perm(y_, x_, 0x05040100u), __builtin_amdgcn_perm(y_, x_, 0x07060302u)}; \
            }                                                                                               \
            hl_ = (int)__builtin_amdgcn_perm(lb_ ? lh_ : NEGCELL, la_ ? ll_ : NEGCELL, 0x05040100u);        \
        }                                                                                                   \
    } while (0)
// (CB = 2) the same from 2-byte strips: HS_/FS_/OS_[k] = H and the outgoing candidates of predecessor row p_ in my strips (cells
// that do not exist: -inf), hl_ = the column left of my strips -- my left neighbour's last column (a strip hands its own
// last H to the right; lane 0's lo strip begins the window: the strip left of it is read on its own when it matters)
#define BAND_FETCH2(p_, hp_, HS_, FS_, OS_, hl_)                                                            \
    do {                                                                                                    \
        if ((p_) == 0) {   /* (local mode: row 0 is H = 0 wherever the sequence has a column) */             \
            const int z_ = pk2(st_lo * W <= L ? 0 : NEGP, st_hi * W <= L ? 0 : NEGP);                        \
            _Pragma("unroll") for (int k = 0; k < W; ++k) { HS_[k] = z_; FS_[k] = pk_add(z_, G2); OS_[k] = CVX ? pk_add(z_, Q2) : NEG2; } \
            hl_ = pk2(st_lo == 0 ? NEGP : 0, 0);                                                            \
        } else {                                                                                            \
            int fl_, fh_;                                                                                   \
            if (ada) { fl_ = (hp_) & 0xffff; fh_ = (int)((unsigned)(hp_) >> 16); }                          \
            else band_strips_of(hp_, bw, W, last_strip, fl_, fh_);                                          \
            const __amdgpu_buffer_rsrc_t rs_ = p16_rsrc((const void*)(g_tb + (size_t)(p_) * (size_t)(SD * BS)), SD * BS * 4); \
            const bool a_ = st_lo >= fl_ && st_lo <= fh_, b_ = st_hi >= fl_ && st_hi <= fh_;                 \
            const int mx_ = (a_ ? 0x0000ffff : 0) | (b_ ? (int)0xffff0000 : 0);                             \
            unsigned dl_[SD], dh_[SD];                                                                      \
            plane_load_strip<SD>(rs_, so_lo, BS, dl_);                                                      \
            plane_load_strip<SD>(rs_, so_hi, BS, dh_);                                                      \
            /* the strip left of the window, for lane 0's lo strip: wave-uniform, rare (the window has slack on both sides) */ \
            int xl_ = NEGP;                                                                                 \
            if (s0 >= 1 && s0 - 1 >= fl_ && s0 - 1 <= fh_ && bl == s0) {                                    \
                const unsigned dw_ = lane < SD ? __builtin_amdgcn_raw_buffer_load_b32(rs_, (unsigned)plane_cell_in_row(SD, BS, (s0 - 1) % BS, min(lane, SD - 1)) * 4u, 0, 0) : 0u; \
                int h_ = (int)(short)((unsigned)__builtin_amdgcn_readlane((int)dw_, 0) & 0xffffu);         \
                for (int t2 = 1; t2 < W; ++t2) {                                                            \
                    const unsigned d_ = (unsigned)__builtin_amdgcn_readlane((int)dw_, (t2 + 1) >> 1);        \
                    const unsigned c_ = ((((t2 + 1) & 1) ? d_ >> 16 : d_ & 0xffffu) - (unsigned)d_cst_) & 0xffffu; \
                    h_ += (int)(c_ & (unsigned)((1 << dbH_) - 1)) + g;                                      \
                }                                                                                           \
                xl_ = h_;                                                                                   \
            }                                                                                               \
            int run_ = (int)__builtin_amdgcn_perm(dh_[0], dl_[0], 0x05040100u);   /* H of my strips' first columns */ \
            _Pragma("unroll") for (int k = 0; k < W; ++k) {                                                 \
                const int cn_ = pk_sub((int)__builtin_amdgcn_perm(dh_[(k + 1) >> 1], dl_[(k + 1) >> 1], ((k + 1) & 1) ? 0x07060302u : 0x05040100u), D_CST); \
                if (k) run_ = pk_add(pk_add(run_, cn_ & D_MH), G2);                                          \
                const int of_ = pk_sub(pk_sub(run_, pk_lshr(cn_, D_SH1) & D_MF), D_EA);                      \
                const int oo_ = CVX ? pk_sub(pk_sub(run_, pk_lshr(cn_, D_SH2)), D_CA) : NEG2;                \
                HS_[k] = (run_ & mx_) | (NEG2 & ~mx_);                                                      \
                FS_[k] = (of_ & mx_) | (NEG2 & ~mx_);                                                       \
                OS_[k] = CVX ? ((oo_ & mx_) | (NEG2 & ~mx_)) : NEG2;                                        \
            }                                                                                               \
            hl_ = sxg_wave_shr1(HS_[W - 1], 0);                                                             \
            { const int l63_ = __builtin_amdgcn_readlane(HS_[W - 1], 63); if (lane == 0) hl_ = pk2(xl_, pk_lo(l63_)); } \
        }                                                                                                   \
    } while (0)
#define BAND_LROW(ptr_)                                                                                     \
    int tp_ = lane;                                                                                         \
    asm volatile("" : "+v"(tp_));                                                                           \
    lds_u32x2* const ptr_ = (lds_u32x2*)(size_t)((unsigned)__builtin_amdgcn_groupstaticsize() +            \
                                                 (unsigned)(LDS_CTL_BYTES + MB) + (unsigned)(tp_ * W) * 8u)

        int Hc[W];
        // (no sibling rule since round 6 -- see dp_fill_p16: a row whose predecessor is the previous row's predecessor fetches and
        //  unpacks that row like any other; next_sib only feeds the band look-ahead below)
        if (np <= 1 && p0 == i - 1 && regs_ok) {
            // register predecessor: its outgoing candidates ARE this row's F and O
#pragma unroll
            for (int k = 0; k < W; ++k) Hc[k] = k ? Hp[k - 1] : Hleft;
        } else {
            // general case: maxima over all predecessors; the register row (if it lines up) is folded from LDS
            BAND_LROW(lrow_t);
            const bool park = regs_ok && np >= 2 && (p0 == i - 1 || p1 == i - 1 || np >= 3);
            if (park) {
#pragma unroll
                for (int k = 0; k < W; ++k) lrow_t[k] = p16_pack_row<CVX>(Hp[k], Fp[k], Op[k]);
            }
            const int hleft_reg = Hleft;
            for (int x = 0; x < max(np, 1); ++x) {
                int p, hp;
                if (x == 0) { p = p0; hp = hp0; }
                else if (x == 1) { p = p1; hp = hp1; }
                else {
                    p = __builtin_amdgcn_readfirstlane(g_preds[pb + x]);
                    if (ada) hp = p == i - 1 ? prev_bw : (p >= 1 ? (int)load_rec(p).x : 0);
                    else hp = p >= 1 ? __builtin_amdgcn_readfirstlane(g_hint[p - 1]) : 0;
                }
                int hl;
                if (CB == 2 && !(p == i - 1 && park)) {
                    if constexpr (CB == 2) {
                        int hs_[W], fs_[W], os_[W];
                        BAND_FETCH2(p, hp, hs_, fs_, os_, hl);
#pragma unroll
                        for (int k = 0; k < W; ++k) {
                            if (x == 0) { Fp[k] = fs_[k]; Op[k] = os_[k]; Hc[k] = hl; }
                            else { Fp[k] = pk_max(Fp[k], fs_[k]); if (CVX) Op[k] = pk_max(Op[k], os_[k]); Hc[k] = pk_max(Hc[k], hl); }
                            hl = hs_[k];
                            SXG_PIN("+v"(Hc[k]), "+v"(Fp[k]), "+v"(Op[k]), "+v"(hl));
                        }
                    }
                } else {
                u32x2 wr[W];
                if (p == i - 1 && park) {
#pragma unroll
                    for (int k = 0; k < W; ++k) wr[k] = lrow_t[k];
                    hl = hleft_reg;
                } else {
                    if constexpr (CB != 2) BAND_FETCH(p, hp, wr, hl);
                }
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    int hs, fs, os;
                    p16_unpack_row(wr[k], hs, fs, os);
                    if (x == 0) { Fp[k] = fs; Op[k] = os; Hc[k] = hl; }
                    else { Fp[k] = pk_max(Fp[k], fs); if (CVX) Op[k] = pk_max(Op[k], os); Hc[k] = pk_max(Hc[k], hl); }
                    hl = hs;
                    SXG_PIN("+v"(Hc[k]), "+v"(Fp[k]), "+v"(Op[k]), "+v"(hl));
                }
                }
            }
        }
#undef BAND_FETCH
#undef BAND_FETCH2
#undef BAND_LROW
        if (!CVX) {
#pragma unroll
            for (int k = 0; k < W; ++k) Op[k] = NEG2;
        }

        BP_MARK(1);   // predecessor rows fetched and folded (all loads drained)
        // ---- pass 1: H before the in-row gaps, strip-local carries
        int a = NEG2, b = NEG2;
        unsigned sc4 = 0;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            if (!(k & 1)) sc4 = __builtin_amdgcn_perm(SC_T1, SC_T0, let[k >> 1]);
            const int sc = (int)__builtin_amdgcn_perm(0u, (k & 1) ? (sc4 << 8) : sc4, 0x09030801u);
            int h = pk_add(Hc[k], sc);
            h = pk_max(h, Fp[k]);
            if (CVX) h = pk_max(h, Op[k]);
            Hc[k] = h;
            a = pk_max(pk_add(a, E2), h);
            if (CVX) b = pk_max(pk_add(b, C2), h);
            SXG_PIN("+v"(Hc[k]), "+v"(a), "+v"(b));
        }
        if (SW) { a = pk_max(a, 0); if (CVX) b = pk_max(b, 0); }
        a = pk_add(a, G2);
        if (CVX) b = pk_add(b, Q2);
        // ---- carries: strips outside the band contribute nothing; lo strips precede hi strips
        const int tWe = lane * We, tWc = lane * Wc;
        int ya_lo = in_lo ? pk_lo(a) - tWe : NEG * 2, ya_hi = in_hi ? pk_hi(a) - tWe - 64 * We : NEG * 2;
        int yb_lo = CVX && in_lo ? pk_lo(b) - tWc : NEG * 2, yb_hi = CVX && in_hi ? pk_hi(b) - tWc - 64 * Wc : NEG * 2;
        ya_lo = sxg_wave_incl_max(ya_lo); ya_hi = sxg_wave_incl_max(ya_hi);
        if (CVX) { yb_lo = sxg_wave_incl_max(yb_lo); yb_hi = sxg_wave_incl_max(yb_hi); }
        {
            const int lo_a = __builtin_amdgcn_readlane(ya_lo, 63), lo_b = __builtin_amdgcn_readlane(yb_lo, 63);
            ya_hi = max(ya_hi, lo_a); yb_hi = max(yb_hi, lo_b);
            ya_lo = sxg_wave_shr1(ya_lo, NEG * 2); ya_hi = sxg_wave_shr1(ya_hi, lo_a);
            yb_lo = sxg_wave_shr1(yb_lo, NEG * 2); yb_hi = sxg_wave_shr1(yb_hi, lo_b);
        }
        const int Ein_lo = max(ya_lo + tWe - We, NEGP);
        const int Ein_hi = max(ya_hi + tWe + 63 * We, NEGP);
        const int Qin_lo = !CVX ? NEGP : max(yb_lo + tWc - Wc, NEGP);
        const int Qin_hi = !CVX ? NEGP : max(yb_hi + tWc + 63 * Wc, NEGP);
        int E = pk2(Ein_lo, Ein_hi), Q = pk2(Qin_lo, Qin_hi);

        // ---- pass 2: final H
        int rowmax = SW ? 0 : NEG2;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            int h = pk_max(Hc[k], E);
            if (CVX) h = pk_max(h, Q);
            if (SW) h = pk_max(h, 0);
            Hc[k] = h;
            rowmax = pk_max(rowmax, h);
            E = pk_max(pk_add(h, G2), pk_add(E, E2));
            if (CVX) Q = pk_max(pk_add(h, Q2), pk_add(Q, C2));
            if constexpr (W >= 11) SXG_PIN("+v"(Hc[k]), "+v"(E), "+v"(Q), "+v"(rowmax));   // (round 6: no pin below W = 11, see dp_fill_p16's pass 2)
        }
        // the column left of my strips in THIS row (the next row's diagonal): my left neighbour's last column, the
        // lo half's last lane feeds lane 0's hi strip; a strip outside the band hands over -inf
        int lh;
        {
            const int mine = (Hc[W - 1] & m2) | (NEG2 & ~m2);
            lh = sxg_wave_shr1(mine, 0);
            const int l63 = __builtin_amdgcn_readlane(mine, 63);
            if (lane == 0) lh = pk2(NEGP, pk_lo(l63));
        }

        BP_MARK(2);   // pass 1, scan, pass 2
        // ---- end cell: only cells of the band count
        rowmax = SW ? (rowmax & m2) : ((rowmax & m2) | (NEG2 & ~m2));
        if (!SW) {
            // global: column L of a row without successors (first strictly greatest); the strip of column L is in a sink's band
            if (flags & ROW_SINK) {
                const int kL = L - last_strip * W;
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    if (k == kL && in_lo && st_lo == last_strip && (bi_lo < 0 || pk_lo(Hc[k]) > best_lo)) { best_lo = pk_lo(Hc[k]); bi_lo = i; bj_lo = L; }
                    if (k == kL && in_hi && st_hi == last_strip && (bi_hi < 0 || pk_hi(Hc[k]) > best_hi)) { best_hi = pk_hi(Hc[k]); bi_hi = i; bj_hi = L; }
                }
            }
        } else {
            // (a row that merely equals a strip's best has a smaller lower half and changes nothing: first strictly greatest, decree S4)
            if ((i & 0xffff) == 0) fold_keys(i, s0);
            unsigned inv = 0xffffu - ((unsigned)i & 0xffffu);
            asm volatile("" : "+v"(inv));
            key_lo = max(key_lo, __builtin_amdgcn_perm((unsigned)rowmax, inv, 0x05040100u));
            key_hi = max(key_hi, ((unsigned)rowmax & 0xffff0000u) | inv);
        }
        // ---- B4: leftmost / rightmost column of the band that holds the row's greatest H; the row's record
        int ml_ = 0, mr_ = 0;
        if (ada) {
            int rmx = rowmax;
            if (bh == last_strip) {   // the strip of column L also has columns beyond L: not cells, and they may hold more than any cell
                const int kL = L - last_strip * W;
                int rv = SW ? 0 : NEG2;
#pragma unroll
                for (int k = 0; k < W; ++k) if (k <= kL) rv = pk_max(rv, Hc[k]);
                rv = SW ? (rv & m2) : ((rv & m2) | (NEG2 & ~m2));
                if (st_lo == last_strip) rmx = (int)(((unsigned)rmx & 0xffff0000u) | ((unsigned)rv & 0x0000ffffu));
                if (st_hi == last_strip) rmx = (int)(((unsigned)rmx & 0x0000ffffu) | ((unsigned)rv & 0xffff0000u));
            }
            const int M = __builtin_amdgcn_readlane(sxg_wave_incl_max(max(pk_lo(rmx), pk_hi(rmx))), 63);
            const unsigned long long mk_lo = __ballot(in_lo && pk_lo(rmx) == M), mk_hi = __ballot(in_hi && pk_hi(rmx) == M);
            // lo strips (origin + lane) come before hi strips (origin + 64 + lane)
            const bool l_hi = mk_lo == 0ull, r_hi = mk_hi != 0ull;
            const int l_ln = (int)__builtin_ctzll(l_hi ? mk_hi : mk_lo), r_ln = 63 - (int)__builtin_clzll(r_hi ? mk_hi : mk_lo);
            const int l_strip = s0 + l_ln + (l_hi ? 64 : 0), r_strip = s0 + r_ln + (r_hi ? 64 : 0);
            // Every lane finds the first and the last column of its two strips that holds M -- packed, six VALU instructions
            // per column (xor, min with 1, two multiply-adds, min, max), no scalar chain -- and the two lanes that matter are
            // read out afterwards.  Columns beyond L (only in the strip of column L) never count.
            const int M2 = pk2(M, M);
            const int islast2 = (st_lo == last_strip ? 0x0000ffff : 0) | (st_hi == last_strip ? (int)0xffff0000 : 0);
            const int kL = bh == last_strip ? L - last_strip * W : W;
            u16x2 first2 = u16x2{0x7fff, 0x7fff}, last2 = u16x2{0, 0};
#pragma unroll
            for (int k = 0; k < W; ++k) {
                int d = Hc[k] ^ M2;
                if (k > kL) d |= islast2;
                const int f = __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(u16x2, d), u16x2{1, 1}));   // 0: the column holds M
                const int kf2 = pk_mad(f, pk2(0x7fff - k, 0x7fff - k), pk2(k, k));              // f ? 0x7fff : k
                const int kl2 = pk_mad(f, pk2(-(k + 1), -(k + 1)), pk2(k + 1, k + 1));           // f ? 0 : k + 1
                first2 = __builtin_elementwise_min(first2, __builtin_bit_cast(u16x2, kf2));
                last2 = __builtin_elementwise_max(last2, __builtin_bit_cast(u16x2, kl2));
            }
            const unsigned fl = (unsigned)__builtin_amdgcn_readlane(__builtin_bit_cast(int, first2), l_ln);
            const unsigned lr = (unsigned)__builtin_amdgcn_readlane(__builtin_bit_cast(int, last2), r_ln);
            const int kf = (int)(l_hi ? fl >> 16 : fl & 0xffffu), kl = (int)(r_hi ? lr >> 16 : lr & 0xffffu) - 1;
            ml_ = l_strip * W + kf; mr_ = r_strip * W + kl;
            prev_bw = bl | (bh << 16); prev_lr = ml_ | (mr_ << 16);
            if (lane == 0) __builtin_amdgcn_raw_buffer_store_b64(u32x2{(unsigned)prev_bw, (unsigned)prev_lr}, rs_meta, 0u, (i - 1) * 32 + 24, REC_AUX);
        }
        BP_MARK(3);   // end cell, (B4) best-cell search and record
        // ---- outgoing candidates, band store.  A sibling successor keeps my own F/O instead.
        next_sib = false;
        int nbl = -1, nbh = -2;    // the next row's band (unknown at a descriptor-chunk edge)
        if (i < N && (i & (CH - 1)) != 0) {
            const i32x4 n0 = lmeta[2 * (i & (CH - 1))], n1 = lmeta[2 * (i & (CH - 1)) + 1];
            const int nnp = __builtin_amdgcn_readfirstlane(n0.y) & 0xffff, np0 = __builtin_amdgcn_readfirstlane(n0.z);
            next_sib = np <= 1 && nnp <= 1 && np0 == p0 && np0 != i;
            const int nhint = __builtin_amdgcn_readfirstlane(n1.w);
            if (!ada) band_strips_of(nhint, bw, W, last_strip, nbl, nbh);
            else if (nnp <= 1 && np0 == i) ada_band(ml_ + 1, mr_ + 1, nhint, nbl, nbh);   // (my register successor)
            else if (next_sib) ada_band(my_pl, my_pr, nhint, nbl, nbh);                   // (same single predecessor as mine)
        }
        const __amdgpu_buffer_rsrc_t rs_plane = p16_rsrc((const void*)(g_tb + (size_t)i * (size_t)(SD * BS)), SD * BS * 4);
// (Round 4 measured a PLANE CUT here: rows that no later row reads back kept only 16 strips either side of the strip of their
// greatest H, the walk reported a miss for a cell such a row had not kept and that sequence's sweep was repeated with whole
// bands.  c3b 1 374 -> 1 206 blocks/s with 10 % of the alignments repeated, i.e. nothing gained by writing a third of the
// cells: the stores' cost is their acknowledgement latency in front of the next row's in-order vmcnt wait, not their volume,
// and the number of store instructions per row does not change.  Dropped; the repeat also needs prep_rows again in B4 mode,
// whose records overwrite the hints.)
#define BAND_STORE(CF, CO)                                                                                  \
    do {                                                                                                    \
        if constexpr (CB == 2) {                                                                            \
            /* delta codes of my two strips (see P16Delta), both halves of a register at once; halfword 0 of a strip = H of its \
               own first column, whose code carries the step 0 */                                              \
            int code_[W], prev_ = Hc[0];                                                                    \
            _Pragma("unroll") for (int k = 0; k < W; ++k) {                                                 \
                const int h_ = Hc[k], of_ = (CF), oo_ = (CO);                                               \
                int cd_ = pk_mad(pk_sub(h_, of_), KF2, pk_sub(h_, prev_));                                   \
                if (CVX) cd_ = pk_mad(pk_sub(h_, oo_), KO2, cd_);                                           \
                code_[k] = cd_;                                                                             \
                prev_ = h_;                                                                                 \
            }                                                                                               \
            plane_store_strip<SD>(rs_plane, in_lo ? so_lo : P16_SLOT_OOB, BS, [&](const int x) -> unsigned { \
                return __builtin_amdgcn_perm((unsigned)(2 * x < W ? code_[2 * x < W ? 2 * x : 0] : 0), (unsigned)(x ? code_[x ? 2 * x - 1 : 0] : Hc[0]), 0x05040100u); }); \
            plane_store_strip<SD>(rs_plane, in_hi ? so_hi : P16_SLOT_OOB, BS, [&](const int x) -> unsigned { \
                return __builtin_amdgcn_perm((unsigned)(2 * x < W ? code_[2 * x < W ? 2 * x : 0] : 0), (unsigned)(x ? code_[x ? 2 * x - 1 : 0] : Hc[0]), 0x07060302u); }); \
        } else {                                                                                            \
        /* (round 6: a strip outside the band goes to a slot beyond the row's descriptor -- dropped by the hardware -- instead of \
            a lane-divergent branch around the stores: see P16_SLOT_OOB in poa_dp16.hip.h) */                  \
            plane_store_strip<W>(rs_plane, in_lo ? so_lo : P16_SLOT_OOB, BS, [&](const int k) -> unsigned { \
                const u32x2 w = p16_pack_row<CVX>(Hc[k], CF, CO);                                           \
                return __builtin_amdgcn_perm(w.y, w.x, 0x05040100u); });                                    \
            plane_store_strip<W>(rs_plane, in_hi ? so_hi : P16_SLOT_OOB, BS, [&](const int k) -> unsigned { \
                const u32x2 w = p16_pack_row<CVX>(Hc[k], CF, CO);                                           \
                return __builtin_amdgcn_perm(w.y, w.x, 0x07060302u); });                                    \
        }                                                                                                   \
    } while (0)
#pragma unroll
        for (int k = 0; k < W; ++k) {
            Fp[k] = pk_max(pk_add(Hc[k], G2), pk_add(Fp[k], E2));
            if (CVX) Op[k] = pk_max(pk_add(Hc[k], Q2), pk_add(Op[k], C2));
            SXG_PIN("+v"(Fp[k]), "+v"(Op[k]));
        }
        BAND_STORE(Fp[k], Op[k]);
#undef BAND_STORE
        // what the next row takes from registers must not exist outside this row's band: forced to -inf only when
        // the band is about to change (or the next band is unknown) -- otherwise those lanes are never read
        if (nbl != bl || nbh != bh) {
#pragma unroll
            for (int k = 0; k < W; ++k) {
                Hc[k] = (Hc[k] & m2) | (NEG2 & ~m2);
                Fp[k] = (Fp[k] & m2) | (NEG2 & ~m2);
                if (CVX) Op[k] = (Op[k] & m2) | (NEG2 & ~m2);
            }
        }
#pragma unroll
        for (int k = 0; k < W; ++k) Hp[k] = Hc[k];
        Hleft = lh;
        pbl = bl; pbh = bh;
        regs_ok = true;
        BP_MARK(4);   // outgoing candidates, band stores (drained)
    }
#ifdef SXG_ROW_PROF
    if (lane == 0 && B.row_prof)
        for (int k = 0; k < 8; ++k) B.row_prof[k] += racc[k];
#endif
#undef BP_MARK
    (void)pbl; (void)pbh;
    if (lane == 0 && cells_out) *cells_out = cells;

    if (SW) {
        // ---- end cell of a local alignment: the keys of the last stretch, then the scalar best; the column is found by the traceback
        fold_keys(N + 1, s0);
        const int val = (int)(unsigned)(ekey >> 32);
        if (ekey == 0ull || val <= 0) { res.best = 0; res.bi = -1; res.bj = -1; }
        else {
            res.best = val;
            res.bi = (int)(0xFFFFFu - (unsigned)((ekey >> 12) & 0xFFFFFu));
            res.bj = -1 - (int)(0xFFFu - (unsigned)(ekey & 0xFFFu));   // -(strip + 1): see traceback_p16
        }
        return res;
    }
    // ---- end cell: greatest score, then smallest row, then smallest column (two candidates per lane)
    unsigned long long key = 0;
    if (bi_lo >= 0)
        key = ((unsigned long long)(unsigned)(best_lo + (1 << 27)) << 35) |
              ((unsigned long long)(0xFFFFFu - (unsigned)bi_lo) << 15) | (unsigned long long)(0x7FFFu - (unsigned)bj_lo);
    if (bi_hi >= 0) {
        const unsigned long long k2 = ((unsigned long long)(unsigned)(best_hi + (1 << 27)) << 35) |
                                      ((unsigned long long)(0xFFFFFu - (unsigned)bi_hi) << 15) | (unsigned long long)(0x7FFFu - (unsigned)bj_hi);
        key = k2 > key ? k2 : key;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(key, d);
        key = o > key ? o : key;
    }
    if (key == 0) { res.best = 0; res.bi = -1; res.bj = -1; }
    else {
        res.best = (int)(unsigned)(key >> 35) - (1 << 27);
        res.bi = (int)(0xFFFFFu - (unsigned)((key >> 15) & 0xFFFFFu));
        res.bj = (int)(0x7FFFu - (unsigned)(key & 0x7FFFu));
    }
    return res;
}

}  // namespace sxg
