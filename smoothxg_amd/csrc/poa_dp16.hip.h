// poa_dp16.hip.h -- packed-int16 DP sweep (gfx950 v_pk_add_i16 / v_pk_max_i16 / v_pk_sub_i16).
//
// Same semantics (S1-S5) and the same row-uniform structure as poa_dp.hip.h, but every VGPR
// holds TWO cells: lane t owns strip "lo" = columns [t*W, (t+1)*W) and strip "hi" = columns
// [T*W + t*W, T*W + (t+1)*W); the low / high 16 bits of a register are the two strips, so every
// add/max serves two cells.  Measured on MI355X (profiles/ubench/valu_rate.hip): a wave64
// integer VALU instruction issues every 4 cycles whether it is 32-bit or packed 16-bit, and the
// 32-bit sweep is bound by exactly that issue rate -- packing is the lever.
//
// What changes with packing:
//  * no packed compare exists, so every "which candidate won" bit is the SIGN of a packed
//    difference, shifted into bit k (lo strip) / bit 16+k (hi strip) of a per-row mask word;
//  * the traceback plane stores those mask words (8 per lane per row) instead of one byte per
//    cell; the 6-way source of H is resolved from "strictly beat the running maximum" bits in
//    priority order Q > E > O > F > D at traceback time;
//  * rows carry OUTGOING gap candidates (max(H+g, F+e), max(H+q, O+c)) instead of F and O: computed
//    once at the end of a row, consumed for free by a register successor and with an unpack by a
//    stored one; stored rows hold packed H plus the two 8-bit distances to the candidates;
//  * the end cell of a local alignment is found with a packed running maximum per row and a
//    wave-uniform search of the column only in rows that improve it.
// Applicability: every reachable score and intermediate fits +-15800 (host check); otherwise the
// 32-bit sweep runs.
#pragma once
#include <hip/hip_runtime.h>
#include "poa_dp.hip.h"

namespace sxg {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

constexpr int NEGP = -16384;  // "minus infinity" of the packed sweep
constexpr int P16_TB_WORDS = 8;
// mask words of one lane and row in the packed traceback plane
// (no STOP word: a local alignment stops where H is 0, and the traceback knows H of every cell it visits
// -- it starts from the best score and undoes one recorded step at a time)
enum : int { PM_GTF = 0, PM_GTO = 1, PM_GTE = 2, PM_GTQ = 3, PM_FX = 4, PM_OX = 5, PM_EX = 6, PM_QX = 7 };

__device__ __forceinline__ int pk_add(int a, int b) { return __builtin_bit_cast(int, (s16x2)(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ int pk_sub(int a, int b) { return __builtin_bit_cast(int, (s16x2)(__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ int pk_max(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ int pk_minu(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }
__device__ __forceinline__ int pk_mad(int a, int b, int c) { return __builtin_bit_cast(int, (s16x2)(__builtin_bit_cast(s16x2, a) * __builtin_bit_cast(s16x2, b) + __builtin_bit_cast(s16x2, c))); }
__device__ __forceinline__ int pk2(int lo, int hi) { return (lo & 0xffff) | (hi << 16); }
__device__ __forceinline__ int pk2s(int x) { return (x & 0xffff) | (x << 16); }  // both halves = x
__device__ __forceinline__ int pk_lo(int v) { return (int)(short)(v & 0xffff); }
__device__ __forceinline__ int pk_hi(int v) { return v >> 16; }
// bit k (lo) / 16+k (hi) <- sign bits of the packed difference d
#define SXG_SIGN_TO(mask, d, k) mask |= (((unsigned)(d)) >> (15 - (k))) & (0x00010001u << (k))

constexpr int LDS16_X = 64;  // ints of exchange scratch after the four [16] arrays of dp_fill

// Row words of the packed ring, per lane and column: packed H, and the distances H - oF, H - oO
// to the row's OUTGOING gap candidates oF = max(H + g, F + e), oO = max(H + q, O + c) -- what every
// successor takes as its F / O.  H >= F and g <= e < 0 bound the distances to [-e, -g] and [-c, -q]:
// one byte each, never clamped (host check: |g|, |q| <= 120).  The EXTEND bit of a candidate is
// implicit: extend won iff oF > H + g iff the distance is below -g.
template <bool CVX>
__device__ __forceinline__ u32x2 p16_pack_row(int h, int of, int oo) {
    const int df = pk_sub(h, of);
    return u32x2{(unsigned)h, (unsigned)(CVX ? (df | (pk_sub(h, oo) << 8)) : df)};
}
__device__ __forceinline__ void p16_unpack_row(u32x2 w, int& h, int& of, int& oo) {
    h = (int)w.x;
    of = pk_sub(h, (int)(w.y & 0x00ff00ffu));
    oo = pk_sub(h, (int)((w.y >> 8) & 0x00ff00ffu));
}
// bit k (lo) / 16+k (hi) <- bit 7 / 23 of v
#define SXG_BIT7_TO(mask, v, k) \
    mask |= ((((unsigned)(v)) >> ((k) <= 7 ? 7 - (k) : 0)) << ((k) > 7 ? (k) - 7 : 0)) & (0x00010001u << (k))

template <int W, bool CVX, bool SW>
__device__ __forceinline__ void dp_fill_p16(const Scoring& S, const RowsView& R, const int N,
                                            const uint8_t* __restrict__ seq, const int L, const DpBuffers& B,
                                            char* smem, DpResult& res) {
    static_assert(W >= 4 && W <= 15, "mask words hold W bits per strip plus the hand-over bit");
    const int T = (int)blockDim.x;
    const int NW = T >> 6;
    const int TW = T * W;           // columns of one half
    const int MB = dp16_meta_bytes(T), CH = MB / 32;  // descriptor staging area: bytes, rows per chunk
    constexpr unsigned ALL = ((1u << W) - 1u) * 0x00010001u;
    int* lds = (int*)smem;
    const i32x4* lmeta = (const i32x4*)(smem + LDS_CTL_BYTES);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int j0 = t * W;           // first column of my lo strip; hi strip starts at TW + j0
    // scoring values are block-uniform: keep them (and everything derived) in SGPRs
    const int g = __builtin_amdgcn_readfirstlane(S.g), e = __builtin_amdgcn_readfirstlane(S.e);
    const int q = __builtin_amdgcn_readfirstlane(S.q), c = __builtin_amdgcn_readfirstlane(S.c);
    const int G2 = pk2(g, g), E2 = pk2(e, e), Q2 = pk2(q, q), C2 = pk2(c, c);
    const int sm = __builtin_amdgcn_readfirstlane(S.m), sn = __builtin_amdgcn_readfirstlane(S.n);
    const int MN2 = pk2(sn - sm, sn - sm), M2 = pk2(sm, sm), ONE2 = 0x00010001, NEG2 = pk2(NEGP, NEGP);
    const int We = W * e, Wc = W * c;
    // distance + CB sets bit 7 of a byte iff the distance reached -g (-q): the candidate was an OPEN
    const unsigned CB = ((unsigned)(128 + g) & 0xffu) * 0x00010001u | ((unsigned)(128 + q) & 0xffu) * 0x01000100u;
    int* tot = lds;            // [4][16]: a_lo, a_hi, b_lo, b_hi inclusive totals per wave
    int* xch = lds + 64;       // [16][2]: (Hc[W-1] packed, ext bits) of every wave's last lane

    // query letters, one byte per (strip, column): register c2 holds (lo_k, hi_k, lo_k+1, hi_k+1)
    constexpr int NL = (W + 1) / 2;
    unsigned let[NL];
#pragma unroll
    for (int k2 = 0; k2 < NL; ++k2) {
        unsigned v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int kc = 2 * k2 + (b >> 1);  // strip-local column (W odd: one spare byte)
            const int j = (b & 1 ? TW : 0) + j0 + kc;
            const unsigned ch = (kc < W && j >= 1 && j <= L) ? (unsigned)seq[j - 1] : 15u;
            v |= (ch > 4u && ch != 15u ? 4u : ch) << (8 * b);
        }
        let[k2] = v;
    }
    // The letters live in LDS, not in registers: six loop-invariant VGPRs less, which is what the
    // allocator otherwise evicts to scratch around the multi-predecessor path -- and a scratch reload
    // is an in-order vmcnt wait behind the fold-step stores just issued.  Lane-major, read back with
    // two wide LDS loads per row.
    // (raw LDS byte offset: the dynamic LDS starts right behind the kernel's static LDS)
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    const unsigned llet_off = (unsigned)__builtin_amdgcn_groupstaticsize() + (unsigned)(LDS_CTL_BYTES + MB + TW * 8 + t * NL * 4);
#pragma unroll
    for (int k2 = 0; k2 < NL; ++k2) ((lds_u32*)(size_t)llet_off)[k2] = let[k2];

    // Rows in HBM (ring, row 0) are laid out [column-in-strip][lane] and the mask plane
    // [row][word][lane]: a load/store instruction then covers 64 consecutive words of a wave
    // instead of 64 different cache lines (the lane-major layout kept the texture-address unit
    // busier than the VALU).  Every access is "uniform pointer"[ut]: scalar base, one loop-invariant
    // lane offset register, no per-access address arithmetic.
    const unsigned ut = (unsigned)t;
    SXG_GLOBAL u32x2* const g_row0 = sxg_uniform(sxg_global((u32x2*)B.row0));
    SXG_GLOBAL u32x2* const g_pool = sxg_uniform(sxg_global((u32x2*)B.pool));
    SXG_GLOBAL uint32_t* const g_tb = sxg_uniform(sxg_global((uint32_t*)B.tb));
    SXG_GLOBAL uint32_t* const g_steps = sxg_uniform(sxg_global(B.steps));
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
            const int j = hf * TW + j0 + k;
            int h = 0;
            if (!SW && j > 0) { const int a = g + (j - 1) * e, b = q + (j - 1) * c; h = max(a > b ? a : b, NEGP); }
            h2[hf] = h;
        }
        Hp[k] = pk2(h2[0], h2[1]);
        Fp[k] = pk_add(Hp[k], G2);                  // nothing to extend in row 0: both candidates open
        Op[k] = CVX ? pk_add(Hp[k], Q2) : NEG2;
    }
    {
        int h2[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int j = hf * TW + j0 - 1;
            int h = 0;
            if (!SW && j > 0) { const int a = g + (j - 1) * e, b = q + (j - 1) * c; h = max(a > b ? a : b, NEGP); }
            h2[hf] = j < 0 ? NEGP : h;
        }
        Hleft = pk2(h2[0], h2[1]);
    }
    {
#pragma unroll
        for (int k = 0; k < W; ++k) (g_row0 + k * T)[ut] = p16_pack_row<CVX>(Hp[k], Fp[k], Op[k]);
    }
    int best_lo = SW ? 0 : NEGP * 2, best_hi = best_lo, bi_lo = -1, bi_hi = -1, bk_lo = 0, bk_hi = 0;
    const int kL_lo = L - j0, kL_hi = L - TW - j0;  // strip-local index of the end column L

    unsigned fxm = 0, oxm = 0;  // EXTEND bits that go with Fp / Op
    bool next_sib = false;      // decided at the end of a row for its successor
#ifdef SXG_ROW_PROF
    unsigned long long racc[8] = {0};  // per row segment; scalar registers (s_memtime deltas)
#define RP_MARK(seg) do { const unsigned long long tn_ = __builtin_readcyclecounter(); racc[seg] += tn_ - rt_; rt_ = tn_; } while (0)
#else
#define RP_MARK(seg) do { } while (0)
#endif
    for (int i = 1; i <= N; ++i) {
        if (B.prio_board) { if ((i & 127) == 1) sxg_balance_prio(B, (unsigned long long)i * (unsigned long long)L); }
        else if ((i & 63) == 1) sxg_rotate_prio(B.prio_rank);
        const int r = i - 1;
        if ((r & (CH - 1)) == 0) {
            __syncthreads();
            SXG_GLOBAL const i32x4* gm = (SXG_GLOBAL const i32x4*)(g_meta + 8 * (size_t)r);
            i32x4* lm = (i32x4*)(smem + LDS_CTL_BYTES);
            const int nrow = min(CH, N - r);
            for (int x = t; x < 2 * nrow; x += T) lm[x] = gm[x];
            __syncthreads();
        }
        const i32x4 m0 = lmeta[2 * (r & (CH - 1))], m1 = lmeta[2 * (r & (CH - 1)) + 1];
        const int pb = __builtin_amdgcn_readfirstlane(m0.x);
        const int info = __builtin_amdgcn_readfirstlane(m0.y);
        const int p0 = __builtin_amdgcn_readfirstlane(m0.z);
        const int s0 = __builtin_amdgcn_readfirstlane(m0.w);
        const int p1 = __builtin_amdgcn_readfirstlane(m1.x);
        const int s1 = __builtin_amdgcn_readfirstlane(m1.y);
        const int myslot = __builtin_amdgcn_readfirstlane(m1.z);
        const int tx = __builtin_amdgcn_readfirstlane(m1.w);
        const int np = info & 0xffff, code = (info >> 16) & 0xff, flags = (info >> 24) & 0xff;
        const unsigned CODE4 = (unsigned)code * 0x01010101u;

#ifdef SXG_ROW_PROF
        unsigned long long rt_ = __builtin_readcyclecounter();
#endif
        int Hc[W];
        // Fp/Op/fxm/oxm arrive holding the previous row's OUTGOING candidates -- or, when the previous
        // row announced this one as its sibling (an alternative allele: the same single predecessor),
        // that row's own F/O, which are this row's too.
        const bool sib = next_sib;
// the column left of my strips in a stored row: lane t-1's last column; lane 0: lo = none,
// hi = last column of the lo half (lane T-1)
#define P16_LOAD_LEFT(sp_base, hl)                                                    \
    do {                                                                              \
        if (t > 0) hl = (int)((sp_base) + (W - 1) * T - 1)[ut].x;                     \
        else hl = pk2(NEGP, pk_lo((int)(sp_base)[TW - 1].x));                         \
    } while (0)
// words of the stored row of predecessor p_ (slot sl_) and the column to their left
#define P16_FETCH(p_, sl_, wr_, hl_)                                                                        \
    do {                                                                                                    \
        SXG_GLOBAL const u32x2* base_ = sxg_uniform(((p_) == 0) ? (SXG_GLOBAL const u32x2*)g_row0 : g_pool + (size_t)(sl_) * TW); \
        _Pragma("unroll") for (int k = 0; k < W; ++k) wr_[k] = (base_ + k * T)[ut];                         \
        P16_LOAD_LEFT(base_, hl_);                                                                          \
    } while (0)

        if (np <= 1 && p0 == i - 1) {
            // register predecessor: its outgoing candidates ARE this row's F and O
#pragma unroll
            for (int k = 0; k < W; ++k) Hc[k] = k ? Hp[k - 1] : Hleft;
        } else if (sib) {
            u32x2 wr[W];
            int hl;
            P16_FETCH(p0, s0, wr, hl);
            Hc[0] = hl;
#pragma unroll
            for (int k = 1; k < W; ++k) Hc[k] = (int)wr[k - 1].x;
        } else {
            const bool reg0 = (p0 == i - 1), reg1 = (np == 2 && p1 == i - 1);
            const bool park = np >= 3;
            // my slice of the parked row, as a raw LDS offset rebuilt from an opaque t (one loop-invariant
            // address register less to spill around this path)
            typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
            int tp_ = t;
            asm volatile("" : "+v"(tp_));
            lds_u32x2* const lrow_t = (lds_u32x2*)(size_t)((unsigned)__builtin_amdgcn_groupstaticsize() +
                                                           (unsigned)(LDS_CTL_BYTES + MB) + (unsigned)(tp_ * W) * 8u);
            if (park) {
#pragma unroll
                for (int k = 0; k < W; ++k) lrow_t[k] = p16_pack_row<CVX>(Hp[k], Fp[k], Op[k]);
            }
            const int ge = reg1 ? 1 : 0;
            const int GE2 = ge ? ONE2 : 0;
            unsigned nfx = 0, nox = 0;  // OPEN (= not EXTEND) bits while the predecessors are folded
            if ((reg0 || reg1) && !park) {
#pragma unroll
                for (int k = 0; k < W; ++k) Hc[k] = k ? Hp[k - 1] : Hleft;
                nfx = ~fxm; nox = ~oxm;
            } else {
                u32x2 wr[W];
                int hl = Hleft;
                if (reg0) {
#pragma unroll
                    for (int k = 0; k < W; ++k) wr[k] = lrow_t[k];
                } else P16_FETCH(p0, s0, wr, hl);
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    int hs;
                    p16_unpack_row(wr[k], hs, Fp[k], Op[k]);
                    const unsigned tn = wr[k].y + CB;
                    SXG_BIT7_TO(nfx, tn, k);
                    if (CVX) SXG_SIGN_TO(nox, tn, k);
                    Hc[k] = hl;
                    hl = hs;
                    SXG_PIN("+v"(Hc[k]), "+v"(Fp[k]), "+v"(Op[k]), "+v"(nfx), "+v"(nox), "+v"(hl));
                }
            }
            for (int x = 1; x < np; ++x) {
                int p, sl;
                if (x == 1) { p = reg1 ? p0 : p1; sl = reg1 ? s0 : s1; }
                else {
                    p = __builtin_amdgcn_readfirstlane(g_preds[pb + x]);
                    sl = (p >= 1 && p != i - 1) ? __builtin_amdgcn_readfirstlane(g_slot[p - 1]) : -1;
                }
                u32x2 wr[W];
                int hl = Hleft;
                if (p == i - 1) {
#pragma unroll
                    for (int k = 0; k < W; ++k) wr[k] = lrow_t[k];
                } else P16_FETCH(p, sl, wr, hl);
                // take-over test "cand + ge > cur" = sign of (cur - cand - ge); the value is the max
                // either way (on a tie both are equal); masks: xor as in the 32-bit sweep
                unsigned dm = ge ? ALL : 0u, fmk = dm, omk = CVX ? dm : 0u;
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    int hs, fs, os;
                    p16_unpack_row(wr[k], hs, fs, os);
                    const unsigned bit = 0x00010001u << k;
                    const unsigned tn = wr[k].y + CB;
                    {
                        unsigned n1 = 0;
                        SXG_BIT7_TO(n1, tn, k);
                        const unsigned rf = (((unsigned)pk_sub(pk_sub(Fp[k], fs), GE2)) >> (15 - k)) & bit;
                        Fp[k] = pk_max(Fp[k], fs);
                        nfx = (nfx & ~rf) | (rf & n1);
                        fmk ^= rf;
                    }
                    if (CVX) {
                        unsigned n2 = 0;
                        SXG_SIGN_TO(n2, tn, k);
                        const unsigned ro = (((unsigned)pk_sub(pk_sub(Op[k], os), GE2)) >> (15 - k)) & bit;
                        Op[k] = pk_max(Op[k], os);
                        nox = (nox & ~ro) | (ro & n2);
                        omk ^= ro;
                    }
                    {
                        const unsigned rd = (((unsigned)pk_sub(pk_sub(Hc[k], hl), GE2)) >> (15 - k)) & bit;
                        Hc[k] = pk_max(Hc[k], hl);
                        dm ^= rd;
                    }
                    hl = hs;
                    SXG_PIN("+v"(Hc[k]), "+v"(Fp[k]), "+v"(Op[k]), "+v"(nfx), "+v"(nox), "+v"(dm), "+v"(fmk), "+v"(omk), "+v"(hl));
                }
                SXG_GLOBAL uint32_t* st = g_steps + ((size_t)(tx + x - 1) * 3) * T;
                st[ut] = dm; (st + T)[ut] = fmk; (st + 2 * T)[ut] = omk;
            }
            fxm = ~nfx & ALL;
            oxm = CVX ? ~nox & ALL : 0u;
        }
#undef P16_FETCH
#undef P16_LOAD_LEFT
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
        // ---- pass 1: H before the in-row gaps, strip-local carries
        unsigned gtf = 0, gto = 0;  // F / O strictly beat the running maximum
        int a = NEG2, b = NEG2;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            // letters (lo_k, hi_k) of column k as a packed pair; 0 where they equal the node letter
            const unsigned x4 = let[k >> 1] ^ CODE4;
            const int lp = (int)__builtin_amdgcn_perm(0u, x4, (k & 1) ? 0x0c030c02u : 0x0c010c00u);
            // 0 = match, 1 = mismatch.  Opaque to the optimiser on purpose: knowing the 0/1 range it
            // rewrites the multiply-add below into two compares, two selects and a byte merge.
            int nm;
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(nm) : "v"(lp), "v"(ONE2));
            int h = pk_mad(nm, MN2, pk_add(Hc[k], M2));    // diagonal + (match ? m : n)
            SXG_SIGN_TO(gtf, pk_sub(h, Fp[k]), k);
            h = pk_max(h, Fp[k]);
            if (CVX) { SXG_SIGN_TO(gto, pk_sub(h, Op[k]), k); h = pk_max(h, Op[k]); }
            Hc[k] = h;
            // a' = max_k (h_k + (W-1-k) e) as a running "extend, or restart here"; the opening cost
            // and the local-alignment clamp (whose best term is k = W-1) are applied once per row below
            a = pk_max(pk_add(a, E2), h);
            if (CVX) b = pk_max(pk_add(b, C2), h);
            SXG_PIN("+v"(Hc[k]), "+v"(gtf), "+v"(gto), "+v"(a), "+v"(b));
        }
        if (SW) { a = pk_max(a, 0); if (CVX) b = pk_max(b, 0); }
        a = pk_add(a, G2);
        if (CVX) b = pk_add(b, Q2);
        // ---- carries (32-bit).  Strip order: lo strips of lanes 0..T-1, then hi strips.
        // y = a - s*W*e with strip index s (lo: t, hi: T + t); E entering strip s = max_{s'<s} y_{s'} + (s-1)*W*e
        // (the lane's offsets are rebuilt from an opaque copy of t every row: hoisted out of the loop
        // they are six more loop-invariant VGPRs, which the allocator spills and reloads per row --
        // and a scratch reload is an in-order vmcnt wait behind every store still in flight)
        int tt = t;
        asm volatile("" : "+v"(tt));
        const int tWe = __mul24(tt, We), tWc = __mul24(tt, Wc);
        int ya_lo = pk_lo(a) - tWe, ya_hi = pk_hi(a) - tWe - T * We;
        int yb_lo = CVX ? pk_lo(b) - tWc : NEG, yb_hi = CVX ? pk_hi(b) - tWc - T * Wc : NEG;
        ya_lo = sxg_wave_incl_max(ya_lo); ya_hi = sxg_wave_incl_max(ya_hi);
        if (CVX) { yb_lo = sxg_wave_incl_max(yb_lo); yb_hi = sxg_wave_incl_max(yb_hi); }
        if (lane == 63) { tot[wv] = ya_lo; tot[16 + wv] = ya_hi; tot[32 + wv] = yb_lo; tot[48 + wv] = yb_hi; }
        RP_MARK(1);  // pass 1 + in-wave scan
        SXG_ROW_BARRIER();  // B1
        RP_MARK(2);  // waiting at B1
        {
            int b0 = NEG * 2, b1 = NEG * 2, b2 = NEG * 2, b3 = NEG * 2, lo_a = NEG * 2, lo_b = NEG * 2;
            for (int x = 0; x < NW; ++x) {
                const int v0 = tot[x], v1 = tot[16 + x], v2 = tot[32 + x], v3 = tot[48 + x];
                lo_a = max(lo_a, v0); lo_b = max(lo_b, v2);
                if (x < wv) { b0 = max(b0, v0); b1 = max(b1, v1); b2 = max(b2, v2); b3 = max(b3, v3); }
            }
            b1 = max(b1, lo_a); b3 = max(b3, lo_b);  // every lo strip precedes every hi strip
            ya_lo = max(ya_lo, b0); ya_hi = max(ya_hi, b1); yb_lo = max(yb_lo, b2); yb_hi = max(yb_hi, b3);
            ya_lo = sxg_wave_shr1(ya_lo, b0); ya_hi = sxg_wave_shr1(ya_hi, b1);
            yb_lo = sxg_wave_shr1(yb_lo, b2); yb_hi = sxg_wave_shr1(yb_hi, b3);
        }
        const int Ein_lo = (tt == 0) ? NEGP : max(ya_lo + tWe - We, NEGP);
        const int Ein_hi = max(ya_hi + tWe + (T - 1) * We, NEGP);
        const int Qin_lo = (tt == 0 || !CVX) ? NEGP : max(yb_lo + tWc - Wc, NEGP);
        const int Qin_hi = !CVX ? NEGP : max(yb_hi + tWc + (T - 1) * Wc, NEGP);
        int E = pk2(Ein_lo, Ein_hi), Q = pk2(Qin_lo, Qin_hi);

        // ---- pass 2: final H and the remaining decision bits
        // exm/qxm: EXTEND bit of E/Q of column k; the decision made at column k belongs to column
        // k+1, so it goes straight to bit k+1 (bit W = the hand-over to the next lane)
        unsigned gte = 0, gtq = 0, exm = 0, qxm = 0;
        int rowmax = SW ? 0 : NEG2;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            int h = Hc[k];
            SXG_SIGN_TO(gte, pk_sub(h, E), k);
            h = pk_max(h, E);
            if (CVX) { SXG_SIGN_TO(gtq, pk_sub(h, Q), k); h = pk_max(h, Q); }
            if (SW) h = pk_max(h, 0);
            Hc[k] = h;
            rowmax = pk_max(rowmax, h);
            const int c1 = pk_add(h, G2), c2 = pk_add(E, E2);
            SXG_SIGN_TO(exm, pk_sub(c1, c2), k + 1);
            E = pk_max(c1, c2);
            if (CVX) {
                const int d1 = pk_add(h, Q2), d2 = pk_add(Q, C2);
                SXG_SIGN_TO(qxm, pk_sub(d1, d2), k + 1);
                Q = pk_max(d1, d2);
            }
            SXG_PIN("+v"(Hc[k]), "+v"(gte), "+v"(gtq), "+v"(exm), "+v"(qxm), "+v"(E), "+v"(Q), "+v"(rowmax));
        }
        // hand my last column and the ext bits of the next column to the right neighbour; the lo
        // half's last lane feeds lane 0's hi strip
        const int xh = Hc[W - 1];
        const int xb = (int)(((exm >> W) & 0x00010001u) | (((qxm >> W) & 0x00010001u) << 1));  // bits 0,1 (lo strip), 16,17 (hi strip)
        exm &= ALL; qxm &= ALL;
        int lh = sxg_wave_shr1(xh, 0), lb = sxg_wave_shr1(xb, 0);
        if (lane == 63) { xch[2 * wv] = xh; xch[2 * wv + 1] = xb; }
        RP_MARK(3);  // carry combine + pass 2
        SXG_ROW_BARRIER();  // B2
        RP_MARK(4);  // waiting at B2
        if (lane == 0) {
            if (wv > 0) { lh = xch[2 * (wv - 1)]; lb = xch[2 * (wv - 1) + 1]; }
            else {
                const int th = xch[2 * (NW - 1)], tb_ = xch[2 * (NW - 1) + 1];  // lane T-1
                lh = pk2(NEGP, pk_lo(th));
                lb = (tb_ & 3) << 16;  // its lo-strip bits become my hi-strip bits; my lo strip starts the row
            }
        }
        exm |= ((unsigned)lb & 0x00010001u);
        if (CVX) qxm |= (((unsigned)lb >> 1) & 0x00010001u);

        // ---- end cell bookkeeping
        if (SW) {
            const bool il = pk_lo(rowmax) > best_lo, ih = pk_hi(rowmax) > best_hi;
            if (__any(il || ih)) {  // wave-uniform: only the waves the best diagonal runs through
                if (il) { best_lo = pk_lo(rowmax); bi_lo = i; }
                if (ih) { best_hi = pk_hi(rowmax); bi_hi = i; }
#pragma unroll
                for (int k = W - 1; k >= 0; --k) {
                    if (il && pk_lo(Hc[k]) == best_lo) bk_lo = k;
                    if (ih && pk_hi(Hc[k]) == best_hi) bk_hi = k;
                }
            }
        } else if (flags & ROW_SINK) {
#pragma unroll
            for (int k = 0; k < W; ++k) {
                if (k == kL_lo && (bi_lo < 0 || pk_lo(Hc[k]) > best_lo)) { best_lo = pk_lo(Hc[k]); bi_lo = i; bk_lo = k; }
                if (k == kL_hi && (bi_hi < 0 || pk_hi(Hc[k]) > best_hi)) { best_hi = pk_hi(Hc[k]); bi_hi = i; bk_hi = k; }
            }
        }

        RP_MARK(5);  // hand-over + end-cell bookkeeping
        // ---- stores
        {
            SXG_GLOBAL uint32_t* dst = g_tb + (size_t)i * P16_TB_WORDS * T;  // [row][word][lane]
            (dst + PM_GTF * T)[ut] = gtf; (dst + PM_GTO * T)[ut] = gto;
            (dst + PM_GTE * T)[ut] = gte; (dst + PM_GTQ * T)[ut] = gtq; (dst + PM_FX * T)[ut] = fxm;
            (dst + PM_OX * T)[ut] = oxm; (dst + PM_EX * T)[ut] = exm; (dst + PM_QX * T)[ut] = qxm;
        }
        RP_MARK(6);  // mask-plane stores
        // ---- outgoing candidates (see p16_pack_row).  A sibling successor -- single predecessor, the
        // same as mine, not me -- wants my own F/O left in place instead.  (Requesting the next row's
        // stored predecessor from here was tried: with the kernel at its 128-VGPR budget any spill
        // reload behind the request is an in-order vmcnt wait on it, and the row got slower.)
        next_sib = false;
        if (np <= 1 && i < N && (i & (CH - 1)) != 0) {
            const i32x4 n0 = lmeta[2 * (i & (CH - 1))];
            const int nnp = __builtin_amdgcn_readfirstlane(n0.y) & 0xffff, np0 = __builtin_amdgcn_readfirstlane(n0.z);
            next_sib = nnp <= 1 && np0 == p0 && np0 != i;
        }
        SXG_GLOBAL u32x2* const rdst = g_pool + (size_t)myslot * TW;
        if (next_sib) {
            if (flags & ROW_STORE) {
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    const int tf = pk_max(pk_add(Hc[k], G2), pk_add(Fp[k], E2));
                    const int to = CVX ? pk_max(pk_add(Hc[k], Q2), pk_add(Op[k], C2)) : NEG2;
                    (rdst + k * T)[ut] = p16_pack_row<CVX>(Hc[k], tf, to);
                }
            }
        } else {
            fxm = 0; oxm = 0;
#pragma unroll
            for (int k = 0; k < W; ++k) {
                const int c1 = pk_add(Hc[k], G2), c2 = pk_add(Fp[k], E2);
                Fp[k] = pk_max(c1, c2);
                SXG_SIGN_TO(fxm, pk_sub(c1, c2), k);
                if (CVX) {
                    const int d1 = pk_add(Hc[k], Q2), d2 = pk_add(Op[k], C2);
                    Op[k] = pk_max(d1, d2);
                    SXG_SIGN_TO(oxm, pk_sub(d1, d2), k);
                }
                SXG_PIN("+v"(Fp[k]), "+v"(Op[k]), "+v"(fxm), "+v"(oxm));
            }
            if (flags & ROW_STORE) {
#pragma unroll
                for (int k = 0; k < W; ++k) (rdst + k * T)[ut] = p16_pack_row<CVX>(Hc[k], Fp[k], Op[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < W; ++k) Hp[k] = Hc[k];
        Hleft = lh;
        RP_MARK(7);  // outgoing candidates + row store
    }
#ifdef SXG_ROW_PROF
    if (t == 0 && B.row_prof)
        for (int k = 0; k < 8; ++k) B.row_prof[k] += racc[k];
#endif
#undef RP_MARK

    // ---- end cell: greatest score, then smallest row, then smallest column (two candidates per lane)
    unsigned long long key = 0;
    if (bi_lo >= 0)
        key = ((unsigned long long)(unsigned)(best_lo + (1 << 27)) << 35) |
              ((unsigned long long)(0xFFFFFu - (unsigned)bi_lo) << 15) | (unsigned long long)(0x7FFFu - (unsigned)(j0 + bk_lo));
    if (bi_hi >= 0) {
        const unsigned long long k2 = ((unsigned long long)(unsigned)(best_hi + (1 << 27)) << 35) |
                                      ((unsigned long long)(0xFFFFFu - (unsigned)bi_hi) << 15) |
                                      (unsigned long long)(0x7FFFu - (unsigned)(TW + j0 + bk_hi));
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
        res.best = (int)(unsigned)(key >> 35) - (1 << 27);
        res.bi = (int)(0xFFFFFu - (unsigned)((key >> 15) & 0xFFFFFu));
        res.bj = (int)(0x7FFFu - (unsigned)(key & 0x7FFFu));
    }
}

// Traceback over the packed mask plane (S5).  The source of H is the LAST candidate in the order
// D, F, O, E, Q that strictly beat the running maximum (first-wins priority D > F > O > E > Q),
// unless the cell is a STOP.
// Windowed walk.  The replay is a chain of dependent reads, so one lane chasing HBM (the first
// version) paid a full memory round trip per step -- 18 % of the slot time on the headline
// workload.  Here the whole of wave 0 takes part: lane l fetches, in ONE round trip, everything a
// step through row (top - l) can need -- the mask words of the two lane-columns around the
// diagonal through the current cell, the row descriptor and the node id -- into an LDS window of 64
// rows; the walk itself then runs out of LDS (every lane executes it redundantly, lane 0 writes)
// and only returns to HBM when it leaves the window (an indel run wider than half a strip, a
// predecessor far up the order, the hi -> lo half crossing) or needs a fold-step plane of a
// multi-predecessor row.
constexpr int TBW_ROWS = 64;
constexpr int TBW_STRIDE = 29;  // dwords per window row (odd: conflict-free fills)
// window row: 2 x P16_TB_WORDS mask words, then
enum : int { EO_PB = 16, EO_INFO = 17, EO_Q0 = 18, EO_Q1 = 19, EO_NODE = 20, EO_STEP = 21 /* ..26 */, EO_TX = 27 };
static_assert(2 * P16_TB_WORDS == EO_PB, "window row layout");
static_assert((TBW_ROWS * TBW_STRIDE + TBW_ROWS) * 4 <= LDS_META_BYTES, "traceback window lives in the descriptor area");
static_assert((TBW_ROWS / 2 * TBW_STRIDE + TBW_ROWS / 2) * 4 <= LDS_META_BYTES / 2, "... also the half-size one");

// H is tracked along the walk (hv): it starts at the end cell's score and every recorded step is undone
// -- a diagonal step subtracts the letter score, leaving a gap state subtracts the opening cost, each
// extension the extension cost.  A local alignment ends at the first H = 0 (STOP has top priority, S5), so
// the sweep records no STOP bit.  The query letters of the window's diagonal ride along in the window.
template <bool PAIRS, int W>
// (views by value: a reference to the kernel's private copy trips an AMDGPU back-end assertion on
// the private-aperture null check for some strip widths)
__device__ __noinline__ int traceback_p16(const RowsView R, const DpBuffers B, const Scoring S_, const uint8_t* seq, const int best_,
                                          const int T_, const int slope_, int i, int j, int32_t* posnode, int32_t* pair_row,
                                          int32_t* pair_pos, char* smem) {
    // arguments arrive in vector registers: make the uniform ones scalar again
    const int T = __builtin_amdgcn_readfirstlane(T_), best = __builtin_amdgcn_readfirstlane(best_);
    // columns the alignment advances per graph row, in 1/256: a graph of N rows against L letters is
    // walked at about L/N columns per row (rows of other branches are skipped), which is where the
    // window is laid
    const int slope = __builtin_amdgcn_readfirstlane(slope_);
    i = __builtin_amdgcn_readfirstlane(i); j = __builtin_amdgcn_readfirstlane(j);
    Scoring S;
    S.m = __builtin_amdgcn_readfirstlane(S_.m); S.n = __builtin_amdgcn_readfirstlane(S_.n); S.g = __builtin_amdgcn_readfirstlane(S_.g);
    S.e = __builtin_amdgcn_readfirstlane(S_.e); S.q = __builtin_amdgcn_readfirstlane(S_.q); S.c = __builtin_amdgcn_readfirstlane(S_.c);
    S.sw = __builtin_amdgcn_readfirstlane(S_.sw); S.convex = S_.convex;
    // outputs and letters through global pointers: a FLAT store also counts on lgkmcnt, and the walk
    // waits on lgkmcnt for its LDS reads every step -- i.e. it waited for the previous step's store to
    // reach HBM (measured: ~2 500 cycles per step)
    SXG_GLOBAL int32_t* const g_posnode = sxg_global(posnode);
    SXG_GLOBAL int32_t* const g_pair_row = sxg_global(pair_row);
    SXG_GLOBAL int32_t* const g_pair_pos = sxg_global(pair_pos);
    SXG_GLOBAL const uint8_t* const g_seq = sxg_global(seq);
    const int TW = T * W;
    const int lane = threadIdx.x & 63;
    const int sw = S.sw;
    // the block's only serial phase (three waves wait for this one): a dependent-read chain that
    // rarely has an instruction ready, so top priority costs the co-residents next to nothing
    __builtin_amdgcn_s_setprio(3);
    uint32_t* win = (uint32_t*)(smem + LDS_CTL_BYTES);
    const int WR = dp16_meta_bytes(T) < LDS_META_BYTES ? TBW_ROWS / 2 : TBW_ROWS;  // window rows
    uint32_t* wlet = win + WR * TBW_STRIDE;  // [WR] query letters of columns jtop, jtop-1, ...
    // first of the two lane-columns fetched for a row whose expected (half-local) column is x
    auto col0 = [&](int x) -> int { return x < W / 2 ? 0 : min((x - W / 2) / W, T - 2); };
    int n = 0, st = SRC_STOP;
    int hv = best, gv = 0;  // H of the current cell (state H) / value of the gap state being walked
    int wtop = -1, wjj = 0, whalf = -1, wj = 0;
#ifdef SXG_ROW_PROF
    unsigned long long tb_steps = 0, tb_loads = 0, tb_t0 = __builtin_readcyclecounter(), tb_ld = 0;
#endif
    for (;;) {
#ifdef SXG_ROW_PROF
        ++tb_steps;
#endif
        if (i == 0) {
            if (j == 0 || sw) break;
            if (PAIRS && lane == 0) { g_pair_row[n] = 0; g_pair_pos[n] = j - 1; }
            ++n; --j;
            continue;
        }
        if (sw && st == SRC_STOP && hv == 0) break;
        const int r = i - 1;
        const int half = j >= TW ? 1 : 0, jj = j - half * TW;
        const int lt = jj / W, bit = (jj - lt * W) + 16 * half;
        int l = wtop - i;
        int c0 = col0(wjj - ((l * slope) >> 8));
        // (the letters must come from the window as well: a global fallback load inside the step makes the
        // compiler wait for vmcnt(0) at the merge, i.e. for the previous step's posnode store to reach HBM)
        if (wtop < 0 || l < 0 || l >= WR || half != whalf || (unsigned)(lt - c0) > 1u || (unsigned)(wj - j) >= (unsigned)WR) {
            wtop = i; wjj = jj; whalf = half; wj = j;
#ifdef SXG_ROW_PROF
            ++tb_loads;
            const unsigned long long tl0 = __builtin_readcyclecounter();
#endif
            const int row = i - lane;
            if (row >= 1 && lane < WR) {
                const int c = col0(jj - ((lane * slope) >> 8));
                SXG_GLOBAL const uint32_t* mw = sxg_global((const uint32_t*)B.tb) + (size_t)row * P16_TB_WORDS * T + c;  // [row][word][lane]
                uint32_t v[2 * P16_TB_WORDS];
#pragma unroll
                for (int x = 0; x < P16_TB_WORDS; ++x) { v[x] = mw[(size_t)x * T]; v[P16_TB_WORDS + x] = mw[(size_t)x * T + 1]; }
                SXG_GLOBAL const i32x4* dm = (SXG_GLOBAL const i32x4*)(sxg_global((const int32_t*)R.meta) + 8 * (size_t)(row - 1));
                const i32x4 d0 = dm[0], d1 = dm[1];
                const int node = sxg_global((const int32_t*)R.row_node)[row - 1];
                uint32_t* e = win + lane * TBW_STRIDE;
#pragma unroll
                for (int x = 0; x < 2 * P16_TB_WORDS; ++x) e[x] = v[x];
                e[EO_PB] = (uint32_t)d0.x; e[EO_INFO] = (uint32_t)d0.y; e[EO_Q0] = (uint32_t)d0.z; e[EO_Q1] = (uint32_t)d1.x;
                e[EO_NODE] = (uint32_t)node; e[EO_TX] = (uint32_t)d1.w;
                if ((d0.y & 0xffff) >= 2) {  // first fold step of a multi-predecessor row (D, F, O planes)
                    SXG_GLOBAL const uint32_t* sp = sxg_global((const uint32_t*)B.steps) + (size_t)d1.w * 3 * T + c;
#pragma unroll
                    for (int w3 = 0; w3 < 3; ++w3) { e[EO_STEP + 2 * w3] = sp[(size_t)w3 * T]; e[EO_STEP + 1 + 2 * w3] = sp[(size_t)w3 * T + 1]; }
                }
            }
            if (lane < WR) wlet[lane] = (j - lane >= 1) ? (uint32_t)g_seq[j - lane - 1] : 255u;  // (letters: one per column)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            l = 0; c0 = col0(jj);
#ifdef SXG_ROW_PROF
            tb_ld += __builtin_readcyclecounter() - tl0;
#endif
        }
        // Everything the walk reads is the same for all lanes; saying so (readfirstlane) keeps its state
        // in scalar registers and its control flow on the scalar unit instead of 64-wide selects and
        // exec-mask juggling (measured before: ~2 700 cycles per step).
#define TBU(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
        const uint32_t* e = win + l * TBW_STRIDE;
        const uint32_t* m = e + (lt - c0) * P16_TB_WORDS;
        const int pb = (int)TBU(e[EO_PB]), info = (int)TBU(e[EO_INFO]), q0 = (int)TBU(e[EO_Q0]), q1 = (int)TBU(e[EO_Q1]), node = (int)TBU(e[EO_NODE]);
        const int np = info & 0xffff;
        auto pred_of = [&](int which) -> int {
            if (np == 0) return 0;
            if (np == 1) return q0;
            const int tx = (int)TBU(e[EO_TX]);
            int ord = 0;
            for (int x = np - 1; x >= 2 && !ord; --x)
                if ((TBU(sxg_global((const uint32_t*)B.steps)[((size_t)(tx + x - 1) * 3 + which) * T + lt]) >> bit) & 1u) ord = x;
            if (!ord) ord = (int)((TBU(e[EO_STEP + 2 * which + (lt - c0)]) >> bit) & 1u);
            return ord == 0 ? q0 : (ord == 1 ? q1 : (int)TBU(sxg_global((const int32_t*)R.preds)[pb + ord]));
        };
        if (st == SRC_STOP) {
            int src;
            const uint32_t mq = TBU(m[PM_GTQ]), me = TBU(m[PM_GTE]), mo = TBU(m[PM_GTO]), mf = TBU(m[PM_GTF]);
            if ((mq >> bit) & 1u) src = SRC_Q;
            else if ((me >> bit) & 1u) src = SRC_E;
            else if ((mo >> bit) & 1u) src = SRC_O;
            else if ((mf >> bit) & 1u) src = SRC_F;
            else src = SRC_D;
            if (src == SRC_D) {
                if (lane == 0) {
                    if (PAIRS) { g_pair_row[n] = i; g_pair_pos[n] = j - 1; }
                    if (posnode) g_posnode[j - 1] = node;
                }
                ++n;
                const int letter = (int)TBU(wlet[wj - j]);
                hv -= (letter == ((info >> 16) & 0xff)) ? S.m : S.n;
                i = pred_of(0);
                --j;
            } else { st = src; gv = hv; }
        } else if (st == SRC_F || st == SRC_O) {
            const bool isf = st == SRC_F;
            const unsigned ext = (TBU(m[isf ? PM_FX : PM_OX]) >> bit) & 1u;
            if (PAIRS && lane == 0) { g_pair_row[n] = i; g_pair_pos[n] = -1; }
            ++n;
            i = pred_of(isf ? 1 : 2);
            if (ext) gv -= isf ? S.e : S.c;
            else { hv = gv - (isf ? S.g : S.q); st = SRC_STOP; }
        } else {
            const bool ise = st == SRC_E;
            const unsigned ext = (TBU(m[ise ? PM_EX : PM_QX]) >> bit) & 1u;
            if (PAIRS && lane == 0) { g_pair_row[n] = 0; g_pair_pos[n] = j - 1; }
            ++n; --j;
            if (ext) gv -= ise ? S.e : S.c;
            else { hv = gv - (ise ? S.g : S.q); st = SRC_STOP; }
        }
    }
#undef TBU
#ifdef SXG_ROW_PROF
    if (lane == 0 && B.row_prof) {
        B.row_prof[8] += tb_steps; B.row_prof[9] += tb_loads;
        B.row_prof[10] += __builtin_readcyclecounter() - tb_t0; B.row_prof[11] += tb_ld;
    }
#endif
    return n;
}

}  // namespace sxg
