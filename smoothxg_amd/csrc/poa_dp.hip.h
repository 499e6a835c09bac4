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

// ---------------------------------------------------------------------------------------
// workgroup context for poa_graph_dev.h
struct WgCtx {
    int* lds;  // >= 2*16+2 ints of LDS scratch
    __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
    __device__ __forceinline__ int nthreads() const { return (int)blockDim.x; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ int atomic_add(int32_t* p, int v) const { return atomicAdd(p, v); }
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
    __device__ int scan_incl_max(int v) const {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        int x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(x, d);
            if (lane >= d) x = max(x, o);
        }
        if (lane == 63) lds[w] = x;
        __syncthreads();
        for (int i = 0; i < w; ++i) x = max(x, lds[i]);
        __syncthreads();
        return x;
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

// ---------------------------------------------------------------------------------------
// packed row words of the row pool
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
    using type = uint2;
    static __device__ __forceinline__ type pack(int h, int f, int o) {
        const unsigned df = (unsigned)min(h - f, 65535), dq = (unsigned)min(h - o, 65535);
        return make_uint2((unsigned)h, df | (dq << 16));
    }
    static __device__ __forceinline__ void unpack(type w, int& h, int& f, int& o) {
        h = (int)w.x;
        f = h - (int)(w.y & 0xffffu);
        o = h - (int)(w.y >> 16);
    }
    static __device__ __forceinline__ int h_of(type w) { return (int)w.x; }
};

struct DpBuffers {
    uint8_t* tb;       // [(rows_cap+1) * Lpad] traceback bytes, row-major, row 0 unused
    uint16_t* tbx16;   // [tbx16_cap * Lpad] ordinals d|f<<5|o<<10 for rows with 2..32 preds
    uint32_t* tbx32;   // [tbx32_cap * Lpad] ordinals d|f<<10|o<<20 for rows with >32 preds
    void* pool;        // [pool_slots * Lpad] packed rows
    void* row0;        // [Lpad] packed virtual source row
};

struct DpResult {
    int best, bi, bj;  // end cell (row index 1-based); bi < 0: empty alignment
};

// ---------------------------------------------------------------------------------------
template <int T, int W, bool CVX, bool H16>
__device__ void dp_fill(const Scoring& S, const RowsView& R, const int N,
                        const uint8_t* __restrict__ seq, const int L, const DpBuffers& B,
                        int* lds, DpResult& res) {
    static_assert(W % 4 == 0, "W must be a multiple of 4");
    constexpr int NW = T / 64;
    constexpr int Lpad = T * W;
    using RWt = RowWord<H16>;
    using Word = typename RWt::type;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int j0 = t * W;
    const int g = S.g, e = S.e, q = S.q, c = S.c, mm = S.m, mn = S.n;
    const int lowclamp = S.sw ? 0 : NEG * 2;
    const int We = W * e, Wc = W * c;
    int* tot_a = lds;            // [NW]
    int* tot_b = lds + NW;       // [NW]
    int* xch_h = lds + 2 * NW;   // [NW]
    int* xch_b = lds + 3 * NW;   // [NW]

    // query letters of my columns: column j pairs with seq[j-1]
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
        if (!S.sw && j > 0) { const int a = g + (j - 1) * e, b = q + (j - 1) * c; h = a > b ? a : b; }
        Hp[k] = h; Fp[k] = NEG; Op[k] = NEG;
    }
    {
        const int j = j0 - 1;
        int h = 0;
        if (!S.sw && j > 0) { const int a = g + (j - 1) * e, b = q + (j - 1) * c; h = a > b ? a : b; }
        Hleft = j < 0 ? NEG : h;
    }
    {
        Word* r0 = (Word*)B.row0 + j0;
#pragma unroll
        for (int k = 0; k < W; ++k) r0[k] = RWt::pack(Hp[k], Fp[k], Op[k]);
    }
    int best = 0, bi = -1, bj = -1;
    __syncthreads();

    for (int i = 1; i <= N; ++i) {
        const int r = i - 1;
        const int pb = __builtin_amdgcn_readfirstlane(R.pred_off[r]);
        const int pe = __builtin_amdgcn_readfirstlane(R.pred_off[r + 1]);
        const int np = pe - pb;
        const int code = __builtin_amdgcn_readfirstlane((int)R.code[r]);
        const int flags = __builtin_amdgcn_readfirstlane((int)R.flags[r]);

        int Dm[W], F[W], O[W];
        unsigned tag[W];  // d | f<<10 | o<<20 | fx<<30 | ox<<31

        const int p0 = np ? __builtin_amdgcn_readfirstlane(R.preds[pb]) : 0;
        if (np <= 1 && p0 == i - 1) {
            // ---- fast path: single predecessor held in registers
#pragma unroll
            for (int k = 0; k < W; ++k) {
                const int c1 = Hp[k] + g, c2 = Fp[k] + e;
                unsigned tg = 0;
                F[k] = c1; if (c2 > c1) { F[k] = c2; tg |= 1u << 30; }
                if (CVX) {
                    const int d1 = Hp[k] + q, d2 = Op[k] + c;
                    O[k] = d1; if (d2 > d1) { O[k] = d2; tg |= 1u << 31; }
                } else O[k] = NEG;
                Dm[k] = k ? Hp[k - 1] : Hleft;
                tag[k] = tg;
            }
        } else {
            for (int x = 0; x < (np ? np : 1); ++x) {
                const int p = np ? __builtin_amdgcn_readfirstlane(R.preds[pb + x]) : 0;
                if (p == i - 1) {
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        const int hs = Hp[k], fs = Fp[k], os = Op[k];
                        const int hl = k ? Hp[k - 1] : Hleft;
                        unsigned tg = x ? tag[k] : 0u;
                        int c1 = hs + g, c2 = fs + e;
                        if (x == 0 || c1 > F[k]) { F[k] = c1; tg = (tg & ~((1023u << 10) | (1u << 30))) | ((unsigned)x << 10); }
                        if (c2 > F[k]) { F[k] = c2; tg = (tg & ~(1023u << 10)) | ((unsigned)x << 10) | (1u << 30); }
                        if (CVX) {
                            c1 = hs + q; c2 = os + c;
                            if (x == 0 || c1 > O[k]) { O[k] = c1; tg = (tg & ~((1023u << 20) | (1u << 31))) | ((unsigned)x << 20); }
                            if (c2 > O[k]) { O[k] = c2; tg = (tg & ~(1023u << 20)) | ((unsigned)x << 20) | (1u << 31); }
                        } else O[k] = NEG;
                        if (x == 0 || hl > Dm[k]) { Dm[k] = hl; tg = (tg & ~1023u) | (unsigned)x; }
                        tag[k] = tg;
                    }
                } else {
                    const Word* src = (p == 0) ? (const Word*)B.row0
                                               : (const Word*)B.pool + (size_t)__builtin_amdgcn_readfirstlane(R.slot[p - 1]) * Lpad;
                    int hl = NEG;
                    if (j0 > 0) hl = RWt::h_of(src[j0 - 1]);
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        int hs, fs, os;
                        RWt::unpack(src[j0 + k], hs, fs, os);
                        unsigned tg = x ? tag[k] : 0u;
                        int c1 = hs + g, c2 = fs + e;
                        if (x == 0 || c1 > F[k]) { F[k] = c1; tg = (tg & ~((1023u << 10) | (1u << 30))) | ((unsigned)x << 10); }
                        if (c2 > F[k]) { F[k] = c2; tg = (tg & ~(1023u << 10)) | ((unsigned)x << 10) | (1u << 30); }
                        if (CVX) {
                            c1 = hs + q; c2 = os + c;
                            if (x == 0 || c1 > O[k]) { O[k] = c1; tg = (tg & ~((1023u << 20) | (1u << 31))) | ((unsigned)x << 20); }
                            if (c2 > O[k]) { O[k] = c2; tg = (tg & ~(1023u << 20)) | ((unsigned)x << 20) | (1u << 31); }
                        } else O[k] = NEG;
                        if (x == 0 || hl > Dm[k]) { Dm[k] = hl; tg = (tg & ~1023u) | (unsigned)x; }
                        tag[k] = tg;
                        hl = hs;
                    }
                }
            }
        }

        // ---- H before the in-row gaps, strip-local carries (pass 1)
        int Hc[W];
        unsigned srcv[W];
        int a = NEG, b = NEG;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const unsigned qb = (qcw[k >> 2] >> (8 * (k & 3))) & 255u;
            int h = Dm[k] + ((int)qb == code ? mm : mn);
            unsigned src = SRC_D;
            if (F[k] > h) { h = F[k]; src = SRC_F; }
            if (CVX && O[k] > h) { h = O[k]; src = SRC_O; }
            Hc[k] = h; srcv[k] = src;
            const int hc = max(h, lowclamp);
            a = max(a + e, hc + g);
            if (CVX) b = max(b + c, hc + q);
        }
        // carries: Ein(t) = max_{s<t} (a_s + (t-1-s)*W*e)
        int ya = a - t * We, yb = CVX ? b - t * Wc : NEG;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int oa = __shfl_up(ya, d), ob = __shfl_up(yb, d);
            if (lane >= d) { ya = max(ya, oa); yb = max(yb, ob); }
        }
        if (NW > 1) {
            if (lane == 63) { tot_a[wv] = ya; tot_b[wv] = yb; }
            __syncthreads();  // B1
            int ba = NEG * 2, bb = NEG * 2;
            for (int x = 0; x < wv; ++x) { ba = max(ba, tot_a[x]); bb = max(bb, tot_b[x]); }
            ya = max(ya, ba); yb = max(yb, bb);
            int ea = __shfl_up(ya, 1), eb = __shfl_up(yb, 1);
            if (lane == 0) { ea = ba; eb = bb; }
            ya = ea; yb = eb;
        } else {
            ya = __shfl_up(ya, 1); yb = __shfl_up(yb, 1);
        }
        int E = (t == 0) ? NEG : ya + (t - 1) * We;
        int Q = (t == 0 || !CVX) ? NEG : yb + (t - 1) * Wc;

        // ---- pass 2: final H, traceback bytes
        unsigned tbw[W / 4];
        unsigned ebit = 0, qbit = 0;  // ext flags of the CURRENT column (column 0 patched later)
#pragma unroll
        for (int k = 0; k < W; ++k) {
            int h = Hc[k];
            unsigned src = srcv[k];
            if (E > h) { h = E; src = SRC_E; }
            if (CVX && Q > h) { h = Q; src = SRC_Q; }
            if (S.sw && h <= 0) { h = 0; src = SRC_STOP; }
            Hc[k] = h;
            const unsigned byte = src | ((tag[k] >> 30) & 1u) * TB_FEXT | ((tag[k] >> 31) & 1u) * TB_OEXT |
                                  ebit * TB_EEXT | qbit * TB_QEXT;
            if ((k & 3) == 0) tbw[k >> 2] = byte; else tbw[k >> 2] |= byte << (8 * (k & 3));
            const int j = j0 + k;
            if (S.sw && j <= L && h > best) { best = h; bi = i; bj = j; }
            if (!S.sw && j == L && (flags & ROW_SINK) && (bi < 0 || h > best)) { best = h; bi = i; bj = j; }
            const int c1 = h + g, c2 = E + e;
            ebit = c2 > c1; E = ebit ? c2 : c1;
            if (CVX) { const int d1 = h + q, d2 = Q + c; qbit = d2 > d1; Q = qbit ? d2 : d1; }
        }
        // hand H of my last column and the ext flags of the next column to the right neighbour
        int xh = Hc[W - 1], xb = (int)(ebit | (qbit << 1));
        int lh = __shfl_up(xh, 1), lb = __shfl_up(xb, 1);
        if (NW > 1) {
            if (lane == 63) { xch_h[wv] = xh; xch_b[wv] = xb; }
            __syncthreads();  // B2
            if (lane == 0 && wv > 0) { lh = xch_h[wv - 1]; lb = xch_b[wv - 1]; }
        }
        if (t == 0) { lh = NEG; lb = 0; }
        tbw[0] |= ((unsigned)(lb & 1) * TB_EEXT) | ((unsigned)((lb >> 1) & 1) * TB_QEXT);

        // ---- stores
        {
            unsigned* dst = (unsigned*)(B.tb + (size_t)i * Lpad + j0);
#pragma unroll
            for (int k4 = 0; k4 < W / 4; ++k4) dst[k4] = tbw[k4];
        }
        if (np > 1) {
            const int tx = __builtin_amdgcn_readfirstlane(R.tbx[r]);
            if (tx >= 0) {
                uint16_t* dst = B.tbx16 + (size_t)tx * Lpad + j0;
#pragma unroll
                for (int k = 0; k < W; k += 2) {
                    const unsigned lo = (tag[k] & 31u) | (((tag[k] >> 10) & 31u) << 5) | (((tag[k] >> 20) & 31u) << 10);
                    const unsigned hi = (tag[k + 1] & 31u) | (((tag[k + 1] >> 10) & 31u) << 5) | (((tag[k + 1] >> 20) & 31u) << 10);
                    *(unsigned*)(dst + k) = lo | (hi << 16);
                }
            } else {
                uint32_t* dst = B.tbx32 + (size_t)(-(tx + 2)) * Lpad + j0;
#pragma unroll
                for (int k = 0; k < W; ++k) dst[k] = tag[k] & 0x3fffffffu;
            }
        }
        if (flags & ROW_STORE) {
            Word* dst = (Word*)B.pool + (size_t)__builtin_amdgcn_readfirstlane(R.slot[r]) * Lpad + j0;
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = RWt::pack(Hc[k], F[k], O[k]);
        }
#pragma unroll
        for (int k = 0; k < W; ++k) { Hp[k] = Hc[k]; Fp[k] = F[k]; Op[k] = O[k]; }
        Hleft = lh;
        if (NW == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }

    // ---- end cell: greatest score, then smallest row, then smallest column
    unsigned long long key = 0;
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
template <bool PAIRS>
__device__ int traceback(const RowsView& R, const DpBuffers& B, const int Lpad, const int sw, int i, int j,
                         int32_t* posnode, int32_t* pair_row, int32_t* pair_pos) {
    int n = 0, st = SRC_STOP;
    for (;;) {
        if (i == 0) {
            if (j == 0 || sw) break;  // H[0][j] = 0 ends a local alignment
            if (PAIRS) { pair_row[n] = 0; pair_pos[n] = j - 1; }
            ++n; --j;
            continue;
        }
        const int r = i - 1;
        const unsigned tbyte = B.tb[(size_t)i * Lpad + j];
        const int pb = R.pred_off[r], np = R.pred_off[r + 1] - pb;
        int dord = 0, ford = 0, oord = 0;
        if (np > 1) {
            const int tx = R.tbx[r];
            if (tx >= 0) {
                const unsigned v = B.tbx16[(size_t)tx * Lpad + j];
                dord = v & 31; ford = (v >> 5) & 31; oord = (v >> 10) & 31;
            } else {
                const unsigned v = B.tbx32[(size_t)(-(tx + 2)) * Lpad + j];
                dord = v & 1023; ford = (v >> 10) & 1023; oord = (v >> 20) & 1023;
            }
        }
        if (st == SRC_STOP) {
            const int src = tbyte & 7;
            if (src == SRC_STOP) break;
            if (src == SRC_D) {
                if (PAIRS) { pair_row[n] = i; pair_pos[n] = j - 1; }
                if (posnode) posnode[j - 1] = R.row_node[r];
                ++n;
                i = np ? R.preds[pb + dord] : 0;
                --j;
            } else st = src;
        } else if (st == SRC_F || st == SRC_O) {
            const int ext = st == SRC_F ? (tbyte & TB_FEXT) : (tbyte & TB_OEXT);
            const int ord = st == SRC_F ? ford : oord;
            if (PAIRS) { pair_row[n] = i; pair_pos[n] = -1; }
            ++n;
            i = np ? R.preds[pb + ord] : 0;
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
