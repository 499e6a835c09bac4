/*
 * poa_simd.c -- CPU ORACLE, vectorised variant (test infrastructure / CPU baseline; see poa_oracle.h).
 *
 * AVX2 int16 row sweep of the recurrences S2 with the value-derived traceback of poa_vtb.c.  This is the
 * "not a strawman" CPU baseline SURVEY.md 8(d) asks for: the reference's spoa engine is an int16 SIMD row
 * sweep as well (spoa itself is absent from /root/reference, so this is OUR vectorisation of OUR
 * restatement -- results are bit-identical to poa_oracle.c, which the tests check).
 *
 * Per graph row, 16 columns per instruction:
 *   f = max_p oF_p, o = max_p oO_p, d = max_p H_p[j-1] + profile[letter of the row][j]      (vertical part)
 *   h0 = max(d, f, o) [, 0]
 *   E[j] = (g - e) + j e + max_{k<j} (h0[k] - k e),  Q likewise with (q, c)                  (in-row gaps as a
 *          prefix maximum: log-step scan inside a vector, running carry across vectors; using h0 instead of
 *          the final H is exact because re-opening from a gap-sourced cell never beats extending the gap)
 *   H = max(h0, E, Q);  oF = max(H + g, f + e);  oO = max(H + q, o + c)
 * H, oF, oO of every row are kept (6 bytes per cell) and the alignment is derived from them afterwards.
 * Not applicable (returns -1, the caller runs the scalar oracle) when the host lacks AVX2 or when a score or
 * a scan operand could leave int16.
 */
#include "poa_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <immintrin.h>

#define NEG16 (-30000)
#define PAD 16

struct poa_simd_ws {
    int16_t *cells;   /* H, oF, oO planes */
    size_t cap;
    int16_t *prof;    /* 5 x stride match/mismatch profile of the query */
    size_t cap_prof;
};

poa_simd_ws_t *poa_simd_ws_new(void) { return (poa_simd_ws_t *)calloc(1, sizeof(poa_simd_ws_t)); }
void poa_simd_ws_free(poa_simd_ws_t *w) {
    if (!w) return;
    free(w->cells); free(w->prof);
    free(w);
}
/* Size the planes once for a block expected to reach `rows` graph rows against sequences of `len` letters
 * (every regrowth is a fresh mapping whose pages fault in again: hundreds of MB per thread).             */
void poa_simd_ws_reserve(poa_simd_ws_t *w, long rows, long len);
int poa_simd_available(void) { return __builtin_cpu_supports("avx2") ? 1 : 0; }

typedef struct { int m, n, g, e, q, c, sw, convex; } sp_t;
static sp_t snormalise(const poa_params_t *p) {
    sp_t r;
    r.m = p->m; r.n = p->n; r.g = p->g; r.e = p->e; r.q = p->q; r.c = p->c;
    r.sw = ((p->mode & 1) == POA_MODE_SW); r.convex = 0;
    if (r.g >= r.e) { r.e = r.g; r.q = r.g; r.c = r.g; }
    else if (r.g <= r.q || r.e >= r.c) { r.q = r.g; r.c = r.e; }
    else r.convex = 1;
    return r;
}
static void *grow(void *p, size_t *cap, size_t bytes) {
    if (bytes <= *cap) return p;
    free(p);
    *cap = bytes + bytes / 2 + 64;   /* (a graph grows with every sequence: few, generous regrowths) */
    void *r = aligned_alloc(64, (*cap + 63) & ~(size_t)63);
    if (!r) { fprintf(stderr, "poa_simd: out of memory (%zu bytes)\n", bytes); abort(); }
    return r;
}

void poa_simd_ws_reserve(poa_simd_ws_t *w, long rows, long len) {
    const size_t LS = (((size_t)len + 1 + 15) & ~(size_t)15) + 2 * PAD;
    w->cells = (int16_t *)grow(w->cells, &w->cap, 3 * ((size_t)rows + 1) * LS * sizeof(int16_t));
}

/* inclusive prefix maximum of the 16 int16 lanes (lane k <- max of lanes 0..k) */
__attribute__((target("avx2"))) static inline __m256i prefix_max16(__m256i v, const __m256i neg) {
    /* within each 128-bit half: shifts by 1, 2, 4 elements, -inf shifted in */
    __m256i s = _mm256_alignr_epi8(v, neg, 14);  /* per half: [neg7, v0..v6] */
    v = _mm256_max_epi16(v, s);
    s = _mm256_alignr_epi8(v, neg, 12);
    v = _mm256_max_epi16(v, s);
    s = _mm256_alignr_epi8(v, neg, 8);
    v = _mm256_max_epi16(v, s);
    /* upper half additionally takes the total of the lower half */
    const __m256i lo_tot = _mm256_permute2x128_si256(v, v, 0x08);             /* [zero, low half] */
    __m256i b = _mm256_shufflehi_epi16(lo_tot, 0xff);                         /* element 7 of each half */
    b = _mm256_unpackhi_epi64(b, b);
    b = _mm256_blend_epi32(neg, b, 0xf0);                                     /* low half: -inf */
    return _mm256_max_epi16(v, b);
}

__attribute__((target("avx2")))
int poa_align_rows_simd(poa_simd_ws_t *ws, int N, const uint8_t *codes, const int32_t *off, const int32_t *pred,
                        const uint8_t *sink, const int32_t *row_node, const uint8_t *seq, int L,
                        const poa_params_t *pp, int32_t *out_node, int32_t *out_pos, int32_t *score) {
    if (!poa_simd_available()) return -1;
    const sp_t P = snormalise(pp);
    if (score) *score = 0;
    if (N <= 0 || L <= 0) return 0;
    /* int16 applicability: scores, the all-gap floor of a global alignment, and the scan operands h0 - k e */
    {
        const long hi = (long)P.m * L, ge = -(long)P.e * L, gc = -(long)P.c * L;
        long floor_ = 0;
        if (!P.sw) {
            const long a = -((long)P.g + (long)(N - 1) * P.e), b = -((long)P.q + (long)(N - 1) * P.c);
            const long a2 = -((long)P.g + (long)(L - 1) * P.e), b2 = -((long)P.q + (long)(L - 1) * P.c);
            floor_ = (a < b ? a : b) + (a2 < b2 ? a2 : b2);
        }
        if (hi + (ge > gc ? ge : gc) > 29000 || floor_ + (ge > gc ? ge : gc) > 29000) return -1;
        for (int i = 1; i <= N; ++i) if (off[i] - off[i - 1] > 65535) return -1;
    }
    const size_t LS = (((size_t)L + 1 + 15) & ~(size_t)15) + 2 * PAD;   /* row stride in elements */
    const size_t plane = ((size_t)N + 1) * LS;
    ws->cells = (int16_t *)grow(ws->cells, &ws->cap, 3 * plane * sizeof(int16_t));
    ws->prof = (int16_t *)grow(ws->prof, &ws->cap_prof, 5 * LS * sizeof(int16_t));
    int16_t *H = ws->cells + PAD, *oF = H + plane, *oO = oF + plane;
#define ROW(M, i) ((M) + (size_t)(i) * LS)
    for (int cdx = 0; cdx < 5; ++cdx) {
        int16_t *pr = ws->prof + (size_t)cdx * LS + PAD;
        for (long j = -PAD; j < (long)LS - PAD; ++j) pr[j] = NEG16;
        for (int j = 1; j <= L; ++j) pr[j] = (int16_t)((seq[j - 1] > 4 ? 4 : seq[j - 1]) == cdx ? P.m : P.n);
    }
    const int nvec = (L + 1 + 15) / 16;
    {   /* row 0 */
        int16_t *h = ROW(H, 0), *f = ROW(oF, 0), *o = ROW(oO, 0);
        for (long j = -PAD; j < (long)LS - PAD; ++j) { h[j] = NEG16; f[j] = NEG16; o[j] = NEG16; }
        for (int j = 0; j <= L; ++j) {
            int v = 0;
            if (!P.sw && j > 0) { const int a = P.g + (j - 1) * P.e, b = P.q + (j - 1) * P.c; v = a > b ? a : b; }
            h[j] = (int16_t)v; f[j] = (int16_t)(v + P.g); o[j] = (int16_t)(v + P.q);
        }
    }
    const __m256i vneg = _mm256_set1_epi16(NEG16), vzero = _mm256_setzero_si256();
    const __m256i vg = _mm256_set1_epi16((short)P.g), ve = _mm256_set1_epi16((short)P.e);
    const __m256i vq = _mm256_set1_epi16((short)P.q), vc = _mm256_set1_epi16((short)P.c);
    const __m256i ramp = _mm256_setr_epi16(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const __m256i ramp_e = _mm256_mullo_epi16(ramp, ve), ramp_c = _mm256_mullo_epi16(ramp, vc);
    static const int32_t zero_pred = 0;
    int best = 0, bi = -1, bj = -1;
    for (int i = 1; i <= N; ++i) {
        int np = off[i] - off[i - 1];
        const int32_t *pl = pred + off[i - 1];
        if (np == 0) { np = 1; pl = &zero_pred; }
        int16_t *h = ROW(H, i), *of = ROW(oF, i), *oo = ROW(oO, i);
        for (long j = -PAD; j < 0; ++j) { h[j] = NEG16; of[j] = NEG16; oo[j] = NEG16; }
        const int16_t *pr = ws->prof + (size_t)codes[i - 1] * LS + PAD;
        /* carries of the two prefix maxima: max over all earlier columns of h0[k] - k e (and - k c) */
        __m256i carry_e = vneg, carry_q = vneg;
        __m256i rowmax = vneg;
        for (int v = 0; v < nvec; ++v) {
            const int jb = v * 16;
            const int16_t *H0 = ROW(H, pl[0]), *F0 = ROW(oF, pl[0]), *O0 = ROW(oO, pl[0]);
            __m256i d = _mm256_loadu_si256((const __m256i *)(H0 + jb - 1));
            __m256i f = _mm256_loadu_si256((const __m256i *)(F0 + jb));
            __m256i o = _mm256_loadu_si256((const __m256i *)(O0 + jb));
            for (int k = 1; k < np; ++k) {
                const int16_t *Hk = ROW(H, pl[k]), *Fk = ROW(oF, pl[k]), *Ok = ROW(oO, pl[k]);
                d = _mm256_max_epi16(d, _mm256_loadu_si256((const __m256i *)(Hk + jb - 1)));
                f = _mm256_max_epi16(f, _mm256_loadu_si256((const __m256i *)(Fk + jb)));
                o = _mm256_max_epi16(o, _mm256_loadu_si256((const __m256i *)(Ok + jb)));
            }
            d = _mm256_adds_epi16(d, _mm256_loadu_si256((const __m256i *)(pr + jb)));
            __m256i h0 = _mm256_max_epi16(d, _mm256_max_epi16(f, o));
            if (P.sw) h0 = _mm256_max_epi16(h0, vzero);
            /* in-row gaps.  u[k] = h0[k] - k e (column k = jb + lane);  E[j] = (g - e) + j e + max_{k<j} u[k] */
            const __m256i jbe = _mm256_set1_epi16((short)(jb * P.e)), jbc = _mm256_set1_epi16((short)(jb * P.c));
            const __m256i col_e = _mm256_add_epi16(jbe, ramp_e), col_c = _mm256_add_epi16(jbc, ramp_c);
            const __m256i ue = _mm256_subs_epi16(h0, col_e), uq = _mm256_subs_epi16(h0, col_c);
            const __m256i ie = prefix_max16(ue, vneg), iq = prefix_max16(uq, vneg);
            /* exclusive = inclusive shifted by one lane, previous vectors' maximum into every lane */
            __m256i xe = _mm256_alignr_epi8(ie, _mm256_permute2x128_si256(ie, vneg, 0x02), 14);
            __m256i xq = _mm256_alignr_epi8(iq, _mm256_permute2x128_si256(iq, vneg, 0x02), 14);
            xe = _mm256_max_epi16(xe, carry_e); xq = _mm256_max_epi16(xq, carry_q);
            const __m256i E = _mm256_adds_epi16(_mm256_adds_epi16(xe, col_e), _mm256_sub_epi16(vg, ve));
            const __m256i Q = _mm256_adds_epi16(_mm256_adds_epi16(xq, col_c), _mm256_sub_epi16(vq, vc));
            __m256i hh = _mm256_max_epi16(h0, _mm256_max_epi16(E, Q));
            /* (columns beyond L hold junk that nothing reads: profile = -inf there keeps it small) */
            _mm256_storeu_si256((__m256i *)(h + jb), hh);
            _mm256_storeu_si256((__m256i *)(of + jb), _mm256_max_epi16(_mm256_adds_epi16(hh, vg), _mm256_adds_epi16(f, ve)));
            _mm256_storeu_si256((__m256i *)(oo + jb), _mm256_max_epi16(_mm256_adds_epi16(hh, vq), _mm256_adds_epi16(o, vc)));
            rowmax = _mm256_max_epi16(rowmax, hh);
            /* carry: lane 15 of the inclusive scans, with everything before */
            {
                __m256i t = _mm256_permute2x128_si256(ie, ie, 0x11);
                t = _mm256_shufflehi_epi16(t, 0xff); t = _mm256_unpackhi_epi64(t, t);
                carry_e = _mm256_max_epi16(carry_e, t);
                t = _mm256_permute2x128_si256(iq, iq, 0x11);
                t = _mm256_shufflehi_epi16(t, 0xff); t = _mm256_unpackhi_epi64(t, t);
                carry_q = _mm256_max_epi16(carry_q, t);
            }
        }
        if (P.sw) {
            /* junk columns (> L) can only hold values <= real ones? not guaranteed: mask by scanning the real range */
            int16_t mx[16];
            _mm256_storeu_si256((__m256i *)mx, rowmax);
            int rm = NEG16;
            for (int k = 0; k < 16; ++k) if (mx[k] > rm) rm = mx[k];
            if (rm > best) {
                for (int j = 0; j <= L; ++j) if (h[j] > best) { best = h[j]; bi = i; bj = j; }
            }
        } else if (sink[i - 1] && (bi < 0 || h[L] > best)) { best = h[L]; bi = i; bj = L; }
    }
    /* ---- derive the alignment (rules: poa_vtb.c) */
    int n = 0;
    if (bi >= 0) {
        if (score) *score = best;
        const int kmax_e = P.convex ? 1 + (P.g - P.q) / (P.c - P.e) : L + 1;
        int i = bi, j = bj, gv = 0;
        enum { ST_H, ST_F, ST_O } st = ST_H;
        for (;;) {
            if (i == 0) {
                if (j == 0 || P.sw) break;
                out_node[n] = -1; out_pos[n] = j - 1; ++n; --j;
                continue;
            }
            int np = off[i] - off[i - 1];
            const int32_t *pl = pred + off[i - 1];
            if (np == 0) { np = 1; pl = &zero_pred; }
            if (st == ST_H) {
                const int hv = ROW(H, i)[j];
                if (P.sw && hv == 0) break;
                int src = 0;
                if (j > 0) {
                    int d = -100000, dp = 0;
                    for (int k = 0; k < np; ++k) if (ROW(H, pl[k])[j - 1] > d) { d = ROW(H, pl[k])[j - 1]; dp = pl[k]; }
                    if (d + ((codes[i - 1] == seq[j - 1]) ? P.m : P.n) == hv) {
                        out_node[n] = row_node[i - 1]; out_pos[n] = j - 1; ++n;
                        i = dp; --j; src = 1;
                    }
                }
                if (!src) {
                    int f = -100000, o = -100000;
                    for (int k = 0; k < np; ++k) {
                        if (ROW(oF, pl[k])[j] > f) f = ROW(oF, pl[k])[j];
                        if (ROW(oO, pl[k])[j] > o) o = ROW(oO, pl[k])[j];
                    }
                    if (f == hv) { st = ST_F; gv = hv; src = 2; }
                    else if (o == hv) { st = ST_O; gv = hv; src = 3; }
                }
                if (!src) {
                    int k = 0;
                    const int16_t *h = ROW(H, i);
                    for (int x = 1; x <= j && x <= kmax_e; ++x) if (h[j - x] + P.g + (x - 1) * P.e == hv) { k = x; break; }
                    if (!k) for (int x = 1; x <= j; ++x) if (h[j - x] + P.q + (x - 1) * P.c == hv) { k = x; break; }
                    if (!k) { fprintf(stderr, "poa_simd: no source for cell (%d,%d)\n", i, j); abort(); }
                    for (int x = 0; x < k; ++x) { out_node[n] = -1; out_pos[n] = j - 1; ++n; --j; }
                }
            } else {
                const int16_t *M = st == ST_F ? oF : oO;
                const int go = st == ST_F ? P.g : P.q, ge = st == ST_F ? P.e : P.c;
                int p = -1;
                for (int k = 0; k < np; ++k) if (ROW(M, pl[k])[j] == gv) { p = pl[k]; break; }
                if (p < 0) { fprintf(stderr, "poa_simd: no predecessor carries the gap at (%d,%d)\n", i, j); abort(); }
                out_node[n] = row_node[i - 1]; out_pos[n] = -1; ++n;
                i = p;
                if (ROW(H, p)[j] + go == gv) st = ST_H; else gv -= ge;
            }
        }
        for (int a = 0, b = n - 1; a < b; ++a, --b) {
            int32_t x = out_node[a]; out_node[a] = out_node[b]; out_node[b] = x;
            x = out_pos[a]; out_pos[a] = out_pos[b]; out_pos[b] = x;
        }
    }
#undef ROW
    return n;
}
