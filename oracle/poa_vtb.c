/*
 * poa_vtb.c -- CPU ORACLE, second restatement of S2-S5 (test infrastructure, PARITY UNPINNED; see
 * poa_oracle.h).  Same recurrences as align_rows() in poa_oracle.c, but NOTHING is recorded while the
 * matrix is filled: the alignment is DERIVED afterwards from the stored values -- H and the outgoing
 * gap candidates oF = max(H+g, F+e), oO = max(H+q, O+c) of every row -- by re-applying the tie rules
 * S3 to the candidates of each visited cell.  Two uses:
 *   - an independent check of poa_oracle.c's recorded-choice traceback (tests/test_oracle.py);
 *   - the executable specification of the HIP packed sweep's traceback (poa_dp16.hip.h), which
 *     stores exactly these three values per cell and derives the walk the same way.
 *
 * Derivation rules at a cell (i,j) with hv = H[i][j] (first match wins, as S3 orders the candidates):
 *   STOP  local mode and hv == 0
 *   D     max_p H[p][j-1] + s(i,j) == hv            -> first p (list order) reaching the maximum
 *   F     max_p oF[p][j] == hv                      -> enter the F walk at (i,j) with gv = hv
 *   O     max_p oO[p][j] == hv                      -> likewise
 *   E     exists k >= 1: H[i][j-k] + g + (k-1)e == hv; only k <= 1 + (g-q)/(c-e) can match in the
 *         convex model (Q[i][j] <= hv bounds H[i][j-k] from above), any k otherwise
 *   Q     what is left; smallest k with H[i][j-k] + q + (k-1)c == hv
 * Walks: F at row r, value gv: first p with oF[p][j] == gv; OPEN iff H[p][j] + g == gv (then H state
 * at (p,j)), else gv -= e and the walk continues at p.  E at (i,j), value gv: OPEN iff
 * H[i][j-1] + g == gv, else gv -= e; so an E gap ends at the smallest k of the rule above.
 */
#include "poa_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define NEG (-(1 << 29))

typedef struct { int m, n, g, e, q, c, sw, convex; } vp_t;

static vp_t vnormalise(const poa_params_t *p) {
    vp_t r;
    r.m = p->m; r.n = p->n; r.g = p->g; r.e = p->e; r.q = p->q; r.c = p->c;
    r.sw = ((p->mode & 1) == POA_MODE_SW); r.convex = 0;
    if (r.g >= r.e) { r.e = r.g; r.q = r.g; r.c = r.g; }
    else if (r.g <= r.q || r.e >= r.c) { r.q = r.g; r.c = r.e; }
    else r.convex = 1;
    return r;
}

int poa_align_csr_vtb(int N, const uint8_t *codes, const int32_t *off, const int32_t *pred,
                      const uint8_t *sink, const uint8_t *seq, int L, const poa_params_t *pp,
                      int32_t *out_node, int32_t *out_pos, int32_t *score) {
    const vp_t P = vnormalise(pp);
    if (score) *score = 0;
    if (N <= 0 || L <= 0) return 0;
    const size_t W = (size_t)L + 1;
    int32_t *H = (int32_t *)malloc(sizeof(int32_t) * 3 * ((size_t)N + 1) * W);
    int32_t *oF = H + ((size_t)N + 1) * W, *oO = oF + ((size_t)N + 1) * W;
    if (!H) { fprintf(stderr, "poa_vtb: out of memory\n"); abort(); }
#define AT(M, i, j) (M)[(size_t)(i) * W + (size_t)(j)]
    for (int j = 0; j <= L; ++j) {
        int h = 0;
        if (!P.sw && j > 0) { const int a = P.g + (j - 1) * P.e, b = P.q + (j - 1) * P.c; h = a > b ? a : b; }
        AT(H, 0, j) = h; AT(oF, 0, j) = h + P.g; AT(oO, 0, j) = h + P.q;   /* F = O = -inf in row 0 */
    }
    static const int32_t zero_pred = 0;
    int best = 0, bi = -1, bj = -1;
    for (int i = 1; i <= N; ++i) {
        int np = off[i] - off[i - 1];
        const int32_t *pl = pred + off[i - 1];
        if (np == 0) { np = 1; pl = &zero_pred; }
        const int code = codes[i - 1];
        int E = NEG, Q = NEG;
        for (int j = 0; j <= L; ++j) {
            int f = NEG * 2, o = NEG * 2, d = NEG * 2;
            for (int k = 0; k < np; ++k) {
                const int p = pl[k];
                if (AT(oF, p, j) > f) f = AT(oF, p, j);
                if (AT(oO, p, j) > o) o = AT(oO, p, j);
                if (j > 0 && AT(H, p, j - 1) > d) d = AT(H, p, j - 1);
            }
            int h = f > o ? f : o;
            if (j > 0) {
                d += (code == seq[j - 1]) ? P.m : P.n;
                const int e1 = AT(H, i, j - 1) + P.g, e2 = E + P.e; E = e1 > e2 ? e1 : e2;
                const int q1 = AT(H, i, j - 1) + P.q, q2 = Q + P.c; Q = q1 > q2 ? q1 : q2;
                if (d > h) h = d;
                if (E > h) h = E;
                if (Q > h) h = Q;
            }
            if (P.sw && h <= 0) h = 0;
            AT(H, i, j) = h;
            { const int a = h + P.g, b = f + P.e; AT(oF, i, j) = a > b ? a : b; }
            { const int a = h + P.q, b = o + P.c; AT(oO, i, j) = a > b ? a : b; }
            if (P.sw && h > best) { best = h; bi = i; bj = j; }
        }
        if (!P.sw && sink[i - 1] && (bi < 0 || AT(H, i, L) > best)) { best = AT(H, i, L); bi = i; bj = L; }
    }
    int n = 0;
    if (bi >= 0) {
        if (score) *score = best;
        const int kmax_e = P.convex ? 1 + (P.g - P.q) / (P.c - P.e) : L + 1;
        int i = bi, j = bj;
        enum { ST_H, ST_F, ST_O } st = ST_H;
        int gv = 0;
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
                const int hv = AT(H, i, j);
                if (P.sw && hv == 0) break;
                int src = 0;   /* 1 D, 2 F, 3 O */
                if (j > 0) {
                    int d = NEG * 2, dp = 0;
                    for (int k = 0; k < np; ++k) if (AT(H, pl[k], j - 1) > d) { d = AT(H, pl[k], j - 1); dp = pl[k]; }
                    if (d + ((codes[i - 1] == seq[j - 1]) ? P.m : P.n) == hv) {
                        out_node[n] = i - 1; out_pos[n] = j - 1; ++n;
                        i = dp; --j; src = 1;
                    }
                }
                if (!src) {
                    int f = NEG * 2, o = NEG * 2;
                    for (int k = 0; k < np; ++k) {
                        if (AT(oF, pl[k], j) > f) f = AT(oF, pl[k], j);
                        if (AT(oO, pl[k], j) > o) o = AT(oO, pl[k], j);
                    }
                    if (f == hv) { st = ST_F; gv = hv; src = 2; }
                    else if (o == hv) { st = ST_O; gv = hv; src = 3; }
                }
                if (!src) {   /* a gap in the graph: E, else Q */
                    int k = 0, isq = 0;
                    for (int x = 1; x <= j && x <= kmax_e; ++x)
                        if (AT(H, i, j - x) + P.g + (x - 1) * P.e == hv) { k = x; break; }
                    if (!k) {
                        isq = 1;
                        for (int x = 1; x <= j; ++x)
                            if (AT(H, i, j - x) + P.q + (x - 1) * P.c == hv) { k = x; break; }
                    }
                    if (!k) { fprintf(stderr, "poa_vtb: no source for cell (%d,%d)\n", i, j); abort(); }
                    (void)isq;
                    for (int x = 0; x < k; ++x) { out_node[n] = -1; out_pos[n] = j - 1; ++n; --j; }
                }
            } else {
                const int32_t *M = st == ST_F ? oF : oO;
                const int go = st == ST_F ? P.g : P.q, ge = st == ST_F ? P.e : P.c;
                int p = -1;
                for (int k = 0; k < np; ++k) if (AT(M, pl[k], j) == gv) { p = pl[k]; break; }
                if (p < 0) { fprintf(stderr, "poa_vtb: no predecessor carries the gap at (%d,%d)\n", i, j); abort(); }
                out_node[n] = i - 1; out_pos[n] = -1; ++n;
                i = p;
                if (AT(H, p, j) + go == gv) st = ST_H; else gv -= ge;
            }
        }
        for (int a = 0, b = n - 1; a < b; ++a, --b) {
            int32_t x = out_node[a]; out_node[a] = out_node[b]; out_node[b] = x;
            x = out_pos[a]; out_pos[a] = out_pos[b]; out_pos[b] = x;
        }
    }
#undef AT
    free(H);
    return n;
}
