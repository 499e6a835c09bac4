/*
 * poa_oracle.c -- CPU ORACLE (test infrastructure, PARITY UNPINNED; see poa_oracle.h).
 *
 * Restates, in scalar C, what the reference asks of spoa at
 *   src/smooth.cpp:752-755  AlignmentEngine::Create(type, m,n,g,e,q,c)
 *   src/smooth.cpp:761      Align(seq, graph)          -> poa_align()
 *   src/smooth.cpp:764      AddAlignment(aln,seq,w)    -> poa_add_alignment()
 *   src/smooth.cpp:773      GenerateConsensus()        -> poa_consensus()
 *   src/smooth.cpp:785-786  GenerateMultipleSequenceAlignment() -> poa_msa()
 *   src/smooth.cpp:716      XXH64(seq,len,0)           -> poa_xxh64()
 * spoa itself is absent (deps/spoa is an empty submodule), so the recurrences follow the
 * published algorithms and every tie-break is a decree of THIS file:
 *
 * SEMANTICS (the spec the HIP path must reproduce bit-for-bit)
 *  S1 gap model (as spoa's Create is documented to choose): g>=e -> linear (e=q=c=g);
 *     else g<=q || e>=c -> affine (q=g, c=e); else convex (two-piece).  One general
 *     two-piece recurrence is evaluated; linear/affine are its degenerate cases.
 *  S2 matrix: row 0 = virtual source, row r+1 = node of topological rank r, columns 0..L.
 *     pred rows of a node = rank+1 of its in-edge tails in EDGE INSERTION ORDER; a node
 *     without in-edges has the single pred row 0.
 *     F[i][j] = max_p max(H[p][j]+g, F[p][j]+e)      (gap in the sequence, piece 1)
 *     O[i][j] = max_p max(H[p][j]+q, O[p][j]+c)      (piece 2)
 *     D[i][j] = max_p H[p][j-1] + (code_i==seq[j-1] ? m : n)         (j>=1)
 *     E[i][j] = max(H[i][j-1]+g, E[i][j-1]+e), Q likewise with q,c   (j>=1)
 *     H[i][j] = max(D,F,O,E,Q); SW additionally clamps at 0.
 *     Row 0: SW H=0; NW H[0][0]=0, H[0][j]=max(g+(j-1)e, q+(j-1)c); F=O=-inf.
 *  S3 ties: the FIRST candidate in this order wins: H: STOP(SW,H==0) > D > F > O > E > Q;
 *     D: preds in list order; F/O: preds in list order, within a pred OPEN before EXTEND;
 *     E/Q: OPEN before EXTEND.  ("later candidate replaces only if strictly greater".)
 *  S4 end cell: SW = first cell in (row,col) order with the strictly greatest H>0 (none ->
 *     empty alignment); NW = first sink row (no out-edge) in rank order with the strictly
 *     greatest H[.][L].
 *  S5 traceback = replay of the recorded choices; emits (node,pos) pairs: D -> (node,j-1),
 *     F/O step -> (node,-1), E/Q step -> (-1,j-1); SW stops at STOP, NW at (0,0).
 *  S6 AddAlignment: positions without an aligned node become new nodes; an aligned
 *     position reuses the node (same letter) or its aligned sibling with that letter, else
 *     a new sibling joins the aligned group.  New node ids are handed out in SEQUENCE
 *     ORDER.  Each consecutive pair of path nodes gets edge weight += 2*weight (both end
 *     bases contribute, as spoa documents); new edges are appended to the tail's out-list
 *     and the head's in-list.
 *  S7 topological order is maintained INCREMENTALLY (any valid order with contiguous
 *     aligned groups is a legal POA order; this one is data-parallel): a new sibling is
 *     placed right after the last old member of its group; a run of new unaligned nodes is
 *     placed right before the group of the next aligned path node (or right after the
 *     group of the previous one if there is none, or at the end).  Same slot -> sequence
 *     order.  new_rank(k-th new node) = slot_k + k.
 *  S8 consensus = heaviest bundle with branch completion; MSA column = aligned group in
 *     rank order.
 *
 * BANDED MODE (A11: the reference's abPOA path, src/smooth.cpp:133-627, always runs abPOA with its adaptive
 * band wb=311, wf=0.03, :266-271; abPOA itself is absent).  By decree:
 *  B1 w = min(wb + (int)(wf * L), 693) columns on either side: abPOA's band size, capped (beyond L = 12 733) so
 *     that a band never spans more than 128 strips -- what one wavefront's window holds in the device kernel.
 *  B2 the band of a row is CENTRED ON THE BACKBONE COORDINATE x of its node and is a whole number of s-column
 *     strips: strips max(0, x - w) / s .. (x + w) / s, where s is the narrowest of 6, 8, 11 with which the band of
 *     the BLOCK's longest sequence spans at most 128 strips (poa_band_strip_width; a stand-alone alignment: its own
 *     length).  x is kept by AddAlignment: the first sequence's nodes
 *     get their own column (i + 1); a new sibling takes the x of the node it is aligned to; a run of new
 *     unaligned nodes continues from the previous aligned position (x + distance), else counts back from the
 *     next one, else is its own column.  (abPOA moves its band with the best-scoring cells of the predecessor
 *     rows; a band known before the row is computed is what lets the MI355X sweep slide a one-wave window
 *     along the diagonal.)  The virtual source row is not banded.
 *  B3 a cell outside its row's band does not exist: H and both outgoing gap candidates are -inf for every
 *     reader, in-row gaps start at the band's first column.  cells = sum of band widths.  The backbone band (banded = 1)
 *     is for local mode only (a global alignment ends in column L of a sink row, which a band around backbone
 *     coordinates need not hold); the adaptive band B4 serves both modes, as smooth_abpoa sets it (src/smooth.cpp:259-271).
 *  B4 ADAPTIVE band (banded = 2; abPOA's published rule -- Gao et al. 2021, "adaptive banding": the band of a node
 *     follows the best-scoring cells of its predecessors and the node's position in the graph).  For the row of node v:
 *       remain(v) = number of edges of the walk that leaves v by its heaviest out-edge (the first of greatest weight in
 *                   out-list order), and so on from node to node, until a node without out-edges;
 *       c   = L - remain(v)                                   (where v would sit if the rest of the query ran along it)
 *       P_l = min over the predecessor rows p of ml(p) + 1,   P_r = max over them of mr(p) + 1
 *             ml(p) / mr(p) = leftmost / rightmost column of row p's band holding the row's greatest H; virtual row: 0, 0
 *       the band covers the columns  max(0, min(P_l, c) - w) .. min(L, max(P_r, c) + w),  w as B1,
 *     in whole strips as B2.  While the greatest H of the first rows of a local alignment is still ambiguous (a 1-mer,
 *     2-mer ... match somewhere) P_l and P_r lie far apart; a band of more than 128 strips keeps the 128 strips
 *     from max(first, min(strip(c) - 64, last - 127)) on (one wavefront's window; strip(c) = clamp(c, 0, L) / s).
 *     B3 applies unchanged.  Stand-alone alignments over a caller's CSR have no edge weights: B4 is a whole-block mode.
 */
#include "poa_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NEG (-(1 << 29))
enum { SRC_STOP = 0, SRC_D = 1, SRC_F = 2, SRC_O = 3, SRC_E = 4, SRC_Q = 5 };
#define TB_FEXT 0x08
#define TB_OEXT 0x10
#define TB_EEXT 0x20
#define TB_QEXT 0x40

struct poa_graph {
    int n_nodes, cap_nodes;
    uint8_t *code;
    int32_t *rank;   /* node -> rank */
    int32_t *order;  /* rank -> node */
    int32_t *leader; /* node -> first node of its aligned group */
    int32_t *xpos;   /* node -> backbone coordinate (B2) */
    int32_t *via;    /* node -> the node it was created aligned to (-1: created unaligned): fixes spoa's aligned-node list order (S7') */
    int32_t *gmem;   /* [leader*5 + code] -> node | -1 (valid at leader rows) */
    int32_t *in_head, *in_tail, *out_head, *out_tail, *in_deg, *out_deg;
    int n_edges, cap_edges;
    int32_t *e_tail, *e_head, *e_next_in, *e_next_out;
    uint32_t *e_w;
    int n_seqs, cap_seqs;
    int64_t *seq_off; /* n_seqs+1 */
    int32_t *path;
    int64_t cap_path;
};

static void *xrealloc(void *p, size_t n) {
    void *r = realloc(p, n ? n : 1);
    if (!r) { fprintf(stderr, "poa_oracle: out of memory\n"); abort(); }
    return r;
}

poa_graph_t *poa_graph_new(void) {
    poa_graph_t *g = (poa_graph_t *)calloc(1, sizeof(*g));
    g->seq_off = (int64_t *)calloc(1, sizeof(int64_t));
    return g;
}
void poa_graph_free(poa_graph_t *g) {
    if (!g) return;
    free(g->code); free(g->rank); free(g->order); free(g->leader); free(g->gmem); free(g->xpos); free(g->via);
    free(g->in_head); free(g->in_tail); free(g->out_head); free(g->out_tail);
    free(g->in_deg); free(g->out_deg);
    free(g->e_tail); free(g->e_head); free(g->e_next_in); free(g->e_next_out); free(g->e_w);
    free(g->seq_off); free(g->path);
    free(g);
}
int poa_graph_num_nodes(const poa_graph_t *g) { return g->n_nodes; }
int poa_graph_num_edges(const poa_graph_t *g) { return g->n_edges; }
int poa_graph_num_seqs(const poa_graph_t *g) { return g->n_seqs; }

static void reserve_nodes(poa_graph_t *g, int n) {
    if (n <= g->cap_nodes) return;
    int c = g->cap_nodes ? g->cap_nodes : 1024;
    while (c < n) c *= 2;
    g->code = (uint8_t *)xrealloc(g->code, c);
#define RS(f) g->f = (int32_t *)xrealloc(g->f, sizeof(int32_t) * (size_t)c)
    RS(rank); RS(order); RS(leader); RS(xpos); RS(via); RS(in_head); RS(in_tail); RS(out_head); RS(out_tail);
    RS(in_deg); RS(out_deg);
#undef RS
    g->gmem = (int32_t *)xrealloc(g->gmem, sizeof(int32_t) * 5 * (size_t)c);
    g->cap_nodes = c;
}
static void reserve_edges(poa_graph_t *g, int n) {
    if (n <= g->cap_edges) return;
    int c = g->cap_edges ? g->cap_edges : 1024;
    while (c < n) c *= 2;
#define RS(f) g->f = (int32_t *)xrealloc(g->f, sizeof(int32_t) * (size_t)c)
    RS(e_tail); RS(e_head); RS(e_next_in); RS(e_next_out);
#undef RS
    g->e_w = (uint32_t *)xrealloc(g->e_w, sizeof(uint32_t) * (size_t)c);
    g->cap_edges = c;
}

static int new_node(poa_graph_t *g, uint8_t code) {
    reserve_nodes(g, g->n_nodes + 1);
    int v = g->n_nodes++;
    g->code[v] = code;
    g->rank[v] = -1;
    g->xpos[v] = 0;
    g->via[v] = -1;
    g->leader[v] = v;
    for (int c = 0; c < 5; ++c) g->gmem[5 * v + c] = -1;
    g->gmem[5 * v + code] = v;
    g->in_head[v] = g->in_tail[v] = g->out_head[v] = g->out_tail[v] = -1;
    g->in_deg[v] = g->out_deg[v] = 0;
    return v;
}

/* S6: edge (u,v) += w; appended to u's out-list / v's in-list when new. */
static void add_edge(poa_graph_t *g, int u, int v, uint32_t w) {
    for (int e = g->out_head[u]; e >= 0; e = g->e_next_out[e])
        if (g->e_head[e] == v) { g->e_w[e] += w; return; }
    reserve_edges(g, g->n_edges + 1);
    int e = g->n_edges++;
    g->e_tail[e] = u; g->e_head[e] = v; g->e_w[e] = w;
    g->e_next_in[e] = g->e_next_out[e] = -1;
    if (g->out_tail[u] >= 0) g->e_next_out[g->out_tail[u]] = e; else g->out_head[u] = e;
    g->out_tail[u] = e; g->out_deg[u]++;
    if (g->in_tail[v] >= 0) g->e_next_in[g->in_tail[v]] = e; else g->in_head[v] = e;
    g->in_tail[v] = e; g->in_deg[v]++;
}

/* ------------------------------------------------------------------------------------ */
/* The DP over a CSR-in-rank-space graph (S1-S5).                                        */

typedef struct {
    int m, n, g, e, q, c, sw;
} norm_params_t;

static norm_params_t normalise(const poa_params_t *p) {
    norm_params_t r;
    r.m = p->m; r.n = p->n; r.g = p->g; r.e = p->e; r.q = p->q; r.c = p->c;
    r.sw = ((p->mode & 1) == POA_MODE_SW);
    if (r.g >= r.e) { r.e = r.g; r.q = r.g; r.c = r.g; }            /* linear  */
    else if (r.g <= r.q || r.e >= r.c) { r.q = r.g; r.c = r.e; }    /* affine  */
    return r;                                                        /* convex  */
}

/* Per-thread workspace: every buffer align_rows / poa_align need is grown on demand and kept, so a block
 * (and every later block of the same OpenMP thread) allocates nothing per alignment.  The first baseline
 * malloc'ed ~25 MB per alignment: 256 threads then spend their time in mmap/munmap and page faults
 * (4.6 Mcells/s/thread against 70 single-threaded), which measured the allocator, not the DP.       */
enum { WS_HM, WS_LAST, WS_FHEAD, WS_FNEXT, WS_SPARE, WS_TB, WS_MP, WS_TBX, WS_CODES, WS_SINK, WS_OFF, WS_PRED,
       WS_ROWNODE, WS_HINT, WS_NBUF };
struct poa_ws {
    void *buf[WS_NBUF];
    size_t cap[WS_NBUF];
    int32_t **rows;      /* every DP row buffer ever allocated (3 * row_cap ints each) */
    int n_rows, cap_rows;
    size_t row_cap;
    int impl;            /* POA_IMPL_SCALAR | POA_IMPL_AVX2 */
    poa_simd_ws_t *simd;
};
poa_ws_t *poa_ws_new(void) { return (poa_ws_t *)calloc(1, sizeof(poa_ws_t)); }
void poa_ws_free(poa_ws_t *w) {
    if (!w) return;
    for (int k = 0; k < WS_NBUF; ++k) free(w->buf[k]);
    for (int k = 0; k < w->n_rows; ++k) free(w->rows[k]);
    free(w->rows);
    poa_simd_ws_free(w->simd);
    free(w);
}
void poa_ws_set_impl(poa_ws_t *w, int impl) { w->impl = impl; }
static void *ws_get(poa_ws_t *w, int k, size_t bytes) {
    if (bytes > w->cap[k]) {
        free(w->buf[k]);
        w->cap[k] = bytes + bytes / 4 + 64;
        w->buf[k] = malloc(w->cap[k]);
        if (!w->buf[k]) { fprintf(stderr, "poa_oracle: out of memory\n"); abort(); }
    }
    return w->buf[k];
}

static int32_t *ws_new_row(poa_ws_t *w) {
    if (w->n_rows == w->cap_rows) {
        w->cap_rows = w->cap_rows ? 2 * w->cap_rows : 1024;
        w->rows = (int32_t **)xrealloc(w->rows, sizeof(int32_t *) * (size_t)w->cap_rows);
    }
    int32_t *r = (int32_t *)malloc(sizeof(int32_t) * w->row_cap);
    if (!r) { fprintf(stderr, "poa_oracle: out of memory\n"); abort(); }
    w->rows[w->n_rows++] = r;
    return r;
}

/* banded = params.banded: 0 = off, 1 = backbone band (B2), 2 = adaptive band (B4), strip width from L;
 * 6 / 8 / 11 = backbone band with that strip width, 0x80 | strip = adaptive band with that strip width */
static int band_is_adaptive(unsigned banded) { return banded == 2 || (banded & 0x80) != 0; }
static int band_strip_of(unsigned banded, int L) { const int s = (int)(banded & 0x7f); return s > 2 ? s : poa_band_strip_width(L); }

/* hint != NULL: banded mode (B1-B3: hint = backbone coordinate of every row; B4: hint = remain() of every row),
 * *band_cells receives the number of cells inside the bands */
static int align_rows(poa_ws_t *ws, int N, const uint8_t *codes, const int32_t *off, const int32_t *pred,
                      const uint8_t *sink, const int32_t *row_node, const uint8_t *seq, int L,
                      const poa_params_t *pp, int32_t *out_node, int32_t *out_pos,
                      int32_t *score, const int32_t *hint, uint64_t *band_cells) {
    poa_ws_t *own_ws = NULL;
    if (!ws) ws = own_ws = poa_ws_new();
    norm_params_t P = normalise(pp);
    if (score) *score = 0;
    if (N <= 0 || L <= 0) { poa_ws_free(own_ws); return 0; }
    for (int i = 1; i <= N; ++i) if (off[i] - off[i - 1] > 65535) { fprintf(stderr, "poa_oracle: in-degree > 65535\n"); abort(); }
    const size_t W = (size_t)L + 1;
    /* rows of H,F,O kept until their last reader is done */
    int32_t **Hm = (int32_t **)ws_get(ws, WS_HM, ((size_t)N + 1) * sizeof(int32_t *));
    memset(Hm, 0, ((size_t)N + 1) * sizeof(int32_t *));
    int32_t *last_use = (int32_t *)ws_get(ws, WS_LAST, sizeof(int32_t) * ((size_t)N + 1));
    for (int i = 0; i <= N; ++i) last_use[i] = i;
    for (int i = 1; i <= N; ++i)
        for (int k = off[i - 1]; k < off[i]; ++k)
            if (last_use[pred[k]] < i) last_use[pred[k]] = i;
    /* free lists keyed by last_use */
    int32_t *free_head = (int32_t *)ws_get(ws, WS_FHEAD, sizeof(int32_t) * ((size_t)N + 2));
    int32_t *free_next = (int32_t *)ws_get(ws, WS_FNEXT, sizeof(int32_t) * ((size_t)N + 1));
    for (int i = 0; i <= N + 1; ++i) free_head[i] = -1;
    for (int i = 0; i <= N; ++i) { free_next[i] = free_head[last_use[i]]; free_head[last_use[i]] = i; }

    /* row buffers: the workspace keeps every buffer it ever allocated; all of them are spare at the start */
    if (3 * W > ws->row_cap) {
        for (int k = 0; k < ws->n_rows; ++k) free(ws->rows[k]);
        ws->n_rows = 0;
        ws->row_cap = 3 * W + 3 * W / 4;
    }
    int32_t **spare = (int32_t **)ws_get(ws, WS_SPARE, sizeof(int32_t *) * ((size_t)N + 2 + (size_t)ws->n_rows));
    int n_spare = 0;
    for (int k = 0; k < ws->n_rows; ++k) spare[n_spare++] = ws->rows[k];
#define ROW_ALLOC() (n_spare ? spare[--n_spare] : ws_new_row(ws))
#define ROW_FREE(p) (spare[n_spare++] = (p))
    uint8_t *tb = (uint8_t *)ws_get(ws, WS_TB, ((size_t)N + 1) * W);
    /* ordinals of the winning pred for D,F,O on multi-pred rows */
    int32_t *mp_index = (int32_t *)ws_get(ws, WS_MP, sizeof(int32_t) * ((size_t)N + 1));
    size_t n_mp = 0;
    for (int i = 1; i <= N; ++i) mp_index[i] = (off[i] - off[i - 1] > 1) ? (int32_t)n_mp++ : -1;
    uint16_t *tbx = (uint16_t *)ws_get(ws, WS_TBX, sizeof(uint16_t) * 3 * (n_mp ? n_mp : 1) * W); /* ordinals < 65536 */

    /* row 0 */
    Hm[0] = ROW_ALLOC();
    {
        int32_t *H = Hm[0], *F = H + W, *O = F + W;
        for (int j = 0; j <= L; ++j) {
            F[j] = O[j] = NEG;
            if (P.sw || j == 0) H[j] = 0;
            else {
                int a = P.g + (j - 1) * P.e, b = P.q + (j - 1) * P.c;
                H[j] = a > b ? a : b;
            }
        }
    }
    int best_i = -1, best_j = -1, best = 0;
    static const int32_t zero_pred = 0;
    const int adaptive = hint && band_is_adaptive(pp->banded);
    int32_t *mlr = NULL;   /* B4: ml, mr of every row */
    if (adaptive) { mlr = (int32_t *)malloc(sizeof(int32_t) * 2 * ((size_t)N + 1)); mlr[0] = 0; mlr[1] = 0; }
    for (int i = 1; i <= N; ++i) {
        int np = off[i] - off[i - 1];
        const int32_t *pl = pred + off[i - 1];
        if (np == 0) { np = 1; pl = &zero_pred; }
        int32_t *H = ROW_ALLOC(), *F = H + W, *O = F + W;
        Hm[i] = H;
        uint8_t *t = tb + (size_t)i * W;
        uint16_t *tx = mp_index[i] >= 0 ? tbx + 3 * (size_t)mp_index[i] * W : NULL;
        const int code = codes[i - 1];
        int E = NEG, Q = NEG;
        int beg = 0, end = L;
        if (hint) {   /* B1, B2 / B4 */
            const int strip = band_strip_of(pp->banded, L);
            const int w0 = POA_BAND_WB + (int)(POA_BAND_WF * L), w = w0 < POA_BAND_WMAX ? w0 : POA_BAND_WMAX, x = hint[i - 1];
            if (!adaptive) {
                beg = ((x - w > 0 ? x - w : 0) / strip) * strip;
                end = ((x + w) / strip) * strip + strip - 1;
            } else {
                const int c = L - x;
                int p_l = 0x7fffffff, p_r = -1;
                for (int k = 0; k < np; ++k) {
                    const int l = mlr[2 * pl[k]] + 1, r = mlr[2 * pl[k] + 1] + 1;
                    if (l < p_l) p_l = l;
                    if (r > p_r) p_r = r;
                }
                int b0 = (p_l < c ? p_l : c) - w, e0 = (p_r > c ? p_r : c) + w;
                if (b0 < 0) b0 = 0;
                if (e0 > L) e0 = L;
                int sb = b0 / strip, se = e0 / strip;
                if (se - sb + 1 > 128) {
                    const int sc = (c < 0 ? 0 : (c > L ? L : c)) / strip;
                    int first = sc - 64;
                    if (first > se - 127) first = se - 127;
                    if (first < sb) first = sb;
                    sb = first; se = first + 127;
                }
                beg = sb * strip;
                end = se * strip + strip - 1;
            }
            if (end > L) end = L;
            if (band_cells && beg <= end) *band_cells += (uint64_t)(end - beg + 1);
        }
        for (int j = 0; j <= L; ++j) {
            if (j < beg || j > end) {   /* B3 */
                H[j] = NEG; F[j] = NEG; O[j] = NEG; t[j] = 0; E = NEG; Q = NEG;
                if (tx) { tx[3 * (size_t)j] = 0; tx[3 * (size_t)j + 1] = 0; tx[3 * (size_t)j + 2] = 0; }
                continue;
            }
            int f = 0, o = 0, d = NEG, fo = 0, oo = 0, dd = 0, fx = 0, ox = 0;
            for (int k = 0; k < np; ++k) {
                const int32_t *Hp = Hm[pl[k]], *Fp = Hp + W, *Op = Fp + W;
                int c1 = Hp[j] + P.g, c2 = Fp[j] + P.e;
                if (k == 0 || c1 > f) { f = c1; fo = k; fx = 0; }
                if (c2 > f) { f = c2; fo = k; fx = 1; }
                c1 = Hp[j] + P.q; c2 = Op[j] + P.c;
                if (k == 0 || c1 > o) { o = c1; oo = k; ox = 0; }
                if (c2 > o) { o = c2; oo = k; ox = 1; }
                if (j > 0) { int c3 = Hp[j - 1]; if (k == 0 || c3 > d) { d = c3; dd = k; } }
            }
            int ex = 0, qx = 0;
            if (j > 0) {
                d += (code == seq[j - 1]) ? P.m : P.n;
                int c1 = H[j - 1] + P.g, c2 = E + P.e;
                E = c1; if (c2 > c1) { E = c2; ex = 1; }
                c1 = H[j - 1] + P.q; c2 = Q + P.c;
                Q = c1; if (c2 > c1) { Q = c2; qx = 1; }
            } else { d = NEG; E = NEG; Q = NEG; }
            int h = d, src = SRC_D;
            if (f > h) { h = f; src = SRC_F; }
            if (o > h) { h = o; src = SRC_O; }
            if (E > h) { h = E; src = SRC_E; }
            if (Q > h) { h = Q; src = SRC_Q; }
            if (P.sw && h <= 0) { h = 0; src = SRC_STOP; }
            H[j] = h; F[j] = f; O[j] = o;
            t[j] = (uint8_t)(src | (fx ? TB_FEXT : 0) | (ox ? TB_OEXT : 0) | (ex ? TB_EEXT : 0) |
                             (qx ? TB_QEXT : 0));
            if (tx) { tx[3 * (size_t)j] = (uint16_t)dd; tx[3 * (size_t)j + 1] = (uint16_t)fo; tx[3 * (size_t)j + 2] = (uint16_t)oo; }
            if (P.sw && h > best) { best = h; best_i = i; best_j = j; }
        }
        if (!P.sw && sink[i - 1] && (best_i < 0 || H[L] > best)) { best = H[L]; best_i = i; best_j = L; }
        if (adaptive) {   /* B4: leftmost / rightmost column of the band holding the row's greatest H */
            int mx = H[beg], l = beg, r = beg;
            for (int j = beg + 1; j <= end; ++j) {
                if (H[j] > mx) { mx = H[j]; l = j; r = j; }
                else if (H[j] == mx) r = j;
            }
            mlr[2 * i] = l; mlr[2 * i + 1] = r;
        }
        /* release rows nobody will read again */
        for (int r = free_head[i]; r >= 0; r = free_next[r]) { ROW_FREE(Hm[r]); Hm[r] = NULL; }
    }
#undef ROW_ALLOC
#undef ROW_FREE
    free(mlr);

    int n = 0;
    if (best_i >= 0) {
        if (score) *score = best;
        int i = best_i, j = best_j, st = SRC_STOP; /* st: STOP == "in H" */
        for (;;) {
            if (i == 0) { /* virtual row: H[0][j] = 0 ends a local alignment */
                if (j == 0 || P.sw) break;
                out_node[n] = -1; out_pos[n] = j - 1; ++n; --j;
                continue;
            }
            const uint8_t t = tb[(size_t)i * W + j];
            const uint16_t *tx = mp_index[i] >= 0 ? tbx + 3 * ((size_t)mp_index[i] * W + j) : NULL;
            const int32_t *pl = pred + off[i - 1];
            const int has = off[i] - off[i - 1];
            if (st == SRC_STOP) {
                int src = t & 7;
                if (src == SRC_STOP) break;
                if (src == SRC_D) {
                    out_node[n] = row_node[i - 1]; out_pos[n] = j - 1; ++n;
                    i = has ? pl[tx ? tx[0] : 0] : 0; --j;
                } else st = src;
            } else if (st == SRC_F || st == SRC_O) {
                int ext = st == SRC_F ? (t & TB_FEXT) : (t & TB_OEXT);
                int ord = tx ? tx[st == SRC_F ? 1 : 2] : 0;
                out_node[n] = row_node[i - 1]; out_pos[n] = -1; ++n;
                i = has ? pl[ord] : 0;
                if (!ext) st = SRC_STOP;
            } else { /* E or Q */
                int ext = st == SRC_E ? (t & TB_EEXT) : (t & TB_QEXT);
                out_node[n] = -1; out_pos[n] = j - 1; ++n; --j;
                if (!ext) st = SRC_STOP;
            }
        }
        for (int a = 0, b = n - 1; a < b; ++a, --b) {
            int32_t x = out_node[a]; out_node[a] = out_node[b]; out_node[b] = x;
            x = out_pos[a]; out_pos[a] = out_pos[b]; out_pos[b] = x;
        }
    }
    poa_ws_free(own_ws);
    return n;
}

/* Stand-alone entry over a caller-supplied CSR (used to check the HIP align-only path). */
int poa_align_csr(int N, const uint8_t *codes, const int32_t *off, const int32_t *pred,
                  const uint8_t *sink, const uint8_t *seq, int L, const poa_params_t *p,
                  int32_t *out_node, int32_t *out_pos, int32_t *score) {
    int32_t *row_node = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N ? N : 1));
    for (int i = 0; i < N; ++i) row_node[i] = i; /* report ranks */
    int n = align_rows(NULL, N, codes, off, pred, sink, row_node, seq, L, p, out_node, out_pos, score, NULL, NULL);
    free(row_node);
    return n;
}

void poa_graph_rows(const poa_graph_t *g, uint8_t *codes, int32_t *off, int32_t *pred,
                    uint8_t *sink, int32_t *row_node) {
    int k = 0;
    off[0] = 0;
    for (int r = 0; r < g->n_nodes; ++r) {
        int v = g->order[r];
        codes[r] = g->code[v];
        sink[r] = g->out_deg[v] == 0;
        if (row_node) row_node[r] = v;
        for (int e = g->in_head[v]; e >= 0; e = g->e_next_in[e]) pred[k++] = g->rank[g->e_tail[e]] + 1;
        off[r + 1] = k;
    }
}

int poa_align_ws(poa_ws_t *ws, const poa_graph_t *g, const uint8_t *seq, int len, const poa_params_t *p,
                 int32_t *out_node, int32_t *out_pos, int32_t *score, uint64_t *cells) {
    int N = g->n_nodes;
    if (cells) *cells = (uint64_t)N * (uint64_t)len;
    if (score) *score = 0;
    if (N == 0 || len == 0) return 0;
    poa_ws_t *own = NULL;
    if (!ws) ws = own = poa_ws_new();
    uint8_t *codes = (uint8_t *)ws_get(ws, WS_CODES, (size_t)N), *sink = (uint8_t *)ws_get(ws, WS_SINK, (size_t)N);
    int32_t *off = (int32_t *)ws_get(ws, WS_OFF, sizeof(int32_t) * ((size_t)N + 1));
    int32_t *pred = (int32_t *)ws_get(ws, WS_PRED, sizeof(int32_t) * ((size_t)g->n_edges + 1));
    int32_t *row_node = (int32_t *)ws_get(ws, WS_ROWNODE, sizeof(int32_t) * (size_t)N);
    poa_graph_rows(g, codes, off, pred, sink, row_node);
    int n = -1;
    const int banded = p->banded && ((p->mode & 1) == POA_MODE_SW || band_is_adaptive(p->banded));   /* global: the adaptive band only */
    if (ws->impl == POA_IMPL_AVX2 && !banded) {   /* (-1: no AVX2 on this host, or the scores leave int16: scalar path) */
        if (!ws->simd) ws->simd = poa_simd_ws_new();
        n = poa_align_rows_simd(ws->simd, N, codes, off, pred, sink, row_node, seq, len, p, out_node, out_pos, score);
    }
    if (n < 0) {
        int32_t *hint = NULL;
        uint64_t bc = 0;
        if (banded) {
            hint = (int32_t *)ws_get(ws, WS_HINT, sizeof(int32_t) * (size_t)N);
            if (band_is_adaptive(p->banded)) poa_graph_row_remain(g, hint); else poa_graph_row_hints(g, hint);
        }
        n = align_rows(ws, N, codes, off, pred, sink, row_node, seq, len, p, out_node, out_pos, score, hint, &bc);
        if (banded && cells) *cells = bc;
    }
    poa_ws_free(own);
    return n;
}
int poa_align(const poa_graph_t *g, const uint8_t *seq, int len, const poa_params_t *p,
              int32_t *out_node, int32_t *out_pos, int32_t *score, uint64_t *cells) {
    return poa_align_ws(NULL, g, seq, len, p, out_node, out_pos, score, cells);
}

/* ------------------------------------------------------------------------------------ */
/* S6 + S7                                                                               */

static int group_start(const poa_graph_t *g, int leader, int n_old) {
    int r = 0x7fffffff;
    for (int c = 0; c < 5; ++c) {
        int v = g->gmem[5 * leader + c];
        if (v >= 0 && v < n_old && g->rank[v] < r) r = g->rank[v];
    }
    return r;
}
static int group_end(const poa_graph_t *g, int leader, int n_old) {
    int r = -1;
    for (int c = 0; c < 5; ++c) {
        int v = g->gmem[5 * leader + c];
        if (v >= 0 && v < n_old && g->rank[v] > r) r = g->rank[v];
    }
    return r;
}

void poa_add_alignment(poa_graph_t *g, const int32_t *aln_node, const int32_t *aln_pos,
                       int n_pairs, const uint8_t *seq, int len, uint32_t weight) {
    if (len <= 0) return;
    const int n_old = g->n_nodes;
    int32_t *posnode = (int32_t *)malloc(sizeof(int32_t) * (size_t)len);
    int32_t *target = (int32_t *)malloc(sizeof(int32_t) * (size_t)len);
    int8_t *kind = (int8_t *)malloc((size_t)len); /* 0 old, 1 new sibling, 2 new unaligned */
    for (int i = 0; i < len; ++i) posnode[i] = -1;
    for (int k = 0; k < n_pairs; ++k)
        if (aln_pos[k] >= 0 && aln_node[k] >= 0) posnode[aln_pos[k]] = aln_node[k];

    for (int i = 0; i < len; ++i) {
        const uint8_t c = seq[i] > 4 ? 4 : seq[i];
        const int a = posnode[i];
        if (a >= 0) {
            const int ld = g->leader[a];
            int v = g->gmem[5 * ld + c];
            if (v >= 0) { target[i] = v; kind[i] = 0; }
            else {
                v = new_node(g, c);
                g->leader[v] = ld;
                g->gmem[5 * ld + c] = v;
                g->xpos[v] = g->xpos[a];
                g->via[v] = a;
                target[i] = v; kind[i] = 1;
            }
        } else { target[i] = new_node(g, c); kind[i] = 2; }
    }
    /* B2: backbone coordinates of the new unaligned nodes (aligned positions: posnode[.] >= 0, old nodes) */
    for (int i = 0; i < len; ++i) {
        if (kind[i] != 2) continue;
        int p = i - 1, s2 = i + 1;
        while (p >= 0 && kind[p] == 2) --p;
        while (s2 < len && kind[s2] == 2) ++s2;
        g->xpos[target[i]] = p >= 0 ? g->xpos[posnode[p]] + (i - p) : (s2 < len ? g->xpos[posnode[s2]] - (s2 - i) : i + 1);
    }
    /* S7 slots, in sequence order */
    {
        const int n_new = g->n_nodes - n_old;
        int32_t *slot = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_new ? n_new : 1));
        int k = 0;
        for (int i = 0; i < len; ++i) {
            if (kind[i] == 0) continue;
            if (kind[i] == 1) slot[k++] = group_end(g, g->leader[target[i]], n_old) + 1;
            else {
                int s = i + 1;
                while (s < len && kind[s] == 2) ++s;
                int p = i - 1;
                while (p >= 0 && kind[p] == 2) --p;
                if (s < len) slot[k++] = group_start(g, g->leader[target[s]], n_old);
                else if (p >= 0) slot[k++] = group_end(g, g->leader[target[p]], n_old) + 1;
                else slot[k++] = n_old;
            }
        }
        /* merge */
        int32_t *new_order = (int32_t *)malloc(sizeof(int32_t) * (size_t)g->n_nodes);
        int r_old = 0, w = 0;
        for (k = 0; k < n_new; ++k) {
            while (r_old < slot[k]) new_order[w++] = g->order[r_old++];
            new_order[w++] = n_old + k;
        }
        while (r_old < n_old) new_order[w++] = g->order[r_old++];
        for (int r = 0; r < g->n_nodes; ++r) { g->order[r] = new_order[r]; g->rank[new_order[r]] = r; }
        free(new_order); free(slot);
    }
    for (int i = 1; i < len; ++i) add_edge(g, target[i - 1], target[i], 2 * weight);
    /* record the path */
    if (g->n_seqs + 2 > g->cap_seqs) {
        g->cap_seqs = g->cap_seqs ? 2 * g->cap_seqs : 64;
        g->seq_off = (int64_t *)xrealloc(g->seq_off, sizeof(int64_t) * ((size_t)g->cap_seqs + 1));
    }
    int64_t base = g->seq_off[g->n_seqs];
    if (base + len > g->cap_path) {
        g->cap_path = 2 * (base + len);
        g->path = (int32_t *)xrealloc(g->path, sizeof(int32_t) * (size_t)g->cap_path);
    }
    memcpy(g->path + base, target, sizeof(int32_t) * (size_t)len);
    g->seq_off[++g->n_seqs] = base + len;
    free(posnode); free(target); free(kind);
}

/* S7' -- spoa's topological order, AS RECOLLECTED (rvaser/spoa Graph::TopologicalSort is absent from the reference snapshot:
 * UNVERIFIED, offered as an option so that the one known divergence of S7 can be switched off and priced):
 *   for every node in id order, depth-first with an explicit stack: a node on top of the stack pushes the tails of its
 *   in-edges that are not done (in-edge insertion order) and -- unless it was itself pushed as somebody's aligned node
 *   ("ignored") -- its aligned nodes that are not done (marking them ignored); when nothing had to be pushed it is done,
 *   and unless ignored it is emitted, followed by its aligned nodes in the order of ITS aligned-node list.
 * A node's aligned-node list follows spoa's AddAlignment: a node created aligned to x starts with x's list followed by x,
 * and is appended to the list of every member the group already had.  Groups stay contiguous, so S8 applies unchanged. */
static int aligned_list(const poa_graph_t *g, int v, int *out) {
    int mem[5], nm = 0;
    const int ld = g->leader[v];
    for (int c = 0; c < 5; ++c) if (g->gmem[5 * ld + c] >= 0) mem[nm++] = g->gmem[5 * ld + c];
    for (int a = 1; a < nm; ++a) { int x = mem[a], b = a; while (b > 0 && mem[b - 1] > x) { mem[b] = mem[b - 1]; --b; } mem[b] = x; }   /* creation order */
    int list[5][5], ln[5];
    for (int k = 0; k < nm; ++k) {
        ln[k] = 0;
        int xi = -1;
        for (int j = 0; j < k; ++j) if (mem[j] == g->via[mem[k]]) xi = j;
        if (k > 0 && xi < 0) xi = 0;   /* (cannot happen: a member is created aligned to an earlier member) */
        if (xi >= 0) { for (int q = 0; q < ln[xi]; ++q) list[k][ln[k]++] = list[xi][q]; list[k][ln[k]++] = mem[xi]; }
        for (int j = 0; j < k; ++j) list[j][ln[j]++] = mem[k];
    }
    for (int k = 0; k < nm; ++k) if (mem[k] == v) { for (int q = 0; q < ln[k]; ++q) out[q] = list[k][q]; return ln[k]; }
    return 0;
}
static void spoa_resort(poa_graph_t *g) {
    const int n = g->n_nodes;
    uint8_t *marks = (uint8_t *)calloc((size_t)n + 1, 1), *ignored = (uint8_t *)calloc((size_t)n + 1, 1);
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * ((size_t)g->n_edges + 6 * (size_t)n + 8));
    int w = 0;
    for (int s0 = 0; s0 < n; ++s0) {
        if (marks[s0] != 0) continue;
        int sp = 0;
        stack[sp++] = s0;
        while (sp > 0) {
            const int curr = stack[sp - 1];
            int valid = 1;
            if (marks[curr] != 2) {
                for (int e = g->in_head[curr]; e >= 0; e = g->e_next_in[e])
                    if (marks[g->e_tail[e]] != 2) { stack[sp++] = g->e_tail[e]; valid = 0; }
                int al[5], na = 0;
                if (!ignored[curr]) {
                    na = aligned_list(g, curr, al);
                    for (int q = 0; q < na; ++q)
                        if (marks[al[q]] != 2) { stack[sp++] = al[q]; ignored[al[q]] = 1; valid = 0; }
                }
                if (valid) {
                    marks[curr] = 2;
                    if (!ignored[curr]) {
                        g->order[w++] = curr;
                        for (int q = 0; q < na; ++q) g->order[w++] = al[q];
                    }
                } else marks[curr] = 1;
            }
            if (valid) --sp;
        }
    }
    if (w != n) { fprintf(stderr, "poa_oracle: spoa_resort emitted %d of %d nodes\n", w, n); abort(); }
    for (int r = 0; r < n; ++r) g->rank[g->order[r]] = r;
    free(marks); free(ignored); free(stack);
}
void poa_graph_spoa_resort(poa_graph_t *g) { spoa_resort(g); }

void poa_graph_nodes(const poa_graph_t *g, uint8_t *code, int32_t *rank, int32_t *group) {
    for (int v = 0; v < g->n_nodes; ++v) {
        if (code) code[v] = g->code[v];
        if (rank) rank[v] = g->rank[v];
        if (group) group[v] = g->leader[v];
    }
}
void poa_graph_edges(const poa_graph_t *g, int32_t *tail, int32_t *head, uint32_t *weight) {
    for (int e = 0; e < g->n_edges; ++e) {
        if (tail) tail[e] = g->e_tail[e];
        if (head) head[e] = g->e_head[e];
        if (weight) weight[e] = g->e_w[e];
    }
}
void poa_graph_row_hints(const poa_graph_t *g, int32_t *hints) {
    for (int r = 0; r < g->n_nodes; ++r) hints[r] = g->xpos[g->order[r]];
}
/* B4: remain() of the node of every rank -- edges of the walk along heaviest out-edges (first of greatest weight in
 * out-list order) down to a node without out-edges.  Reverse rank order: every out-neighbour has a greater rank. */
void poa_graph_row_remain(const poa_graph_t *g, int32_t *remain) {
    for (int r = g->n_nodes - 1; r >= 0; --r) {
        const int v = g->order[r];
        int best = -1;
        uint32_t bw = 0;
        for (int e = g->out_head[v]; e >= 0; e = g->e_next_out[e])
            if (best < 0 || g->e_w[e] > bw) { best = g->e_head[e]; bw = g->e_w[e]; }
        remain[r] = best < 0 ? 0 : remain[g->rank[best]] + 1;
    }
}
int poa_graph_seq_len(const poa_graph_t *g, int s) { return (int)(g->seq_off[s + 1] - g->seq_off[s]); }
void poa_graph_seq_path(const poa_graph_t *g, int s, int32_t *nodes) {
    memcpy(nodes, g->path + g->seq_off[s], sizeof(int32_t) * (size_t)poa_graph_seq_len(g, s));
}

/* ------------------------------------------------------------------------------------ */
/* S8 heaviest bundle (Lee 2003) with branch completion.                                 */

static int branch_completion(const poa_graph_t *g, int r, int64_t *sc, int32_t *pr) {
    const int start = g->order[r];
    for (int e = g->out_head[start]; e >= 0; e = g->e_next_out[e])
        for (int f = g->in_head[g->e_head[e]]; f >= 0; f = g->e_next_in[f])
            if (g->e_tail[f] != start) sc[g->e_tail[f]] = -1;
    int64_t max_score = 0;
    int max_node = -1;
    for (int i = r + 1; i < g->n_nodes; ++i) {
        const int v = g->order[i];
        sc[v] = -1; pr[v] = -1;
        for (int e = g->in_head[v]; e >= 0; e = g->e_next_in[e]) {
            const int t = g->e_tail[e];
            if (sc[t] == -1) continue;
            const int64_t w = g->e_w[e];
            if (sc[v] < w || (sc[v] == w && sc[pr[v]] <= sc[t])) { sc[v] = w; pr[v] = t; }
        }
        if (pr[v] != -1) sc[v] += sc[pr[v]];
        if (max_score < sc[v]) { max_score = sc[v]; max_node = v; }
    }
    return max_node;
}

int poa_consensus(const poa_graph_t *g, int32_t *out_nodes) {
    const int N = g->n_nodes;
    if (N == 0) return 0;
    int64_t *sc = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int32_t *pr = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    for (int v = 0; v < N; ++v) { sc[v] = -1; pr[v] = -1; }
    int mx = -1;
    for (int r = 0; r < N; ++r) {
        const int v = g->order[r];
        for (int e = g->in_head[v]; e >= 0; e = g->e_next_in[e]) {
            const int t = g->e_tail[e];
            const int64_t w = g->e_w[e];
            if (sc[v] < w || (sc[v] == w && sc[pr[v]] <= sc[t])) { sc[v] = w; pr[v] = t; }
        }
        if (pr[v] != -1) sc[v] += sc[pr[v]];
        if (mx == -1 || sc[mx] < sc[v]) mx = v;
    }
    while (g->out_deg[mx] != 0) {
        int nx = branch_completion(g, g->rank[mx], sc, pr);
        if (nx < 0) break;
        mx = nx;
    }
    int n = 0;
    for (int v = mx; v != -1; v = pr[v]) out_nodes[n++] = v;
    for (int a = 0, b = n - 1; a < b; ++a, --b) { int32_t x = out_nodes[a]; out_nodes[a] = out_nodes[b]; out_nodes[b] = x; }
    free(sc); free(pr);
    return n;
}

int poa_msa(const poa_graph_t *g, int with_consensus, char *out) {
    static const char dec[5] = {'A', 'C', 'G', 'T', 'N'};
    const int N = g->n_nodes;
    int32_t *col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N ? N : 1));
    int ncol = 0;
    for (int r = 0; r < N; ++r) {
        const int v = g->order[r];
        if (r > 0 && g->leader[g->order[r - 1]] == g->leader[v]) col[v] = ncol - 1;
        else col[v] = ncol++;
    }
    if (out) {
        const int rows = g->n_seqs + (with_consensus ? 1 : 0);
        memset(out, '-', (size_t)rows * (size_t)ncol);
        for (int s = 0; s < g->n_seqs; ++s)
            for (int64_t k = g->seq_off[s]; k < g->seq_off[s + 1]; ++k)
                out[(size_t)s * ncol + col[g->path[k]]] = dec[g->code[g->path[k]]];
        if (with_consensus) {
            int32_t *cn = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N ? N : 1));
            int n = poa_consensus(g, cn);
            for (int k = 0; k < n; ++k) out[(size_t)g->n_seqs * ncol + col[cn[k]]] = dec[g->code[cn[k]]];
            free(cn);
        }
    }
    free(col);
    return ncol;
}

/* ------------------------------------------------------------------------------------ */

poa_graph_t *poa_block_run(const uint8_t *bases, const int32_t *seq_off, int n_seqs,
                           const uint32_t *weights, const poa_params_t *p,
                           int32_t *scores, uint64_t *cells) {
    return poa_block_run_ws(NULL, bases, seq_off, n_seqs, weights, p, scores, cells);
}
poa_graph_t *poa_block_run_ws(poa_ws_t *ws, const uint8_t *bases, const int32_t *seq_off, int n_seqs,
                              const uint32_t *weights, const poa_params_t *p,
                              int32_t *scores, uint64_t *cells) {
    poa_graph_t *g = poa_graph_new();
    int64_t maxpairs = 16, maxlen = 0;
    for (int s = 0; s < n_seqs; ++s) {
        maxpairs += 2 * (int64_t)(seq_off[s + 1] - seq_off[s]);
        if (seq_off[s + 1] - seq_off[s] > maxlen) maxlen = seq_off[s + 1] - seq_off[s];
    }
    if (ws && ws->impl == POA_IMPL_AVX2 && poa_simd_available()) {
        /* rows a block of similar sequences is expected to reach (same estimate as the GPU engine's arenas) */
        const double grow_f = 1.0 + 0.0165 * n_seqs;
        if (!ws->simd) ws->simd = poa_simd_ws_new();
        poa_simd_ws_reserve(ws->simd, (long)(maxlen * (grow_f > 2.0 ? grow_f : 2.0)) + 1024, (long)maxlen);
    }
    int32_t *an = (int32_t *)malloc(sizeof(int32_t) * (size_t)maxpairs);
    int32_t *ap = (int32_t *)malloc(sizeof(int32_t) * (size_t)maxpairs);
    poa_params_t pb = *p;   /* B2: one strip width for all alignments of the block, from its longest sequence */
    if (pb.banded && ((pb.mode & 1) == POA_MODE_SW || band_is_adaptive(pb.banded)))
        pb.banded = (uint8_t)((band_is_adaptive(pb.banded) ? 0x80 : 0) | poa_band_strip_width((long)maxlen));
    for (int s = 0; s < n_seqs; ++s) {
        const uint8_t *seq = bases + seq_off[s];
        const int len = seq_off[s + 1] - seq_off[s];
        int32_t sc = 0;
        uint64_t cl = 0;
        int n = poa_align_ws(ws, g, seq, len, &pb, an, ap, &sc, &cl);
        if (scores) scores[s] = sc;
        if (cells) cells[s] = cl;
        poa_add_alignment(g, an, ap, n, seq, len, weights ? weights[s] : 1);
        if (pb.mode & POA_ORDER_SPOA) spoa_resort(g);   /* S7' */
    }
    free(an); free(ap);
    return g;
}

int poa_blocks_run_omp(const uint8_t *bases, const int64_t *seq_off, const int32_t *blk_off,
                       int n_blocks, const uint32_t *weights, const poa_params_t *p,
                       int n_threads, int32_t *scores, uint64_t *cells_total,
                       int32_t *n_nodes_out, int32_t *n_edges_out) {
    return poa_blocks_run_omp2(bases, seq_off, blk_off, n_blocks, weights, p, n_threads, POA_IMPL_SCALAR, scores,
                               cells_total, n_nodes_out, n_edges_out);
}
int poa_blocks_run_omp2(const uint8_t *bases, const int64_t *seq_off, const int32_t *blk_off,
                        int n_blocks, const uint32_t *weights, const poa_params_t *p,
                        int n_threads, int impl, int32_t *scores, uint64_t *cells_total,
                        int32_t *n_nodes_out, int32_t *n_edges_out) {
    uint64_t total = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel reduction(+ : total)
#endif
    {
    poa_ws_t *ws = poa_ws_new();   /* one workspace per thread, reused by every block the thread takes */
    poa_ws_set_impl(ws, impl);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int b = 0; b < n_blocks; ++b) {
        const int s0 = blk_off[b], ns = blk_off[b + 1] - s0;
        int32_t *off = (int32_t *)malloc(sizeof(int32_t) * ((size_t)ns + 1));
        for (int s = 0; s <= ns; ++s) off[s] = (int32_t)(seq_off[s0 + s] - seq_off[s0]);
        uint64_t *cl = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(ns ? ns : 1));
        poa_graph_t *g = poa_block_run_ws(ws, bases + seq_off[s0], off, ns, weights ? weights + s0 : NULL, p,
                                          scores ? scores + s0 : NULL, cl);
        for (int s = 0; s < ns; ++s) total += cl[s];
        if (n_nodes_out) n_nodes_out[b] = g->n_nodes;
        if (n_edges_out) n_edges_out[b] = g->n_edges;
        poa_graph_free(g);
        free(off); free(cl);
    }
    poa_ws_free(ws);
    }
    if (cells_total) *cells_total = total;
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* Independent scorer of an alignment (test helper).                                      */

static int gap_cost(const norm_params_t *P, int k) {
    int a = P->g + (k - 1) * P->e, b = P->q + (k - 1) * P->c;
    return a > b ? a : b;
}
static int has_edge(const poa_graph_t *g, int u, int v) {
    for (int e = g->out_head[u]; e >= 0; e = g->e_next_out[e]) if (g->e_head[e] == v) return 1;
    return 0;
}
int32_t poa_rescore(const poa_graph_t *g, const uint8_t *seq, int len, const poa_params_t *pp,
                    const int32_t *an, const int32_t *ap, int n) {
    norm_params_t P = normalise(pp);
    int32_t s = 0;
    int last_node = -1, last_pos = -1, run = 0, run_kind = 0; /* 1 = node gap, 2 = seq gap */
    for (int k = 0; k < n; ++k) {
        const int v = an[k], j = ap[k];
        if (v < 0 && j < 0) return INT32_MIN;
        if (v >= 0) {
            if (v >= g->n_nodes) return INT32_MIN;
            if (last_node >= 0 && !has_edge(g, last_node, v)) return INT32_MIN;
            if (last_node < 0 && !P.sw && g->in_deg[v] != 0) return INT32_MIN;
            last_node = v;
        }
        if (j >= 0) {
            if (j >= len) return INT32_MIN;
            if (last_pos >= 0 ? j != last_pos + 1 : (!P.sw && j != 0)) return INT32_MIN;
            last_pos = j;
        }
        const int kd = (v >= 0 && j >= 0) ? 0 : (v >= 0 ? 1 : 2);
        if (kd != run_kind) { if (run_kind) s += gap_cost(&P, run); run = 0; run_kind = kd; }
        if (kd == 0) s += (g->code[v] == seq[j]) ? P.m : P.n; else ++run;
    }
    if (run_kind) s += gap_cost(&P, run);
    if (!P.sw && n > 0) {
        if (last_pos != len - 1) return INT32_MIN;
        if (last_node >= 0 && g->out_deg[last_node] != 0) return INT32_MIN;
    }
    return s;
}

/* ------------------------------------------------------------------------------------ */
/* XXH64 -- restated from the published xxHash specification (Cyan4973/xxHash,            */
/* doc/xxhash_spec.md); the reference calls it at src/smooth.cpp:716 with seed 0.         */

#define XP1 0x9E3779B185EBCA87ULL
#define XP2 0xC2B2AE3D27D4EB4FULL
#define XP3 0x165667B19E3779F9ULL
#define XP4 0x85EBCA77C2B2AE63ULL
#define XP5 0x27D4EB2F165667C5ULL
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t xround(uint64_t acc, uint64_t in) { return rotl64(acc + in * XP2, 31) * XP1; }
static inline uint64_t xmerge(uint64_t h, uint64_t v) { return (h ^ xround(0, v)) * XP1 + XP4; }

uint64_t poa_xxh64(const void *data, uint64_t len, uint64_t seed) {
    const uint8_t *p = (const uint8_t *)data, *end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        const uint8_t *lim = end - 32;
        do {
            v1 = xround(v1, rd64(p)); v2 = xround(v2, rd64(p + 8));
            v3 = xround(v3, rd64(p + 16)); v4 = xround(v4, rd64(p + 24));
            p += 32;
        } while (p <= lim);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else h = seed + XP5;
    h += len;
    while (p + 8 <= end) { h ^= xround(0, rd64(p)); h = rotl64(h, 27) * XP1 + XP4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * XP1; h = rotl64(h, 23) * XP2 + XP3; p += 4; }
    while (p < end) { h ^= (*p) * XP5; h = rotl64(h, 11) * XP1; ++p; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
}
