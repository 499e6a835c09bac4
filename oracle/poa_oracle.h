/*
 * poa_oracle.h -- CPU ORACLE for the blocked-POA hot path of pangenome/smoothxg.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (smoothxg_amd/csrc, the
 * C-ABI in include/sxg_poa.h) never links, calls or falls back to anything here.
 *
 * PARITY UNPINNED.  The arithmetic of this path lives in third-party libraries that are
 * ABSENT from /root/reference (deps/spoa, deps/abPOA, deps/odgi, deps/xxHash are empty
 * submodule directories; .gitmodules:31-57 carries URLs only, the pinned commits were
 * lost with .git).  The reference holds no golden vector, known-answer test or fixture
 * for this path (CMakeLists.txt:562-567 checks an exit status only).  This file therefore
 * restates the PUBLISHED algorithms (Lee, Grasso & Sharlow 2002 partial order alignment;
 * Gotoh affine gaps; two-piece affine "convex" gaps; Lee 2003 heaviest bundle) and
 * anchors on the reference's own call sites:
 *   spoa::AlignmentEngine::Create(kSW|kNW, m,n,g,e,q,c)      src/smooth.cpp:752-755
 *   engine->Align(seq, graph)                                 src/smooth.cpp:761
 *   graph.AddAlignment(alignment, seq, weight)                src/smooth.cpp:764
 *   graph.GenerateConsensus()                                 src/smooth.cpp:773
 *   graph.GenerateMultipleSequenceAlignment(add_consensus)    src/smooth.cpp:785-786
 *   XXH64(seq, len, 0) dedup key                              src/smooth.cpp:716
 * Every tie-break is fixed BY DECREE below (DESIGN.md "Semantics"); the only piece
 * pinned against an independent implementation is XXH64 (python `xxhash` wheel and the
 * published empty-input vector 0xEF46DB3751D8E999).
 */
#ifndef POA_ORACLE_H
#define POA_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POA_MODE_SW 0 /* local; smoothxg default (src/main.cpp:487) */
#define POA_MODE_NW 1 /* global; smoothxg -Z */
#define POA_ORDER_SPOA 0x10 /* OR-ed into mode: after every AddAlignment the graph is re-sorted depth-first the way spoa is
                               BELIEVED to do it (decree S7', poa_oracle.c) instead of being kept in order incrementally (S7) */

typedef struct {
    int8_t m, n, g, e, q, c; /* spoa sign convention: m>0 match, n<=0 mismatch, gaps <=0
                                (src/smooth.cpp:2098-2106 negates the CLI values)       */
    uint8_t mode;            /* POA_MODE_SW | POA_MODE_NW                               */
    uint8_t banded;          /* != 0: abPOA-style band wb=311, wf=0.03 (src/smooth.cpp:266-271), decrees B1-B4 in
                                poa_oracle.c; local mode only (ignored for POA_MODE_NW).  1 = band around the backbone
                                coordinate (B2), 2 = ADAPTIVE band (B4: abPOA's rule), strip width from the length of the
                                sequence being aligned; 6, 8 or 11 = B2 with that strip width, 0x80 | 6, 8, 11 = B4 with that
                                strip width (what poa_block_run sets for every alignment of a block: from the block's longest) */
} poa_params_t;

#define POA_BAND_WB 311
#define POA_BAND_WF 0.03
#define POA_BAND_WMAX 693 /* decree B1: the half-width is capped so that a band never exceeds 128 strips (L > 12 733) */
/* decree B2: the band is a whole number of strips; the strip is the narrowest of 6, 8, 11 columns with which a band of
 * the block's longest sequence spans at most 128 strips (2w/strip + 2) -- one wavefront's window in the device kernel */
static inline int poa_band_strip_width(long maxlen) {
    const long w0 = POA_BAND_WB + (long)(POA_BAND_WF * (double)maxlen), w = w0 < POA_BAND_WMAX ? w0 : POA_BAND_WMAX;
    return 2 * w <= 756 ? 6 : (2 * w <= 1008 ? 8 : 11);
}

typedef struct poa_graph poa_graph_t;

/* Implementations of Align (same results): the scalar recorded-choice DP of poa_oracle.c, or the AVX2
 * int16 row sweep with value-derived traceback of poa_simd.c (falls back to scalar when the host has no
 * AVX2 or the scores leave int16).  A workspace keeps every buffer an alignment needs; one per thread.  */
#define POA_IMPL_SCALAR 0
#define POA_IMPL_AVX2 1
typedef struct poa_ws poa_ws_t;
typedef struct poa_simd_ws poa_simd_ws_t;
poa_ws_t *poa_ws_new(void);
void poa_ws_free(poa_ws_t *w);
void poa_ws_set_impl(poa_ws_t *w, int impl);
poa_simd_ws_t *poa_simd_ws_new(void);
void poa_simd_ws_free(poa_simd_ws_t *w);
void poa_simd_ws_reserve(poa_simd_ws_t *w, long rows, long len);
/* returns the number of pairs, or -1 when not applicable (caller uses the scalar path) */
int poa_align_rows_simd(poa_simd_ws_t *w, int n_rows, const uint8_t *codes, const int32_t *off, const int32_t *pred,
                        const uint8_t *sink, const int32_t *row_node, const uint8_t *seq, int len,
                        const poa_params_t *p, int32_t *out_node, int32_t *out_pos, int32_t *score);
int poa_simd_available(void);

poa_graph_t *poa_graph_new(void);
void poa_graph_free(poa_graph_t *g);
int poa_graph_num_nodes(const poa_graph_t *g);
int poa_graph_num_edges(const poa_graph_t *g);
int poa_graph_num_seqs(const poa_graph_t *g);

/* Align seq (codes 0..4 = A,C,G,T,N) to g.  out_node/out_pos need room for
 * (num_nodes + len) pairs.  A pair is (node id | -1, seq pos | -1), forward order.
 * Returns number of pairs; *score = optimal score; *cells = num_nodes * len.           */
int poa_align(const poa_graph_t *g, const uint8_t *seq, int len, const poa_params_t *p,
              int32_t *out_node, int32_t *out_pos, int32_t *score, uint64_t *cells);

int poa_align_ws(poa_ws_t *ws, const poa_graph_t *g, const uint8_t *seq, int len, const poa_params_t *p,
                 int32_t *out_node, int32_t *out_pos, int32_t *score, uint64_t *cells);

/* Same DP over a caller-supplied CSR in rank space (rows as poa_graph_rows() lays them
 * out); reported node ids are ranks.  Checks the HIP align-only entry point.            */
int poa_align_csr(int n_rows, const uint8_t *codes, const int32_t *off, const int32_t *pred,
                  const uint8_t *sink, const uint8_t *seq, int len, const poa_params_t *p,
                  int32_t *out_node, int32_t *out_pos, int32_t *score);

/* Same contract as poa_align_csr, second restatement (poa_vtb.c): nothing is recorded during the fill,
 * the alignment is derived from the stored values H / oF / oO by re-applying the tie rules S3.        */
int poa_align_csr_vtb(int n_rows, const uint8_t *codes, const int32_t *off, const int32_t *pred,
                      const uint8_t *sink, const uint8_t *seq, int len, const poa_params_t *p,
                      int32_t *out_node, int32_t *out_pos, int32_t *score);

/* Fuse an alignment into the graph (spoa Graph::AddAlignment semantics, see .c).       */
void poa_graph_spoa_resort(poa_graph_t *g); /* S7': spoa's depth-first re-sort (as recollected, unverified) */
void poa_add_alignment(poa_graph_t *g, const int32_t *aln_node, const int32_t *aln_pos,
                       int n_pairs, const uint8_t *seq, int len, uint32_t weight);

/* Flat views (caller-provided buffers). */
void poa_graph_nodes(const poa_graph_t *g, uint8_t *code, int32_t *rank, int32_t *group);
void poa_graph_edges(const poa_graph_t *g, int32_t *tail, int32_t *head, uint32_t *weight);
/* CSR in RANK space for a stand-alone alignment call: row r (0-based rank) has codes[r],
 * preds pred[off[r]..off[r+1]) given as ROW indices (rank+1; 0 = virtual source row),
 * sink[r] = 1 if the node has no out-edge.  pred needs room for num_edges + num_nodes. */
void poa_graph_rows(const poa_graph_t *g, uint8_t *codes, int32_t *off, int32_t *pred,
                    uint8_t *sink, int32_t *row_node);
/* backbone coordinate of the node at every rank (decree B2 / the HIP sweep's band hints) */
void poa_graph_row_hints(const poa_graph_t *g, int32_t *hints);
void poa_graph_row_remain(const poa_graph_t *g, int32_t *remain);
int poa_graph_seq_len(const poa_graph_t *g, int s);
void poa_graph_seq_path(const poa_graph_t *g, int s, int32_t *nodes);

/* Heaviest-bundle consensus; out needs num_nodes entries; returns length. */
int poa_consensus(const poa_graph_t *g, int32_t *out_nodes);
/* MSA: returns number of columns; rows (num_seqs [+1 consensus]) x cols bytes of
 * 'A','C','G','T','N','-' written row-major into out (may be NULL to query size).      */
int poa_msa(const poa_graph_t *g, int with_consensus, char *out);

/* Whole block: sequentially Align + AddAlignment over seqs (src/smooth.cpp:760-769).
 * scores[n_seqs], cells[n_seqs] optional.  Returns the graph (caller frees).           */
poa_graph_t *poa_block_run(const uint8_t *bases, const int32_t *seq_off, int n_seqs,
                           const uint32_t *weights, const poa_params_t *p,
                           int32_t *scores, uint64_t *cells);

poa_graph_t *poa_block_run_ws(poa_ws_t *ws, const uint8_t *bases, const int32_t *seq_off, int n_seqs,
                              const uint32_t *weights, const poa_params_t *p, int32_t *scores, uint64_t *cells);

/* Many blocks, OpenMP `parallel for schedule(dynamic,1)` as src/smooth.cpp:1931.
 * Only aggregate outputs (for the CPU baseline): total cells and per-block node/edge
 * counts + checksum of (scores).  Returns 0.                                            */
int poa_blocks_run_omp(const uint8_t *bases, const int64_t *seq_off, const int32_t *blk_off,
                       int n_blocks, const uint32_t *weights, const poa_params_t *p,
                       int n_threads, int32_t *scores, uint64_t *cells_total,
                       int32_t *n_nodes_out, int32_t *n_edges_out);

/* ... with a chosen implementation (POA_IMPL_*) and one workspace per thread */
int poa_blocks_run_omp2(const uint8_t *bases, const int64_t *seq_off, const int32_t *blk_off,
                        int n_blocks, const uint32_t *weights, const poa_params_t *p,
                        int n_threads, int impl, int32_t *scores, uint64_t *cells_total,
                        int32_t *n_nodes_out, int32_t *n_edges_out);

/* XXH64 (Cyan4973/xxHash, published algorithm), seed as given. */
uint64_t poa_xxh64(const void *data, uint64_t len, uint64_t seed);

/* Score an alignment independently of the DP (test helper): walks the pairs and charges
 * match/mismatch and two-piece gap costs.  Returns INT32_MIN if the pairs do not describe
 * a valid path through g for seq.                                                        */
int32_t poa_rescore(const poa_graph_t *g, const uint8_t *seq, int len, const poa_params_t *p,
                    const int32_t *aln_node, const int32_t *aln_pos, int n_pairs);

#ifdef __cplusplus
}
#endif
#endif
