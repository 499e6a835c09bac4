/*
 * sxg_smooth.h -- C ABI of the host-side rows around the blocked-POA engine:
 *
 *   A2  append_to_sequence (flank padding)            src/smooth.cpp:75-126
 *   A3  sequence collection, orientation, XXH64 dedup src/smooth.cpp:676-743
 *   A4  padding size                                  src/smooth.cpp:1946-1970
 *   A14 adaptive POA scores per block (-a)            src/smooth.cpp:1972-2069
 *   A8' MSA -> MAF rows of a block (-m)               src/smooth.cpp:782-905, src/maf.hpp:35-66
 *   A9  build_odgi_SPOA (POA graph -> block graph)    src/smooth.cpp:2576-2654
 *   A10 unchop + topological order + re-copy          src/smooth.cpp:935-1010
 *   8f-1 lacing of the block graphs + GFA writer      src/main.cpp:599-1061
 *   8f-3 minimal GFA reader (S/P lines)               src/xg.cpp:696-741 (what the path needs of it)
 *
 * Built by g++ into libsxgsmooth.so (no HIP inside): the POA itself is reached through a callback
 * with the signature of sxg_poa_batch_run (include/sxg_poa.h), so production passes the GPU engine
 * and nothing in this library can fall back to a CPU aligner.
 *
 * odgi (unchop, topological_order, to_gfa) is absent from the reference snapshot, so those three
 * are restated by decree: parity with real odgi bytes is unpinned.
 * All strings returned through char** are malloc'ed; release with sxg_smooth_free().
 */
#ifndef SXG_SMOOTH_H
#define SXG_SMOOTH_H
#include <stddef.h>
#include <stdint.h>
#include "sxg_poa.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sxg_graph sxg_graph;       /* input graph: node sequences + paths (XG's role) */
typedef struct sxg_blockset sxg_blockset; /* blockset_t: blocks of path ranges, src/blocks.hpp:29-120 */

/* POA provider: exactly sxg_poa_batch_run's contract, ctx is its handle.  Both callbacks are only called while the
 * sxg_* entry point they were passed to is running -- `run` possibly from a worker thread of that call (one call at a
 * time), `free` on the calling thread, once per successful `run`, before the entry point returns: callbacks and ctx
 * need not outlive the call. */
typedef int (*sxg_poa_run_fn)(void *ctx, const sxg_poa_batch_in *in, sxg_poa_batch_out *out);
typedef void (*sxg_poa_free_fn)(sxg_poa_batch_out *out);

/* smoothxg's knobs on this path with their defaults (src/main.cpp:291-361,487). */
/* 2: sxg_smooth_params starts with struct_size (set by sxg_smooth_default_params, checked by every entry point: a caller
 *    built against another header gets SXG_E_INVALID instead of fields read from whatever follows its struct) and ends
 *    with poa_spoa_order; scores whose engine form leaves int8 are rejected. */
#define SXG_SMOOTH_ABI_VERSION 2
int sxg_smooth_abi_version(void);

typedef struct sxg_smooth_params {
    uint32_t struct_size;                             /* sizeof(sxg_smooth_params) of the caller's header */
    int32_t poa_m, poa_n, poa_g, poa_e, poa_q, poa_c; /* CLI convention: positive penalties (1,4,6,2,26,1) */
    int32_t local_alignment;                          /* 1 = default; 0 = -Z */
    float poa_padding_fraction;                       /* -O, default 0.001 */
    uint64_t max_block_depth_for_padding_more;        /* -Y, default 1000 */
    int32_t add_consensus;                            /* embed Consensus_<block> paths (last iteration) */
    const char *consensus_base_name;                  /* default "Consensus_" */
    int32_t adaptive_poa_params;                      /* -a: per-block scores from the estimated identity, default 0 */
    int32_t kmer_size;                                /* -k: k-mer size of the identity estimate, default 17 */
    int32_t use_abpoa;                                /* -A (src/main.cpp:130-132): the smooth_abpoa path, src/smooth.cpp:133-627.  The
                                                         scores are then in abPOA's convention -- a gap of k costs min(g + k e, q + k c);
                                                         g = 0: linear, q = 0: affine (the 4-parameter form sets q = c = 0,
                                                         src/main.cpp:355-358) --, alignments run with the ADAPTIVE band
                                                         (params.banded = 2: wb = 311, wf = 0.03, src/smooth.cpp:266-271,2090) and the
                                                         consensus path keeps only nodes some sequence visits (build_odgi_abPOA,
                                                         src/smooth.cpp:2542-2548).  Default 0. */
    int32_t abpoa_band_local;                         /* use_abpoa with LOCAL alignment (the default mode): 1 = banded as the call
                                                         site asks (src/smooth.cpp:266-271 sets wb / wf for both modes; default);
                                                         0 = full matrix -- upstream abPOA is believed to switch its adaptive band
                                                         off in local mode (abpoa_post_set_para; unverifiable here, the library is
                                                         absent from the snapshot), in which case this is what -A without -Z runs.
                                                         Global alignment (-Z) is banded either way. */
    int32_t poa_spoa_order;                           /* 1 = the engine re-sorts a block's graph after every alignment the way
                                                         spoa's TopologicalSort is believed to (SXG_ORDER_SPOA, include/sxg_poa.h:
                                                         decree S7', restated from memory, unverified) instead of keeping it in
                                                         order incrementally (decree S7).  DEFAULT 1 since round 6: it is what
                                                         graph.AddAlignment does at src/smooth.cpp:764, and the re-sort is a
                                                         parallel, incremental device phase now (1-5 % of the kernel time at full batches,
                                                         DESIGN.md).  Ignored with use_abpoa (abPOA does not call spoa's sort).
                                                         0 = the order of rounds 1-5. */
} sxg_smooth_params;

void sxg_smooth_default_params(sxg_smooth_params *p);
const char *sxg_smooth_last_error(void);
void sxg_smooth_free(void *p);

/* GFA reader: S, P and L lines (the L lines only serve block discovery's edge-jump limit).
 * Letters outside ACGT become N as in XG's 3-bit alphabet (src/xg.cpp:24-53). */
int sxg_graph_from_gfa(const char *text, size_t len, sxg_graph **out);
void sxg_graph_free(sxg_graph *g);
int64_t sxg_graph_node_count(const sxg_graph *g);
int64_t sxg_graph_path_count(const sxg_graph *g);

/* Test/demo blockset: every path is cut into consecutive windows of >= target_bp and the k-th window
 * of every path forms block k.  This is NOT the reference's smoothable_blocks()/break_blocks() (block
 * discovery is out of scope): it only provides a valid partition of all path
 * steps -- roughly homologous for collinear haplotypes -- so that collection, POA and lacing can be
 * exercised end to end.  Ranges of a block are ordered longest first (src/blocks.cpp:206-219). */
int sxg_blockset_by_path_windows(const sxg_graph *g, uint64_t target_bp, sxg_blockset **out);

/* path_range_t of the reference (src/blocks.hpp:29-33): steps [step_begin, step_end) of path `path`
 * (rank of the P line in the input GFA), `length` = their bases (0 = let the library compute it; a
 * non-zero value is checked against the steps). */
typedef struct sxg_path_range {
    int64_t path, step_begin, step_end, length;
} sxg_path_range;
/* blockset_t from the caller's own blocks (src/blocks.hpp:70-120, what smoothable_blocks / break_blocks
 * produce): block k owns ranges[blk_off[k] .. blk_off[k+1]) in alignment order. */
int sxg_blockset_from_ranges(const sxg_graph *g, int64_t n_blocks, const int64_t *blk_off, const sxg_path_range *ranges,
                             sxg_blockset **out);
/* Block discovery, src/blocks.cpp:7-327 (smoothable_blocks): the greedy sweep over the nodes in rank order
 * with the weight / path-length / path-jump / edge-jump limits, ranges broken at steps an earlier block took,
 * blocks split into the connected components of their path adjacencies.  smoothxg passes
 * max_block_weight = target_poa_length * n_haps (-w), max_block_path_length = target_poa_length (-l),
 * max_path_jump (-j, default 100), max_edge_jump (-e, default 0 = off), src/main.cpp:282-283,376-377,447-455.
 * Needs the L lines of the GFA (edge jumps).  Decrees: stable ordering of equal-length ranges, id order = XG
 * node order. */
int sxg_blockset_smoothable(const sxg_graph *g, uint64_t max_block_weight, uint64_t max_block_path_length,
                            uint64_t max_path_jump, uint64_t max_edge_jump, int order_paths_from_longest,
                            sxg_blockset **out);
/* The cutting half of break_blocks, src/breaks.cpp:210-330: a block that holds a range longer than max_poa_length (-q,
 * default 2 * target) is cut.  As the reference always does (break_repeats = true, src/main.cpp:476), the ranges of such a
 * block of at least 2 * min_copy_length bases are searched for a tandem repeat first (src/breaks.cpp:224-272); if any is
 * found, EVERY range of the block is cut at half the mean repeat length, otherwise the long ranges are cut blindly at
 * max_poa_length.  sxg_blockset_break uses the reference's repeat parameters (min_copy_length 1000, max_copy_length 20000,
 * min_autocorr_z 5, autocorr_stride 50: src/main.cpp:285-286,457-458); sxg_blockset_break_ex takes them (break_repeats = 0:
 * blind cuts only).  The repeat detector stands in for sautocorr::repeat, an un-vendored dependency absent from the
 * snapshot, BY DECREE: match-fraction autocorrelation at every lag in [min_copy_length,
 * max_copy_length] over positions sampled every autocorr_stride bases, z-score across the lags, first lag of greatest z
 * if that z reaches min_autocorr_z.  Identity splitting (src/breaks.cpp:335+; off by default) is not applied. */
int sxg_blockset_break(const sxg_graph *g, const sxg_blockset *in, uint64_t max_poa_length, int order_paths_from_longest,
                       sxg_blockset **out);
int sxg_blockset_break_ex(const sxg_graph *g, const sxg_blockset *in, uint64_t max_poa_length, int break_repeats,
                          uint64_t min_copy_length, uint64_t max_copy_length, double min_autocorr_z, uint64_t autocorr_stride,
                          int order_paths_from_longest, sxg_blockset **out);
int64_t sxg_blockset_block_size(const sxg_blockset *b, int64_t block_id);              /* ranges of a block, -1 on error */
int sxg_blockset_block_ranges(const sxg_blockset *b, int64_t block_id, sxg_path_range *out); /* out[block size] */
void sxg_blockset_free(sxg_blockset *b);
int64_t sxg_blockset_size(const sxg_blockset *b);

/* A2-A4 for one block, as text (one line per record) for inspection and tests:
 *   "padding\t<poa_padding>"
 *   "seq\t<rank>\t<weight>\t<sequence>"                     dedup'd, alignment order
 *   "dup\t<rank>\t<path_range index>\t<is_rev 0|1>\t<name>"  every original range            */
int sxg_block_collect_text(const sxg_graph *g, const sxg_blockset *b, int64_t block_id,
                           const sxg_smooth_params *p, char **out_text);

/* A14, the score tiers of src/smooth.cpp:2032-2069: CLI-convention scores (m,n,g,e,q,c) for a block
 * whose identity threshold is `est_identity_threshold`; below 0.90 the set/default scores are kept. */
void sxg_adaptive_poa_scores(float est_identity_threshold, const int32_t set_scores[6], int32_t out_scores[6]);

/* A14, the identity threshold of one block (src/smooth.cpp:1980-2030): the block's path-range
 * sequences of at least 8*k bases, all-vs-all identity = 1 - mash distance, the 30 % percentile of
 * the sorted identities, floored at 0.7.  *n_used = sequences that took part; the threshold is only
 * defined (and only applied) when *n_used > 1.
 * The reference hashes and compares with rkmh/mkmh, an un-vendored dependency absent from the
 * snapshot; by decree the Jaccard index is taken EXACTLY over the sets of
 * canonical k-mers (no sketch), distance = -ln(2J/(1+J))/k, 1 when J = 0.  k <= 32. */
int sxg_block_identity_threshold(const sxg_graph *g, const sxg_blockset *b, int64_t block_id, int32_t kmer_size,
                                 float *est_identity_threshold, int32_t *n_used);

/* A9+A10 for one block given its POA result: the normalised block graph as GFA text. */
int sxg_block_graph_gfa(const sxg_graph *g, const sxg_blockset *b, int64_t block_id,
                        const sxg_smooth_params *p, sxg_poa_run_fn run, sxg_poa_free_fn fre, void *ctx,
                        char **out_gfa);

/* MAF rows of one block (src/smooth.cpp:782-905): the block's MSA (consensus row last when
 * add_consensus) with the padding blanked (first/last poa_padding non-gap characters of every row),
 * all-gap flank columns trimmed, and one record per (sequence, duplicate) in the reference's emission
 * order, as text lines  "<src>\t<start>\t<size>\t<+|->\t<srcSize>\t<text>".  For a reverse range
 * start counts from the end of the path (MAF convention, :873-876). */
int sxg_block_maf_rows(const sxg_graph *g, const sxg_blockset *b, int64_t block_id, const sxg_smooth_params *p,
                       sxg_poa_run_fn run, sxg_poa_free_fn fre, void *ctx, char **out_rows);

/* The same rows as one MAF block, formatted as write_maf_rows does (src/maf.hpp:35-66: "s " lines,
 * columns padded to the widest entry, a blank line at the end).  The reference walks a hash map there,
 * so its order of sources is unspecified; here sources appear in first-emission order. */
int sxg_block_maf(const sxg_graph *g, const sxg_blockset *b, int64_t block_id, const sxg_smooth_params *p,
                  sxg_poa_run_fn run, sxg_poa_free_fn fre, void *ctx, char **out_maf);

/* One smoothing iteration over all blocks: collect -> ONE batched POA call -> block graphs ->
 * lace -> validate (every path spells its original sequence, src/main.cpp:770-810) -> unchop ->
 * GFA text.  Returns SXG_OK, or SXG_E_INVALID if validation fails. */
int sxg_smooth_gfa(const sxg_graph *g, const sxg_blockset *b, const sxg_smooth_params *p,
                   sxg_poa_run_fn run, sxg_poa_free_fn fre, void *ctx, char **out_gfa);

/* A13 + 8f-4: the iteration with the in-order MAF consumer of smooth_and_lace (src/smooth.cpp:1600-1919): every
 * block's MAF rows are merged into groups of blocks whose path ranges continue one another (contiguous-path Jaccard
 * >= the threshold, -M / -J), a block that joins a group in the opposite orientation is FLIPPED -- its block graph is
 * rebuilt reverse-complemented (src/smooth.cpp:2352-2436), which changes the laced GFA --, merged groups get one
 * merged consensus path (src/main.cpp:870-960) and the MAF text is written group by group (src/smooth.cpp:1312-1544).
 * Decrees: hash-map walks of the reference become insertion order; the flip applies the intended transformation
 * (see sxg_smooth.cpp: the reference indexes its handle table with keys it never inserted). */
typedef struct sxg_merge_params {
    int32_t merge_blocks;                 /* -M, default 0 */
    double contiguous_path_jaccard;       /* -J, default 1.0 */
    int32_t preserve_unmerged_consensus;  /* -N, default 0 */
    uint64_t max_merged_groups_in_memory; /* default 50 */
    const char *maf_header;               /* written before the first block (src/main.cpp:489-520); NULL = none */
} sxg_merge_params;
void sxg_merge_default_params(sxg_merge_params *mp);
int sxg_smooth_maf_gfa(const sxg_graph *g, const sxg_blockset *b, const sxg_smooth_params *p, const sxg_merge_params *mp,
                       sxg_poa_run_fn run, sxg_poa_free_fn fre, void *ctx, char **out_gfa, char **out_maf, int64_t *n_flipped);

#ifdef __cplusplus
}
#endif
#endif
