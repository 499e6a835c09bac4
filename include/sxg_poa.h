/*
 * sxg_poa.h -- C ABI of the MI355X-native blocked partial-order-alignment engine.
 *
 * Drop-in boundary for the per-block POA of pangenome/smoothxg.  The reference has no
 * plugin/FFI interface for this path; the seam is the block of spoa calls inside
 *   smooth_spoa()  src/smooth.cpp:752-786   (Create / Align / AddAlignment / consensus / MSA)
 * executed once per block by the OpenMP loop of
 *   smooth_and_lace()  src/smooth.cpp:1931-2312.
 * This library replaces that inner sequence for a whole BATCH of blocks per call:
 * everything above it (XG access, padding, dedup: src/smooth.cpp:676-743) and below it
 * (build_odgi_SPOA, unchop, lacing: src/smooth.cpp:2576-2654, 935-1010, src/main.cpp:599+)
 * stays with the caller.  INTEGRATION.md shows the binding a smoothxg maintainer adds.
 *
 * Conventions
 *   - bases are codes 0..4 = A,C,G,T,N (the XG alphabet, src/xg.cpp:24-53); >4 is read as N.
 *   - scoring uses spoa's sign convention exactly as smooth_spoa receives it
 *     (src/smooth.cpp:2098-2106): m > 0, n,g,e,q,c <= 0; mode 0 = local (kSW, default,
 *     src/main.cpp:487), 1 = global (kNW).
 *   - all functions return 0 on success or a negative SXG_E_* code; they never throw,
 *     exit() or abort() (the reference exit(1)s, src/smooth.cpp:943).  The text of the
 *     last error of the calling thread is available from sxg_poa_last_error().
 *   - every *_out struct is library-owned; release it with the matching *_free.
 *   - a handle is bound to one GPU and one HIP stream; one in-flight batch per handle.
 *     Use one handle per host thread / per rank.
 *
 * What "drop-in" does and does not promise: alignment SCORES are those of the published recurrences and are
 * checked against independent implementations; node ids, topological ranks, tie-breaks between equally good
 * alignments, the consensus and the MSA column order follow the tie rules written down in oracle/poa_oracle.c
 * (S1-S8, B1-B4), NOT necessarily spoa's / abPOA's: both are absent from the reference snapshot, so their
 * choices could not be pinned ("PARITY UNPINNED").  One divergence is KNOWN and is a switch: after
 * every AddAlignment spoa re-sorts the whole graph depth-first (node-id order, in-edge tails first, aligned siblings
 * together); with mode | SXG_ORDER_SPOA (what libsxgsmooth.so and bench.py ask for by default since round 6) the engine
 * does the same, as recollected; without it the order is kept incrementally (decree S7: new nodes are slotted next to
 * their aligned group).  Any such order is a legal POA order, but it decides ties between equally good alignments, the
 * end cell of a local alignment and the MSA column order.  Graphs are valid POA graphs of the same sequences either way --
 * every path spells its sequence -- but need not be byte-identical to the reference's.
 */
#ifndef SXG_POA_H
#define SXG_POA_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 5: sxg_poa_batch_out gained block_cycles (before _owner): per-block time on the device, the reference's POA_DEBUG column
 *    poa.time.ms (src/smooth.cpp:2121-2265); sxg_poa_sharded_timing.
 * 4: block graphs on the device: sxg_poa_batch_in gained want_block_graph / bg_consensus_visited_only / bg_trim at its END
 *    (callers must zero-initialise the struct, as before), sxg_poa_batch_out the bg_* arrays before _owner; stats.bg_ms.
 * 3: params.banded = 2 (adaptive band); sxg_poa_batch_run_sharded fails on all ranks together and needs the exchange buffer
 *    that sxg_poa_comm_init / _attach allocate; a band miss is repaired inside the engine at any length.
 * 2 (round 2, never numbered): params.reserved became params.banded -- callers must zero it --, stats.reserved became
 *    dom_clock_mhz, status 7 (SXG_ST_BAND_MISS), SXG_POA_MAX_SEQ_LEN 12287 -> 26623 with SXG_POA_MAX_SEQ_LEN_WIDE. */
#define SXG_POA_ABI_VERSION 5

#define SXG_MODE_LOCAL 0  /* spoa::AlignmentType::kSW */
#define SXG_MODE_GLOBAL 1 /* spoa::AlignmentType::kNW */
/* OR-ed into sxg_poa_params::mode: after every AddAlignment the block's graph is re-sorted depth-first the way spoa's
 * Graph::TopologicalSort is BELIEVED to do it (node-id order, in-edge tails first, aligned nodes together; restated from
 * memory -- the library is absent from the reference snapshot -- as decree S7' of oracle/poa_oracle.c) instead of being kept
 * in order incrementally (decree S7).  Both are valid POA orders; they differ in how ties between equally good alignments
 * fall.  Round 6: every thread of the block's workgroup walks its own roots and only the pieces of the order the last alignment
 * touched are walked again -- 1.01 x the kernel time on 64 x 5 kbp blocks, 1.05 x on 8000 blocks of 16 x 1 kbp; the
 * host library sets the flag by default. */
#define SXG_ORDER_SPOA 0x10

/* return codes */
#define SXG_OK 0
#define SXG_E_INVALID (-1)   /* bad argument */
#define SXG_E_NODEVICE (-2)  /* no usable HIP device / HIP runtime error */
#define SXG_E_NOMEM (-3)     /* device or host allocation failed */
#define SXG_E_BLOCK (-4)     /* at least one block failed; see out->status[] */
#define SXG_NOT_ROOT 1        /* sxg_poa_batch_run_sharded on a rank other than 0: its share is done, the results are on rank 0 */

/* per-block status (out->status[b]) */
#define SXG_ST_OK 0
#define SXG_ST_ROWS_OVERFLOW 1
#define SXG_ST_POOL_OVERFLOW 2
#define SXG_ST_TBX_OVERFLOW 3
#define SXG_ST_NODES_OVERFLOW 4
#define SXG_ST_TOO_LONG 5 /* a sequence exceeds SXG_POA_MAX_SEQ_LEN (local) / SXG_POA_MAX_SEQ_LEN_WIDE (see below) */
#define SXG_ST_RANGE_OVERFLOW 6 /* internal: scores left the narrow sweep's range; the engine re-runs the block wider */
#define SXG_ST_INTERNAL 8       /* an internal invariant failed (e.g. the banded traceback left its band): final, please report */
#define SXG_ST_BAND_MISS 7      /* internal: the packed sweep's traceback left its band of kept cells; re-run wider */

/* Longest sequence.  Local alignment (smoothxg's default) with m * length < 30000 runs the packed int16 sweep, whose
 * largest workgroup covers 2 * 1024 * 13 columns: enough for -l 13k cut at -q 2 * 13k (src/main.cpp:376) plus padding.
 * Global alignment runs the packed sweep with m * length < 14500 whatever its gap scores (round 5: the sweep clamps far below
 * any score a related pair reaches and its traceback checks itself; a block whose walk does meet a clamped cell is re-run on
 * the 32-bit sweep).  Longer global alignments and score sets outside the int16 range (|g| or |q| > 120) need the 32-bit sweep:
 * 1024 lanes * 12 columns. */
#define SXG_POA_MAX_SEQ_LEN 26623
#define SXG_POA_MAX_SEQ_LEN_WIDE 12287

/* The six scores + alignment type that smooth_spoa hands to
 * spoa::AlignmentEngine::Create (src/smooth.cpp:752-755).                                */
typedef struct sxg_poa_params {
    int8_t m, n, g, e, q, c;
    uint8_t mode;
    uint8_t banded; /* 0 = full matrix (the spoa path, smooth_spoa).
                       2 = ADAPTIVE band as the abPOA path runs it (smooth_abpoa, src/smooth.cpp:133-627: wb = 311,
                           wf = 0.03 at :266-271, `true` at :2090): the band of a graph row spans, w = 311 + (int)(0.03 L)
                           columns either side, from the leftmost to the rightmost best-scoring column of its predecessor
                           rows (+1) and the node's position counted back from the end of the graph -- abPOA's published
                           rule, restated as decree B4 of oracle/poa_oracle.c (abPOA itself is absent from the reference
                           snapshot); whole 6-, 8- or 11-column strips, at most 128 strips.
                       1 = STATIC band: w columns either side of the node's backbone coordinate (decrees B1-B3), known
                           before the sweep starts; same kernel, no per-row search of the best cell.
                       The adaptive band serves local AND global alignments (smooth_abpoa sets wb / wf for both modes,
                       src/smooth.cpp:259-271); the static band is for local mode only (a global alignment that asks for
                       it runs the full matrix, and so does a global alignment whose scores leave the packed sweep's
                       int16 range: affine defaults beyond ~3.9 kbp).  out->cells then counts band cells.
                       sxg_poa_align_batch ignores it.                                                            */
} sxg_poa_params;

typedef struct sxg_poa_handle sxg_poa_handle;

/* A batch of blocks.  Block b owns sequences blk_off[b] .. blk_off[b+1]-1, in the order
 * they are to be aligned (smoothxg: longest first, src/blocks.cpp:206-219); sequence s owns
 * bases[seq_off[s] .. seq_off[s+1]).  weights[s] is the dedup multiplicity passed to
 * AddAlignment (src/smooth.cpp:764); NULL = all 1.                                        */
typedef struct sxg_poa_batch_in {
    int32_t n_blocks;
    const int32_t *blk_off;       /* [n_blocks+1] */
    const int64_t *seq_off;       /* [n_seqs+1]   */
    const uint8_t *bases;         /* [seq_off[n_seqs]] */
    const uint32_t *weights;      /* [n_seqs] or NULL */
    const sxg_poa_params *params; /* [n_blocks] if per_block_params else [1] */
    int32_t per_block_params;
    int32_t want_consensus; /* GenerateConsensus(), src/smooth.cpp:773 */
    int32_t want_msa;       /* GenerateMultipleSequenceAlignment(), src/smooth.cpp:785 */
    /* ABI 4.  The NORMALISED BLOCK GRAPH of every block, computed on the device right after the block's chain of
     * alignments: what smooth_spoa returns to its caller (src/smooth.cpp:931-1010) -- build_odgi_SPOA (:2576-2654: one
     * node per POA node, a path per sequence with bg_trim[b] = poa_padding steps trimmed at both ends, consensus path,
     * nodes no path visits dropped), only path-supported edges (:980-994), unchop (:935) and the topological re-numbering
     * (:947), the last two by the decrees of DESIGN.md section 9 (odgi is absent from the reference snapshot).
     *   0 = off (the caller builds block graphs from the raw POA results);
     *   1 = bg_* arrays of the result are filled in addition to the raw results;
     *   2 = ... and seq_path_nodes (one node id per base: two thirds of the download) is left out (NULL);
     *   3 = ... and the per-node / per-edge arrays of the raw POA graphs and cons_nodes as well (NULL; status, n_nodes,
     *       n_edges, the offsets, score and cells stay): for a caller that only laces block graphs (sxg_smooth_gfa).
     *   A request for the MSA (want_msa) turns 2 and 3 into 1: the MSA is formatted from the per-base paths. */
    int32_t want_block_graph;
    int32_t bg_consensus_visited_only; /* 1 = the consensus path keeps only nodes some sequence path visits (build_odgi_abPOA,
                                          src/smooth.cpp:2542-2548); 0 = every consensus node (build_odgi_SPOA, :2624-2627) */
    const int32_t *bg_trim;            /* [n_blocks] poa_padding of every block (src/smooth.cpp:2611); NULL = 0 */
} sxg_poa_batch_in;

/* Per-block POA results, dense and block-major.  Node ids are block-local, 0-based, in
 * creation order (build_odgi_SPOA adds 1, src/smooth.cpp:2587).                            */
typedef struct sxg_poa_batch_out {
    int32_t n_blocks;
    int64_t n_seqs;
    int32_t *status;       /* [n_blocks] SXG_ST_* */
    int64_t *node_off;     /* [n_blocks+1] */
    uint8_t *node_code;    /* letter of every node */
    int32_t *node_rank;    /* topological rank inside the block */
    int32_t *node_group;   /* aligned-group leader (node id); group == MSA column class */
    int64_t *edge_off;     /* [n_blocks+1] */
    int32_t *edge_tail;    /* edges in creation order */
    int32_t *edge_head;
    uint32_t *edge_weight;
    int32_t *seq_path_nodes; /* [total bases] node of every base, indexed like in->bases;
                                replaces sequences()[i] + Successor(i), src/smooth.cpp:2604-2610 */
    int32_t *score;          /* [n_seqs] optimal alignment score of sequence s vs. the graph
                                of its predecessors (0 for the first sequence of a block)    */
    uint64_t *cells;         /* [n_seqs] DP cells = graph nodes x sequence length           */
    int64_t *cons_off;       /* [n_blocks+1] (NULL unless want_consensus) */
    int32_t *cons_nodes;     /* consensus node ids, src/smooth.cpp:2624-2637 */
    int64_t *msa_off;        /* [n_blocks+1] byte offsets into msa (NULL unless want_msa) */
    int32_t *msa_cols;       /* [n_blocks] columns; rows = #seqs (+1 consensus row if wanted) */
    char *msa;               /* row-major 'A','C','G','T','N','-' (GAP_CHAR, src/smooth.cpp:8-11) */
    /* Block graphs (in->want_block_graph; NULL otherwise).  Node ids are block-local, 0-based, in the block graph's final
     * order -- topological: Kahn over the edges, smallest id first (decree of DESIGN.md section 9) --, so every edge runs
     * forward-to-forward from a lower to a higher id and the edges listed per tail, heads ascending, ARE the L lines of
     * the block graph in their order.  A path per (dedup'd) input sequence in the orientation it was aligned in: the caller
     * reverses + flips it for ranges it collected in reverse (src/smooth.cpp:2612-2615) and repeats it for duplicates
     * (:2598-2602).  A failed block has no nodes. */
    int64_t *bg_node_off;     /* [n_blocks+1] */
    int32_t *bg_node_len;     /* [nodes] bases of every node */
    int32_t *bg_node_outdeg;  /* [nodes] out-degree: the edges of the nodes follow one another in bg_edge_to (CSR order) */
    uint8_t *bg_node_indeg;   /* [nodes] in-degree, saturating at 255 */
    int64_t *bg_seq_off;      /* [n_blocks+1] where the block's node sequences start in bg_seq */
    char *bg_seq;             /* node sequences back to back in node order, 'A','C','G','T','N' */
    int64_t *bg_edge_off;     /* [n_blocks+1] */
    int32_t *bg_edge_to;      /* [edges] head of every edge, ascending per tail */
    int64_t *bg_step_off;     /* [n_seqs+1] */
    int32_t *bg_steps;        /* node id of every step of every sequence path */
    int64_t *bg_cons_off;     /* [n_blocks+1] (NULL unless want_consensus) */
    int32_t *bg_cons_steps;   /* steps of the consensus path */
    /* Per-block time on the device (what the reference's -DPOA_DEBUG build tabulates per block as poa.time.ms,
     * src/smooth.cpp:2121-2265, 2319-2350): shader-clock cycles between the moment a slot took the block and the moment it
     * wrote the block's status -- every alignment, traceback and graph update of the block, in its LAST run (a block that
     * was re-run on a larger arena or a wider sweep reports that run).  Milliseconds = cycles / (stats.dom_clock_mhz * 1e3).
     * NULL in the results of a sharded run (the peers' blobs do not carry it). */
    uint64_t *block_cycles;   /* [n_blocks] */
    void *_owner;
} sxg_poa_batch_out;

/* Stand-alone Align(sequence, graph) problems (src/smooth.cpp:761).  Graph p is given in
 * topological order: row r of problem p is global row row_off[p]+r; its predecessors are
 * preds[pred_off[R] .. pred_off[R+1]) as 1-based row numbers LOCAL to the problem (rank+1),
 * in edge-insertion order; an empty list means "source".                                   */
typedef struct sxg_poa_align_in {
    int32_t n;
    const int64_t *row_off;  /* [n+1] */
    const uint8_t *row_code; /* [rows] */
    const uint8_t *row_sink; /* [rows] 1 = node has no out-edge */
    const int64_t *pred_off; /* [rows+1] */
    const int32_t *preds;
    const int64_t *seq_off; /* [n+1] */
    const uint8_t *bases;
    const sxg_poa_params *params;
    int32_t per_problem_params;
} sxg_poa_align_in;

/* Alignment p = pairs pair_off[p]..pair_off[p+1]: (row | -1, sequence position | -1) in
 * forward order -- spoa::Alignment with ranks in place of node ids.                         */
typedef struct sxg_poa_align_out {
    int32_t n;
    int32_t *status;
    int32_t *score;
    int64_t *pair_off;
    int32_t *pair_row;
    int32_t *pair_pos;
    void *_owner;
} sxg_poa_align_out;

/* Timing / accounting of the last execute on a handle. */
typedef struct sxg_poa_stats {
    double kernel_ms;     /* HIP-event time of the POA kernels on the handle's stream */
    uint64_t cells;       /* DP cells evaluated */
    uint64_t dp_launches; /* number of POA kernel launches */
    uint64_t algo_bytes;  /* algorithmic bytes (2*n_cross*sizeof(score)+1 per cell) */
    int32_t n_slots;      /* resident workgroups used */
    int32_t retries;      /* blocks re-run with a larger arena */
    uint64_t device_bytes;/* device memory of the block arenas */
    /* the dominant launch (most cells) of the last execute: what bench.py's roofline quotes */
    double dom_kernel_ms;
    uint64_t dom_cells, dom_algo_bytes;
    int32_t dom_threads, dom_cols_per_lane; /* launch geometry */
    int32_t dom_row_mode;  /* 0/1 = 32-bit sweep (int16 / int32 row words), 2 = packed-int16 sweep, 3 = banded packed sweep */
    int32_t dom_clock_mhz; /* shader clock that launch ran at (cycles of its slots / their 100 MHz wall ticks); 0 = unknown */
    double bg_ms;          /* HIP-event time of the block-graph kernel (want_block_graph), not part of kernel_ms */
} sxg_poa_stats;

int sxg_poa_abi_version(void);
/* Measurement aids (no counterpart in the reference; SURVEY sections 5 and 8d).
 * _measure_copy: a streaming device-to-device copy of `bytes` bytes on this engine's device and stream, timed with HIP events over
 *   `reps` launches: *gbps = (bytes read + bytes written) / time.  The figure bench.py prints beside the data sheet's HBM peak.
 * _roctx_available: 1 when a ROCTx marker library was found -- _upload / _execute / _download and the pack / exchange halves of
 *   _execute_sharded then run inside named ranges (rocprofv3 --marker-trace); 0: the ranges are no-ops. */
int sxg_poa_measure_copy(sxg_poa_handle *h, uint64_t bytes, int reps, double *gbps);
int sxg_poa_roctx_available(void);
int sxg_poa_device_count(void);
const char *sxg_poa_last_error(void);

int sxg_poa_create(int device, sxg_poa_handle **out);
void sxg_poa_destroy(sxg_poa_handle *h);

/* One-shot: upload, run, download. */
int sxg_poa_batch_run(sxg_poa_handle *h, const sxg_poa_batch_in *in, sxg_poa_batch_out *out);
/* Staged form (what bench.py times): inputs become resident in HBM with _upload; _execute
 * runs the whole batch on the device and leaves results in HBM; _download copies them out. */
int sxg_poa_batch_upload(sxg_poa_handle *h, const sxg_poa_batch_in *in);
int sxg_poa_batch_execute(sxg_poa_handle *h);
int sxg_poa_batch_download(sxg_poa_handle *h, sxg_poa_batch_out *out);
void sxg_poa_batch_free(sxg_poa_batch_out *out);

/* Zero-copy view of the executed batch's results in HBM (valid until the next upload on this
 * handle): what a caller hands to RCCL to reassemble the per-block graphs across GPUs before
 * lacing (src/main.cpp:599+) without a host round trip.  Node/edge arrays use the worst-case
 * layout: block b's entries start at index seq_off[blk_off[b]].                              */
typedef struct sxg_poa_device_view {
    int32_t n_blocks;
    int64_t n_seqs, n_bases;
    const int32_t *status, *n_nodes, *n_edges;           /* [n_blocks] */
    const uint8_t *node_code;                            /* [n_bases] worst-case layout */
    const int32_t *node_rank, *node_group;               /* [n_bases] */
    const int32_t *edge_tail, *edge_head;                /* [n_bases] */
    const uint32_t *edge_weight;                         /* [n_bases] */
    const int32_t *seq_path_nodes;                       /* [n_bases] dense */
    const int32_t *score;                                /* [n_seqs] */
} sxg_poa_device_view;
int sxg_poa_batch_device_view(sxg_poa_handle *h, sxg_poa_device_view *out);

/* Multi-GPU, one handle (= one GPU) per rank.  Blocks are independent (src/smooth.cpp:1931): every rank aligns its
 * share and the only exchange is the reassembly of the results on the rank that laces (src/main.cpp:599+), over
 * RCCL/xGMI.  The communicator is either created here from an id that rank 0 generated and handed to the others
 * (any side channel: MPI, a file, torch.distributed), or a ncclComm_t the caller already owns.
 * sxg_poa_batch_run_sharded is sxg_poa_batch_run for that communicator: EVERY rank calls it with the SAME batch;
 * blocks are dealt by cost (longest first onto the least loaded rank), results travel to rank 0 as one blob per
 * peer (exact size, grouped ncclSend/ncclRecv) and rank 0 returns them in the batch's block order.  Other ranks
 * return SXG_NOT_ROOT with an empty result.  Without a communicator it equals sxg_poa_batch_run. */
#define SXG_POA_COMM_ID_BYTES 128
int sxg_poa_comm_unique_id(uint8_t *id);                                   /* ncclGetUniqueId (rank 0) */
int sxg_poa_comm_init(sxg_poa_handle *h, const uint8_t *id, int nranks, int rank);
int sxg_poa_comm_attach(sxg_poa_handle *h, void *nccl_comm, int nranks, int rank); /* caller-owned ncclComm_t */
void sxg_poa_comm_destroy(sxg_poa_handle *h);
int sxg_poa_batch_run_sharded(sxg_poa_handle *h, const sxg_poa_batch_in *in, sxg_poa_batch_out *out);
/* The same in three stages (sxg_poa_batch_run_sharded = upload + execute + download), for callers that keep the inputs
 * resident or cut the shares themselves:
 *   _upload_sharded    every rank, SAME batch: deals the blocks by cost and uploads this rank's share;
 *   _execute_sharded   every rank (collective): aligns the uploaded share -- dealt above, or uploaded by the caller with
 *                      sxg_poa_batch_upload --, packs the results into one device blob, all-gathers the blob sizes and sends
 *                      every blob to rank 0 (grouped ncclSend/ncclRecv).  A failure on one rank is returned by ALL ranks;
 *                      a collective that does not complete within SXG_POA_COMM_TIMEOUT_S seconds (default 600) aborts the
 *                      communicator and returns SXG_E_NODEVICE instead of hanging;
 *   _download_sharded  rank 0: the results of all ranks in the batch's block order (needs the deal of _upload_sharded);
 *                      other ranks: SXG_NOT_ROOT.
 * sxg_poa_sharded_info: ranks whose counts arrived in the last exchange and bytes rank 0 received from its peers. */
int sxg_poa_batch_upload_sharded(sxg_poa_handle *h, const sxg_poa_batch_in *in);
int sxg_poa_batch_execute_sharded(sxg_poa_handle *h);
int sxg_poa_batch_download_sharded(sxg_poa_handle *h, const sxg_poa_batch_in *in, sxg_poa_batch_out *out);
int sxg_poa_sharded_info(sxg_poa_handle *h, int32_t *ranks_seen, uint64_t *bytes_received);
/* Host-clock milliseconds of the last _execute_sharded on this rank: packing its results into one device blob, and the
 * exchange (two size all-gathers + the blobs to rank 0; includes waiting for the slowest rank to finish its share). */
int sxg_poa_sharded_timing(sxg_poa_handle *h, double *pack_ms, double *exchange_ms);
/* Test entry: the same partition / packing / assembly with `nranks` simulated ranks on this one GPU. */
int sxg_poa_batch_run_sharded_local(sxg_poa_handle *h, const sxg_poa_batch_in *in, int nranks, sxg_poa_batch_out *out);

int sxg_poa_align_batch(sxg_poa_handle *h, const sxg_poa_align_in *in, sxg_poa_align_out *out);
void sxg_poa_align_free(sxg_poa_align_out *out);

int sxg_poa_get_stats(sxg_poa_handle *h, sxg_poa_stats *out);
/* Cap on device memory the handle may use for scratch arenas (bytes; 0 = default 3/4 of free). */
int sxg_poa_set_memory_budget(sxg_poa_handle *h, uint64_t bytes);

/* XXH64 of a sequence: the dedup key smooth_spoa uses (src/smooth.cpp:716, seed 0). */
uint64_t sxg_xxh64(const void *data, uint64_t len, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
