// integration_snippet2.cpp -- TEST HARNESS: INTEGRATION.md's "ready-made phases" and multi-GPU snippets, compiled and
// linked as a caller would (g++ -std=c++17, include/sxg_smooth.h, -lsxgsmooth -lsxgpoa).  tests/test_smooth_host.py
// checks that the blocks between the markers are, line for line, the code shown in INTEGRATION.md.  On a GPU-less box
// the program stops at sxg_poa_create (no device, no CPU fallback) after having exercised the host-only calls.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include "sxg_smooth.h"

static int smooth_once(sxg_poa_handle* engine, const char* text, size_t len, uint64_t target, uint64_t n_haps, char** gfa_out) {
// ---- INTEGRATION.md snippet begin
sxg_graph* g;     sxg_graph_from_gfa(text, len, &g);                       // S, P, L lines
sxg_blockset *b0, *b;
sxg_blockset_smoothable(g, /*-w*/ target * n_haps, /*-l*/ target, /*-j*/ 100, /*-e*/ 0, /*longest first*/ 1, &b0);
sxg_blockset_break(g, b0, /*-q*/ 2 * target, 1, &b);                       // src/main.cpp:447-470
// ...or hand over blocks found elsewhere: sxg_blockset_from_ranges(g, n_blocks, blk_off, ranges, &b)
sxg_smooth_params p;  sxg_smooth_default_params(&p);
char* gfa;
int rc = sxg_smooth_gfa(g, b, &p, (sxg_poa_run_fn)sxg_poa_batch_run, (sxg_poa_free_fn)sxg_poa_batch_free, engine, &gfa);
// with -M / MAF output: sxg_smooth_maf_gfa(g, b, &p, &merge_params, run, free, engine, &gfa, &maf, &n_flipped)
// ---- INTEGRATION.md snippet end
    *gfa_out = rc == SXG_OK ? gfa : nullptr;
    const long long nblk = (long long)sxg_blockset_size(b);
    sxg_blockset_free(b0); sxg_blockset_free(b); sxg_graph_free(g);
    std::printf("blocks %lld rc %d\n", nblk, rc);
    return rc;
}

static int sharded_run(sxg_poa_handle* h, int rank, int nranks, const sxg_poa_batch_in& in, sxg_poa_batch_out& out) {
// ---- INTEGRATION.md snippet begin
uint8_t id[SXG_POA_COMM_ID_BYTES];
if (rank == 0) sxg_poa_comm_unique_id(id);          // ncclGetUniqueId
/* broadcast id to the other ranks by whatever launched them (MPI_Bcast, a file, a TCP store) */
sxg_poa_comm_init(h, id, nranks, rank);             // or sxg_poa_comm_attach(h, existing_ncclComm, nranks, rank)
int rc = sxg_poa_batch_run_sharded(h, &in, &out);   // LPT share per rank; rank 0 receives every block's results
if (rc == SXG_NOT_ROOT) { /* this rank's share is done; rank 0 runs phase 3 and the lacing */ }
// ---- INTEGRATION.md snippet end
    return rc;
}

int main() {
    const std::string text = "H\tVN:Z:1.0\nS\t1\tACGTACGTAC\nS\t2\tGGATTACA\nS\t3\tTTGACCA\nL\t1\t+\t2\t+\t0M\nL\t2\t+\t3\t+\t0M\nL\t1\t+\t3\t+\t0M\n"
                             "P\ta\t1+,2+,3+\t*\nP\tb\t1+,3+\t*\n";
    // host-only part first: discovery works without a device
    sxg_graph* g = nullptr;
    if (sxg_graph_from_gfa(text.data(), text.size(), &g) != SXG_OK) return 2;
    sxg_blockset* b0 = nullptr;
    if (sxg_blockset_smoothable(g, 1000, 500, 100, 0, 1, &b0) != SXG_OK) return 3;
    std::printf("discovery: %lld blocks\n", (long long)sxg_blockset_size(b0));
    sxg_blockset_free(b0); sxg_graph_free(g);
    sxg_poa_handle* h = nullptr;
    const int rc = sxg_poa_create(0, &h);
    if (rc != SXG_OK) { std::printf("no device (%d): %s\n", rc, sxg_poa_last_error()); return rc == SXG_E_NODEVICE ? 0 : 1; }
    char* gfa = nullptr;
    const int r2 = smooth_once(h, text.data(), text.size(), 500, 2, &gfa);
    if (gfa) { std::printf("gfa %zu bytes\n", std::strlen(gfa)); sxg_smooth_free(gfa); }
    if (false) { sxg_poa_batch_in in; sxg_poa_batch_out out; std::memset(&in, 0, sizeof in); std::memset(&out, 0, sizeof out); (void)sharded_run(h, 0, 1, in, out); }
    sxg_poa_destroy(h);
    return r2 == SXG_OK ? 0 : 1;
}
