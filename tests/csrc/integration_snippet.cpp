// integration_snippet.cpp -- TEST HARNESS: INTEGRATION.md's phase-2 binding, compiled and linked as a
// smoothxg maintainer would (g++ -std=c++17, include/sxg_poa.h, -lsxgpoa).  tests/test_smooth_host.py checks
// that the block between the markers is, line for line, the code shown in INTEGRATION.md.
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>
// ---- INTEGRATION.md snippet begin
#include "sxg_poa.h"

// A,C,G,T,N -> 0..4 (the XG alphabet, src/xg.cpp:24-53)
static inline uint8_t sxg_code(char c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1;
                 case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

struct pending_block_t {                       // what phase 1 keeps per block
    std::vector<std::string> seqs;             // src/smooth.cpp:651  (dedup'd, alignment order)
    std::vector<uint32_t>    weights;          // src/smooth.cpp:652
    /* dup_is_revs, dup_seq_names, dup_rank_in_path_ranges, all_names_in_original_order … */
};

int run_poa_on_gpu(sxg_poa_handle* h, const std::vector<pending_block_t>& blocks,
                   int8_t m, int8_t n, int8_t g, int8_t e, int8_t q, int8_t c,   // as passed at :2098-2106
                   bool local_alignment, bool want_consensus, bool want_msa,
                   sxg_poa_batch_out* out) {
    std::vector<int32_t> blk_off{0};
    std::vector<int64_t> seq_off{0};
    std::vector<uint8_t> bases;
    std::vector<uint32_t> weights;
    for (auto& b : blocks) {
        for (size_t i = 0; i < b.seqs.size(); ++i) {
            for (char ch : b.seqs[i]) bases.push_back(sxg_code(ch));
            seq_off.push_back((int64_t)bases.size());
            weights.push_back(b.weights[i]);
        }
        blk_off.push_back((int32_t)(seq_off.size() - 1));
    }
    sxg_poa_params p{m, n, g, e, q, c,
                     (uint8_t)(local_alignment ? SXG_MODE_LOCAL : SXG_MODE_GLOBAL), 0};
    sxg_poa_batch_in in{};
    in.n_blocks = (int32_t)blocks.size();
    in.blk_off = blk_off.data();  in.seq_off = seq_off.data();
    in.bases = bases.data();      in.weights = weights.data();
    in.params = &p;               in.per_block_params = 0;   // 1 with -a adaptive tiers (:2028-2062)
    in.want_consensus = want_consensus;  in.want_msa = want_msa;
    int rc = sxg_poa_batch_run(h, &in, out);
    if (rc != SXG_OK) std::cerr << "[smoothxg] POA engine: " << sxg_poa_last_error() << std::endl;
    return rc;   // SXG_E_BLOCK: inspect out->status[b]; the reference would have exit(1)'d (:943)
}
// ---- INTEGRATION.md snippet end

int main() {
    std::printf("abi %d, %d HIP device(s)\n", sxg_poa_abi_version(), sxg_poa_device_count());
    sxg_poa_handle* h = nullptr;
    const int rc = sxg_poa_create(0, &h);
    if (rc != SXG_OK) {   // no GPU here: the engine says so, it has no CPU fallback
        std::printf("create: %d (%s)\n", rc, sxg_poa_last_error());
        return rc == SXG_E_NODEVICE ? 0 : 1;
    }
    std::vector<pending_block_t> blocks(2);
    blocks[0].seqs = {"ACGTACGTACGTTACG", "ACGTACGAACGTTACG", "ACGTACGTACGTACG"}; blocks[0].weights = {1, 2, 1};
    blocks[1].seqs = {"GATTACAGATTACA", "GATTACAGATACA"};                        blocks[1].weights = {1, 1};
    sxg_poa_batch_out out;
    const int r2 = run_poa_on_gpu(h, blocks, 1, -4, -6, -2, -26, -1, true, true, false, &out);
    if (r2 == SXG_OK)
        for (int b = 0; b < out.n_blocks; ++b)
            std::printf("block %d: %lld nodes, %lld edges, consensus of %lld\n", b, (long long)(out.node_off[b + 1] - out.node_off[b]),
                        (long long)(out.edge_off[b + 1] - out.edge_off[b]), (long long)(out.cons_off[b + 1] - out.cons_off[b]));
    sxg_poa_batch_free(&out);
    sxg_poa_destroy(h);
    return r2 == SXG_OK ? 0 : 1;
}
