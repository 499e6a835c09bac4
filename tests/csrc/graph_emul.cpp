// graph_emul.cpp -- TEST HARNESS ONLY.  Compiles the product's data-parallel graph code
// (smoothxg_amd/csrc/poa_graph_dev.h) for the host with a one-thread execution context and
// drives it with the oracle's DP, so the graph-update / rank / row-prep / consensus logic
// can be checked against the oracle without a GPU.  Never part of the shipped library.
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../smoothxg_amd/csrc/poa_graph_dev.h"
#include "../../smoothxg_amd/csrc/poa_bgraph_dev.h"
#include "../../oracle/poa_oracle.h"

using namespace sxg;

struct SerialCtx {
    static constexpr int GB = 4, GBH = 4, SCAN_K = 4;
    int tid() const { return 0; }
    int nthreads() const { return 1; }
    void sync() {}
    int scan_excl_add(int v, int* total) { *total = v; return 0; }
    int scan_excl_max(int v, int* total) { *total = v; return -0x7fffffff; }
    int reduce_max(int v) { return v; }
    int atomic_add(int32_t* p, int v) { int o = *p; *p += v; return o; }
    int atomic_min(int32_t* p, int v) { int o = *p; if (v < o) *p = v; return o; }
    int load_fresh(const int32_t* p) { return *p; }
    std::vector<int> hp = std::vector<int>(4);   // (tiny on purpose: the ordering phase's spill to HBM gets exercised)
    int* heap() { return hp.data(); }
    int heap_cap() { return (int)hp.size(); }
};

extern "C" int emul_block_run(const uint8_t* bases, const int32_t* seq_off, int n_seqs,
                              const uint32_t* weights, const poa_params_t* p, int pool_slots,
                              int32_t* out_counts /* n_nodes, n_edges, n_cons */, uint8_t* code,
                              int32_t* rank, int32_t* leader, int32_t* e_tail, int32_t* e_head,
                              uint32_t* e_w, int32_t* paths, int32_t* scores, int32_t* cons, int32_t* row_hints,
                              int32_t* row_remain, int spoa_order) {
    int64_t cap = 1, maxlen = 1;
    for (int s = 0; s < n_seqs; ++s) {
        cap += seq_off[s + 1] - seq_off[s];
        if (seq_off[s + 1] - seq_off[s] > maxlen) maxlen = seq_off[s + 1] - seq_off[s];
    }
    const size_t C = (size_t)cap + 2;
    std::vector<int32_t> hdr(2, 0), rk(C), ord(C), ordt(C), ld(C), gm(5 * C), ih(C), it(C), oh(C), ot(C),
        id(C), od(C), et(C), eh(C), eni(C), eno(C), posn(C), tgt(C), nidx(C), nxa(C), pva(C), sla(C), xps(C);
    std::vector<int32_t> via(C), dfs(8 * C + 8), drec(16 * C + 16), sprank(C), spfirst(C), spcnt(C);
    std::vector<uint8_t> lst(4 * C + 64);
    std::vector<uint32_t> ew(C);
    std::vector<uint8_t> cd(C);
    std::vector<int8_t> kd(C);
    GraphView G{&hdr[0], &hdr[1], cd.data(), rk.data(), ord.data(), ordt.data(), ld.data(), gm.data(),
                ih.data(), it.data(), oh.data(), ot.data(), id.data(), od.data(), et.data(), eh.data(),
                eni.data(), eno.data(), ew.data(), posn.data(), tgt.data(), nidx.data(), nxa.data(),
                pva.data(), sla.data(), kd.data(), xps.data(), via.data(), dfs.data(), drec.data(), sprank.data(), spfirst.data(), spcnt.data()};
    std::vector<uint8_t> rcode(C), rflags(C), sink(C);
    std::vector<int32_t> poff(C + 1), preds(C), slot(C), tbx(C), sseq(C + 1), rnode(C), meta(8 * C);
    RowsView R{rcode.data(), rflags.data(), poff.data(), preds.data(), slot.data(), tbx.data(), sseq.data(),
               rnode.data(), meta.data()};
    RowCaps caps{(int)C, pool_slots, (int)C};
    SerialCtx c;
    std::vector<int32_t> an(2 * C), ap(2 * C);
    for (int s = 0; s < n_seqs; ++s) {
        const uint8_t* seq = bases + seq_off[s];
        const int len = seq_off[s + 1] - seq_off[s];
        for (int i = 0; i < len; ++i) posn[i] = -1;
        int32_t sc = 0;
        if (hdr[0] > 0 && len > 0) {
            int st = prep_rows(c, G, R, caps);
            if (st != ST_OK) return st;
            for (int r = 0; r < hdr[0]; ++r) sink[r] = (rflags[r] & ROW_SINK) ? 1 : 0;
            int n = poa_align_csr(hdr[0], rcode.data(), poff.data(), preds.data(), sink.data(), seq, len, p,
                                  an.data(), ap.data(), &sc);
            for (int k = 0; k < n; ++k)
                if (an[k] >= 0 && ap[k] >= 0) posn[ap[k]] = rnode[an[k]];
        }
        if (scores) scores[s] = sc;
        const int n_before = hdr[0];
        add_alignment(c, G, seq, len, weights ? weights[s] : 1u, paths + seq_off[s], spoa_order == 0 || spoa_order == 3);   // (3: the kept order maintained as well)
        // S7' (2: the per-node words in the "LDS" copy whatever the graph's size, and the first sequence named as the static chain;
        //      3: in the "LDS" copy, no static chain; 1: only graphs below 16 nodes on the chip, the others in the slot's scratch)
        //      4: as 2, but every re-sort builds everything (nothing kept from the previous one)
        if (spoa_order) spoa_resort(c, G, lst.data(), spoa_order == 1 ? 64 : (int)lst.size(), (spoa_order & 1) ? 0 : seq_off[1] - seq_off[0], len, n_before,
                                    s == 0 || spoa_order == 4);
    }
    std::vector<int64_t> csc(C);
    std::vector<int32_t> cpr(C);
    out_counts[0] = hdr[0]; out_counts[1] = hdr[1];
    out_counts[2] = consensus_serial(G, csc.data(), cpr.data(), cons);
    memcpy(code, cd.data(), hdr[0]);
    memcpy(rank, rk.data(), 4 * (size_t)hdr[0]);
    memcpy(leader, ld.data(), 4 * (size_t)hdr[0]);
    memcpy(e_tail, et.data(), 4 * (size_t)hdr[1]);
    memcpy(e_head, eh.data(), 4 * (size_t)hdr[1]);
    memcpy(e_w, ew.data(), 4 * (size_t)hdr[1]);
    if (row_hints) for (int r = 0; r < hdr[0]; ++r) row_hints[r] = xps[ord[r]];   // band hints of the packed sweep (decree B2)
    if (row_remain && hdr[0] > 0) rows_remain(c, G, hdr[0], row_remain);          // decree B4 (pointer jumping over heaviest out-edges)
    return 0;
}


// The block-graph phase (poa_bgraph_dev.h) on one block's POA results.  Outputs are sized by the caller:
// node arrays [V], eto [E + n_cons], steps [total bases], nsteps [n_seqs], cons_steps [n_cons], counts [5].
extern "C" int emul_block_graph(const uint8_t* node_code, int V, const int32_t* e_tail, const int32_t* e_head, int E,
                                const int32_t* paths, const uint8_t* bases, const int64_t* seq_off, int n_seqs, int trim,
                                const int32_t* cons, int n_cons, int cons_mode, int32_t* node_len, int32_t* node_outdeg,
                                uint8_t* node_indeg, char* seq, int32_t* eto, int32_t* steps, int32_t* nsteps, int32_t* cons_steps,
                                int32_t* counts) {
    int64_t maxlen = 1;
    for (int s = 0; s < n_seqs; ++s) maxlen = std::max<int64_t>(maxlen, seq_off[s + 1] - seq_off[s]);
    const size_t C = (size_t)V + 2, EC = (size_t)E + (size_t)n_cons + 2, TC = (size_t)std::max<int64_t>(std::max<int64_t>(V, maxlen), n_cons) + 2;
    std::vector<std::vector<int32_t>> a(32, std::vector<int32_t>(C));
    std::vector<int32_t> ehead(EC), eused(EC), csucc(EC), cc((size_t)n_cons + 2), tmp(TC), flag(4);
    BgScratch W{a[0].data(), a[1].data(), a[2].data(), a[3].data(), a[4].data(), ehead.data(), eused.data(), a[5].data(), a[6].data(), a[7].data(),
                a[8].data(), a[9].data(), a[10].data(), a[11].data(), a[12].data(), a[13].data(), a[14].data(), a[15].data(), a[16].data(),
                a[17].data(), a[18].data(), a[19].data(), a[20].data(), csucc.data(), cc.data(), tmp.data(), a[21].data(), a[22].data(),
                a[23].data(), a[24].data(), flag.data()};
    BgIn I{node_code, V, E, e_tail, e_head, paths, bases, seq_off, 0, n_seqs, trim, cons, n_cons, cons_mode};
    BgOut O{node_len, node_outdeg, node_indeg, seq, eto, steps, nsteps, cons_steps, counts};
    SerialCtx c;
    block_graph(c, I, W, O);
    return counts[BGC_STATUS];
}

// Layout of the packed sweeps' traceback plane (poa_types.h): out[slot * W + k] = dword of cell (slot, k) inside its row.
// Also replays what the sweeps' 16-byte stores do -- group gi of a lane's strip goes to dwords
// [4 * gi * BS + slot * gw, + gw) -- and returns the number of cells whose stored place differs from plane_cell_in_row.
extern "C" int emul_plane_layout(int W, int BS, int32_t* out) {
    int bad = 0;
    for (int slot = 0; slot < BS; ++slot)
        for (int k = 0; k < W; ++k) out[slot * W + k] = sxg::plane_cell_in_row(W, BS, slot, k);
    const int NG = (W + 3) / 4;
    for (int slot = 0; slot < BS; ++slot)
        for (int gi = 0; gi < NG; ++gi) {
            const int gw = sxg::plane_group_width(W, gi);
            for (int x = 0; x < gw; ++x)
                if (4 * gi * BS + slot * gw + x != out[slot * W + 4 * gi + x]) ++bad;
        }
    return bad;
}
