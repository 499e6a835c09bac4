// poa_bgraph_dev.h -- the normalised BLOCK GRAPH of a block, data-parallel, written against the same execution
// context `Ctx` as poa_graph_dev.h: device code in sxg_poa.hip (Ctx = workgroup), and -- for logic tests only -- host code
// with a one-thread context (tests/csrc/graph_emul.cpp).
//
// Replaces, per block, what smooth_spoa does after its alignments:
//   build_odgi_SPOA      src/smooth.cpp:2576-2654   one node per POA node, a path per sequence with the padding steps trimmed
//                                                   (:2611), consensus path (:2624-2637), nodes no path visits dropped (:2639-2653)
//   (build_odgi_abPOA    src/smooth.cpp:2542-2548   the consensus keeps only nodes some sequence visits)
//   unchop               src/smooth.cpp:935         decree: u+ -> v+ merges when it is the only edge on u's right and on v's
//                                                   left side and no path starts or ends inside the link
//   topological order    src/smooth.cpp:947         decree: Kahn over the edges, smallest id first
//   re-copy              src/smooth.cpp:950-1010    only path-supported edges (:980-994)
// so that the host receives compact block graphs (node sequences, edges, step lists) instead of one node id per base and
// rebuilds nothing.  Same results as sxg_smooth.cpp::cblock_from_raw (tests: tests/test_graph_emul.py on the CPU,
// tests/test_gpu_smooth.py on the GPU).
//
// Ctx must provide what poa_graph_dev.h needs plus
//   int atomic_add(int32_t*, int);   int load_fresh(const int32_t*);   // a load that sees other lanes' atomics
//   int* heap();  int heap_cap();                                        // fast scratch of the serial ordering phase
#pragma once
#include "poa_graph_dev.h"

namespace sxg {

struct BgIn {
    const uint8_t* node_code;   // [V] letters of the block's POA nodes
    int V, E;                   // POA nodes / edges of the block
    const int32_t *e_tail, *e_head;   // [E] POA edges, creation order
    const int32_t* paths;       // node of every base of the block (sequence s at seq_off[s] - seq_off[s0])
    const uint8_t* bases;       // the bases themselves, same indexing (validation: every base sits on a node of its letter)
    const int64_t* seq_off;     // the batch's offsets
    int s0, s1;                 // the block's sequences
    int trim;                   // steps trimmed at both ends of every sequence path (poa_padding)
    const int32_t* cons;        // consensus node ids
    int n_cons;
    int cons_mode;              // 0 = no consensus path, 1 = every consensus node, 2 = only nodes some sequence visits
};
// scratch: [V+2] unless said otherwise
struct BgScratch {
    int32_t *vis, *front, *back;          // node is on a path / first / last node of a path
    int32_t *ecnt, *eoff;                 // CSR over tails: POA edges, then the candidate consensus links
    int32_t *ehead, *eused;               // [E + n_cons + 2]
    int32_t *outdeg, *indeg, *only;       // supported edges of a node; its successor when there is one
    int32_t *nxt, *prv;                   // unchop links
    int32_t *pj0, *pj1, *pd0, *pd1;       // pointer jumping: chain head / distance from it
    int32_t *cid;                         // chain id of a head (rank among heads)
    int32_t *head_of, *tail_of, *clen, *indc, *coff, *newid;   // by chain
    int32_t *csucc;                       // [E + n_cons + 2] chain-level successors (CSR by chain)
    int32_t *cc;                          // [n_cons + 2] consensus after its filter
    int32_t *tmp;                         // [max(V, longest sequence) + 2]
    int32_t *lenN, *odN, *soffN, *eoffN;  // by final node id
    int32_t *flag;                        // [4]
};
struct BgOut {
    int32_t *node_len, *node_outdeg;      // [n] by final node id
    uint8_t* node_indeg;
    char* seq;                            // [kept nodes] node sequences back to back
    int32_t* eto;                         // [edges] heads, ascending per tail
    int32_t* steps;                       // steps of the block's sequences, back to back
    int32_t* nsteps;                      // [s1 - s0] steps per sequence
    int32_t* cons_steps;
    int32_t* counts;                      // n nodes, n edges, sequence bytes, consensus steps, status (0 = fine)
};
enum : int { BGC_NODES = 0, BGC_EDGES, BGC_SEQ, BGC_CONS, BGC_STATUS, BGC_N };

// min-heap of ints on caller-provided storage
SXG_HD void bg_heap_push(int* h, int& n, int v) {
    int i = n++;
    while (i > 0) { const int p = (i - 1) >> 1; if (h[p] <= v) break; h[i] = h[p]; i = p; }
    h[i] = v;
}
SXG_HD int bg_heap_pop(int* h, int& n) {
    const int top = h[0], v = h[--n];
    int i = 0;
    for (;;) {
        int c = 2 * i + 1;
        if (c >= n) break;
        if (c + 1 < n && h[c + 1] < h[c]) ++c;
        if (h[c] >= v) break;
        h[i] = h[c]; i = c;
    }
    if (n > 0) h[i] = v;
    return top;
}

template <class Ctx>
SXG_HD_PHASE void block_graph(Ctx& c, const BgIn I, const BgScratch W, const BgOut O) {
    const int T = c.nthreads(), t = c.tid();
    const int V = I.V, E = I.E, ns = I.s1 - I.s0;
    const int64_t b0 = I.seq_off[I.s0];
    constexpr int GB = 4;
    int err = 0;
    // ---- P0: clear
    for (int v = t; v <= V; v += T) {
        W.vis[v] = 0; W.front[v] = 0; W.back[v] = 0; W.ecnt[v] = 0; W.outdeg[v] = 0; W.indeg[v] = 0; W.only[v] = -1;
        W.nxt[v] = -1; W.prv[v] = -1;
    }
    c.sync();
    // ---- P1: nodes the (trimmed) sequence paths visit, their ends; every base sits on a node of its letter
    for (int s = 0; s < ns; ++s) {
        const int64_t so = I.seq_off[I.s0 + s] - b0;
        const int len = (int)(I.seq_off[I.s0 + s + 1] - I.seq_off[I.s0 + s]);
        const int lo = I.trim, hi = len - I.trim;
        for (int k0 = t; k0 < len; k0 += GB * T) {
            int a[GB], cd[GB], bs[GB];
#pragma unroll
            for (int u = 0; u < GB; ++u) { const int k = k0 + u * T; a[u] = k < len ? I.paths[so + k] : -1; bs[u] = k < len ? (int)I.bases[so + k] : 0; }
#pragma unroll
            for (int u = 0; u < GB; ++u) cd[u] = a[u] >= 0 && a[u] < V ? (int)I.node_code[a[u]] : -1;
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const int k = k0 + u * T;
                if (k >= len) continue;
                if (cd[u] < 0 || cd[u] != (bs[u] > 4 ? 4 : bs[u])) { err = 1; continue; }
                if (k >= lo && k < hi) {
                    W.vis[a[u]] = 1;
                    if (k == lo) W.front[a[u]] = 1;
                    if (k == hi - 1) W.back[a[u]] = 1;
                }
            }
        }
    }
    err = c.reduce_max(err);   // (also the barrier)
    if (err) { if (t == 0) { for (int x = 0; x < BGC_N; ++x) O.counts[x] = 0; O.counts[BGC_STATUS] = ST_INTERNAL; } for (int s = t; s < ns; s += T) O.nsteps[s] = 0; c.sync(); return; }
    // ---- P2: the consensus path (after its filter); its nodes count as visited, its ends as path ends
    int ncc = 0;
    if (I.cons_mode) {
        ncc = array_excl_sum(c, I.n_cons, [&](int k) { return (I.cons_mode == 1 || W.vis[I.cons[k]]) ? 1 : 0; }, W.tmp);
        c.sync();
        for (int k = t; k < I.n_cons; k += T)
            if (I.cons_mode == 1 || W.vis[I.cons[k]]) W.cc[W.tmp[k]] = I.cons[k];
        c.sync();
        for (int k = t; k < ncc; k += T) {
            const int v = W.cc[k];
            W.vis[v] = 1;
            if (k == 0) W.front[v] = 1;
            if (k == ncc - 1) W.back[v] = 1;
        }
        c.sync();
    }
    // ---- P3: CSR of the candidate edges by tail: the POA edges, then (filtered consensus only) the links between
    // consecutive consensus nodes -- a dropped node in between makes a link the POA graph does not have
    const int nx = I.cons_mode == 2 && ncc > 1 ? ncc - 1 : 0;
    for (int e = t; e < E; e += T) c.atomic_add(&W.ecnt[I.e_tail[e]], 1);
    for (int k = t; k < nx; k += T) c.atomic_add(&W.ecnt[W.cc[k]], 1);
    c.sync();
    array_excl_sum(c, V, [&](int v) { return c.load_fresh(&W.ecnt[v]); }, W.eoff);
    if (t == 0) W.eoff[V] = E + nx;
    c.sync();
    for (int v = t; v < V; v += T) W.ecnt[v] = 0;   // (now the fill cursors)
    for (int e = t; e < E + nx; e += T) W.eused[e] = 0;
    c.sync();
    for (int e = t; e < E; e += T) { const int a = I.e_tail[e]; W.ehead[W.eoff[a] + c.atomic_add(&W.ecnt[a], 1)] = I.e_head[e]; }
    c.sync();
    for (int k = t; k < nx; k += T) { const int a = W.cc[k]; W.ehead[W.eoff[a] + c.atomic_add(&W.ecnt[a], 1)] = W.cc[k + 1]; }
    c.sync();
    // ---- P4: path-supported edges = consecutive steps of the trimmed paths (src/smooth.cpp:980-994) and of the consensus
    auto mark = [&](int a, int b) {
        const int lo = W.eoff[a], hi = W.eoff[a + 1];
        for (int x = lo; x < hi; ++x) if (W.ehead[x] == b) { W.eused[x] = 1; return; }
        err = 1;   // (a step pair that is no edge: cannot happen on a POA result)
    };
    for (int s = 0; s < ns; ++s) {
        const int64_t so = I.seq_off[I.s0 + s] - b0;
        const int len = (int)(I.seq_off[I.s0 + s + 1] - I.seq_off[I.s0 + s]);
        const int lo = I.trim, hi = len - I.trim;
        for (int k0 = lo + t; k0 + 1 < hi; k0 += GB * T) {
            int a[GB], b[GB], l0[GB], h0[GB];
#pragma unroll
            for (int u = 0; u < GB; ++u) { const int k = k0 + u * T; a[u] = k + 1 < hi ? I.paths[so + k] : -1; b[u] = k + 1 < hi ? I.paths[so + k + 1] : -1; }
#pragma unroll
            for (int u = 0; u < GB; ++u) l0[u] = a[u] >= 0 ? W.eoff[a[u]] : 0;
#pragma unroll
            for (int u = 0; u < GB; ++u) h0[u] = a[u] >= 0 ? W.ehead[l0[u]] : -2;   // the first successor (most nodes have one)
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                if (a[u] < 0) continue;
                if (h0[u] == b[u] && l0[u] < W.eoff[a[u] + 1]) W.eused[l0[u]] = 1; else mark(a[u], b[u]);
            }
        }
    }
    for (int k = t; k + 1 < ncc; k += T) mark(W.cc[k], W.cc[k + 1]);
    err = c.reduce_max(err);
    if (err) { if (t == 0) { for (int x = 0; x < BGC_N; ++x) O.counts[x] = 0; O.counts[BGC_STATUS] = ST_INTERNAL; } for (int s = t; s < ns; s += T) O.nsteps[s] = 0; c.sync(); return; }
    // ---- P5: degrees over the supported edges
    for (int a = t; a < V; a += T) {
        if (!W.vis[a]) continue;
        int od = 0, first = -1;
        for (int x = W.eoff[a]; x < W.eoff[a + 1]; ++x)
            if (W.eused[x]) { ++od; if (first < 0) first = W.ehead[x]; c.atomic_add(&W.indeg[W.ehead[x]], 1); }
        W.outdeg[a] = od; W.only[a] = first;
    }
    c.sync();
    // ---- P6: unchop links
    for (int u = t; u < V; u += T) {
        if (!W.vis[u] || W.outdeg[u] != 1) continue;
        const int v = W.only[u];
        if (c.load_fresh(&W.indeg[v]) != 1 || W.back[u] || W.front[v]) continue;
        W.nxt[u] = v; W.prv[v] = u;
    }
    c.sync();
    // ---- P7: head of every node's chain and its distance from it, by pointer jumping
    for (int x = t; x < V; x += T) { const int p = W.prv[x]; W.pj0[x] = p >= 0 ? p : x; W.pd0[x] = p >= 0 ? 1 : 0; }
    c.sync();
    SXG_GP int32_t* pj[2] = {(SXG_GP int32_t*)W.pj0, (SXG_GP int32_t*)W.pj1};
    SXG_GP int32_t* pd[2] = {(SXG_GP int32_t*)W.pd0, (SXG_GP int32_t*)W.pd1};
    int cur = 0;
    for (int round = 0; round < 32; ++round) {
        int open = 0;
        for (int x0 = t; x0 < V; x0 += GB * T) {
            int p1[GB], d1[GB], p2[GB], d2[GB];
#pragma unroll
            for (int u = 0; u < GB; ++u) { const int x = x0 + u * T; p1[u] = x < V ? pj[cur][x] : 0; d1[u] = x < V ? pd[cur][x] : 0; }
#pragma unroll
            for (int u = 0; u < GB; ++u) { p2[u] = pj[cur][p1[u]]; d2[u] = pd[cur][p1[u]]; }
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const int x = x0 + u * T;
                if (x >= V) continue;
                pj[cur ^ 1][x] = p2[u]; pd[cur ^ 1][x] = d1[u] + d2[u];
                open |= (int)(p2[u] != p1[u]);
            }
        }
        cur ^= 1;
        open = c.reduce_max(open);
        if (!open) break;
    }
    SXG_GP int32_t* const hd = pj[cur];
    SXG_GP int32_t* const ds = pd[cur];
    // ---- P8: chains numbered by their heads, in node order
    const int nc = array_excl_sum(c, V, [&](int x) { return (W.vis[x] && W.prv[x] < 0) ? 1 : 0; }, W.cid);
    const int n1 = array_excl_sum(c, V, [&](int x) { return W.vis[x] ? 1 : 0; }, W.tmp);   // (kept nodes = sequence bytes)
    c.sync();
    for (int x = t; x < V; x += T) {
        if (!W.vis[x]) continue;
        if (W.prv[x] < 0) { const int ci = W.cid[x]; W.head_of[ci] = x; W.indc[ci] = c.load_fresh(&W.indeg[x]); }
        if (W.nxt[x] < 0) { const int ci = W.cid[hd[x]]; W.tail_of[ci] = x; W.clen[ci] = ds[x] + 1; }
    }
    c.sync();
    // chain-level successors (CSR by chain): the supported edges of the chain's last node
    const int ne = array_excl_sum(c, nc, [&](int ci) { return W.outdeg[W.tail_of[ci]]; }, W.coff);
    if (t == 0) W.coff[nc] = ne;
    c.sync();
    for (int ci = t; ci < nc; ci += T) {
        const int a = W.tail_of[ci];
        int w = W.coff[ci];
        for (int x = W.eoff[a]; x < W.eoff[a + 1]; ++x) if (W.eused[x]) W.csucc[w++] = W.cid[W.ehead[x]];
    }
    c.sync();
    // ---- P9: the order (decree): Kahn, smallest chain first.  One lane; its ready set lives in fast scratch.
    if (t == 0) {
        int* h = c.heap();
        const int cap = c.heap_cap();
        bool spilled = false;
        for (int pass = 0; pass < 2; ++pass) {
            int hn = 0, k = 0;
            bool over = false;
            if (pass == 1) { h = (int*)W.tmp; for (int ci = 0; ci < nc; ++ci) W.indc[ci] = c.load_fresh(&W.indeg[W.head_of[ci]]); }   // (the ready set outgrew the fast scratch: again, in HBM)
            const int lim = pass == 0 ? cap : nc + 1;
            for (int ci = 0; ci < nc && !over; ++ci) if (W.indc[ci] == 0) { if (hn >= lim) over = true; else h[hn++] = ci; }   // (ascending: already a heap)
            while (hn > 0 && !over) {
                const int ci = bg_heap_pop(h, hn);
                W.newid[ci] = k++;
                for (int x = W.coff[ci]; x < W.coff[ci + 1]; ++x) {
                    const int cb = W.csucc[x];
                    if (--W.indc[cb] == 0) { if (hn >= lim) { over = true; break; } bg_heap_push(h, hn, cb); }
                }
            }
            if (!over) { if (k != nc) spilled = true; break; }   // (k != nc: a cycle -- impossible on a POA graph)
        }
        W.flag[0] = spilled ? 1 : 0;
    }
    c.sync();
    if (c.load_fresh(&W.flag[0])) { if (t == 0) { for (int x = 0; x < BGC_N; ++x) O.counts[x] = 0; O.counts[BGC_STATUS] = ST_INTERNAL; } for (int s = t; s < ns; s += T) O.nsteps[s] = 0; c.sync(); return; }
    // ---- P10: nodes and edges in the final order
    for (int ci = t; ci < nc; ci += T) {
        const int id = W.newid[ci], a = W.tail_of[ci];
        W.lenN[id] = W.clen[ci]; W.odN[id] = W.outdeg[a];
        O.node_len[id] = W.clen[ci]; O.node_outdeg[id] = W.outdeg[a];
        const int idg = c.load_fresh(&W.indeg[W.head_of[ci]]);
        O.node_indeg[id] = (uint8_t)(idg > 255 ? 255 : idg);
    }
    c.sync();
    array_excl_sum(c, nc, [&](int id) { return W.lenN[id]; }, W.soffN);
    array_excl_sum(c, nc, [&](int id) { return W.odN[id]; }, W.eoffN);
    c.sync();
    for (int x = t; x < V; x += T) {   // letters: a node's place in its chain is its distance from the head
        if (!W.vis[x]) continue;
        const int cd = I.node_code[x] > 4 ? 4 : I.node_code[x];
        O.seq[W.soffN[W.newid[W.cid[hd[x]]]] + ds[x]] = (char)((0x4E54474341ull >> (8 * cd)) & 0xffu);   // "ACGTN"
    }
    for (int ci = t; ci < nc; ci += T) {   // edges of a node: the chain-level successors in their final numbering, ascending
        const int w0 = W.eoffN[W.newid[ci]], n = W.coff[ci + 1] - W.coff[ci];
        for (int k = 0; k < n; ++k) {
            const int v = W.newid[W.csucc[W.coff[ci] + k]];
            int j = k;
            while (j > 0 && O.eto[w0 + j - 1] > v) { O.eto[w0 + j] = O.eto[w0 + j - 1]; --j; }
            O.eto[w0 + j] = v;
        }
    }
    // ---- P11: paths: a chain is stepped on once, at its head (a path enters a chain nowhere else)
    int base = 0;
    for (int s = 0; s < ns; ++s) {
        const int64_t so = I.seq_off[I.s0 + s] - b0;
        const int len = (int)(I.seq_off[I.s0 + s + 1] - I.seq_off[I.s0 + s]);
        const int lo = I.trim, m = len - 2 * I.trim;
        int cnt = 0;
        if (m > 0) {
            cnt = array_excl_sum(c, m, [&](int k) { return W.prv[I.paths[so + lo + k]] < 0 ? 1 : 0; }, W.tmp);
            c.sync();
            for (int k0 = t; k0 < m; k0 += GB * T) {
                int x[GB], id[GB];
#pragma unroll
                for (int u = 0; u < GB; ++u) x[u] = k0 + u * T < m ? I.paths[so + lo + k0 + u * T] : -1;
#pragma unroll
                for (int u = 0; u < GB; ++u) id[u] = x[u] >= 0 && W.prv[x[u]] < 0 ? W.newid[W.cid[x[u]]] : -1;
#pragma unroll
                for (int u = 0; u < GB; ++u) if (id[u] >= 0) O.steps[base + W.tmp[k0 + u * T]] = id[u];
            }
            c.sync();
        }
        if (t == 0) O.nsteps[s] = cnt;
        base += cnt;
    }
    int ncs = 0;
    if (I.cons_mode && ncc > 0) {
        ncs = array_excl_sum(c, ncc, [&](int k) { return W.prv[W.cc[k]] < 0 ? 1 : 0; }, W.tmp);
        c.sync();
        for (int k = t; k < ncc; k += T) { const int x = W.cc[k]; if (W.prv[x] < 0) O.cons_steps[W.tmp[k]] = W.newid[W.cid[x]]; }
    }
    if (t == 0) { O.counts[BGC_NODES] = nc; O.counts[BGC_EDGES] = ne; O.counts[BGC_SEQ] = n1; O.counts[BGC_CONS] = ncs; O.counts[BGC_STATUS] = ST_OK; }
    c.sync();
}

}  // namespace sxg
