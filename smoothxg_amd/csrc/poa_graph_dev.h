// poa_graph_dev.h -- data-parallel POA graph maintenance, written against an execution
// context `Ctx` so the SAME source runs as device code inside the persistent block kernel
// (Ctx = workgroup: LDS scans, s_barrier) and, for logic tests only, on the host with a
// one-thread context (tests/csrc/graph_emul.cpp).  There is no host fallback in the
// product: smoothxg_amd/csrc/sxg_poa.hip only instantiates the device context.
//
// Replaces spoa::Graph::AddAlignment + TopologicalSort (call site src/smooth.cpp:764) and
// spoa::Graph::GenerateConsensus (src/smooth.cpp:773).  Semantics S6-S8 of DESIGN.md.
//
// Ctx must provide:
//   int tid(), nthreads();  void sync();
//   int scan_excl_add(int v, int* total);   // block-wide exclusive +scan in tid order
//   int scan_incl_max(int v);               // block-wide inclusive max-scan in tid order
//   int reduce_max(int v);                  // block-wide max, result on every thread
//   int atomic_add(int32_t* p, int v);
#pragma once
#include "poa_types.h"

namespace sxg {

template <class Ctx, class F, class P>
SXG_HD int array_excl_sum(Ctx& c, int n, F get, P out) {
    const int T = c.nthreads(), t = c.tid();
    int carry = 0;
    for (int base = 0; base < n; base += T) {
        const int i = base + t;
        const int v = i < n ? get(i) : 0;
        int tot;
        const int p = c.scan_excl_add(v, &tot);
        if (i < n) out[i] = carry + p;
        carry += tot;
    }
    return carry;
}

// out[i] = max_{x<=i} get(x)
template <class Ctx, class F, class P>
SXG_HD void array_incl_max(Ctx& c, int n, F get, P out) {
    const int T = c.nthreads(), t = c.tid();
    int carry = -0x7fffffff;
    for (int base = 0; base < n; base += T) {
        const int i = base + t;
        int v = i < n ? get(i) : -0x7fffffff;
        v = c.scan_incl_max(v);
        if (v < carry) v = carry;
        if (i < n) out[i] = v;
        carry = c.reduce_max(v);
    }
}

// out[i] = min_{x>=i} get(x)
template <class Ctx, class F, class P>
SXG_HD void array_suffix_min(Ctx& c, int n, F get, P out) {
    const int T = c.nthreads(), t = c.tid();
    int carry = -0x7fffffff;
    for (int base = 0; base < n; base += T) {
        const int ir = base + t;          // reversed index
        const int i = n - 1 - ir;
        int v = ir < n ? -get(i) : -0x7fffffff;
        v = c.scan_incl_max(v);
        if (v < carry) v = carry;
        if (ir < n) out[i] = -v;
        carry = c.reduce_max(v);
    }
}

SXG_HD int group_start(const GraphView& G, int leader, int n_old) {
    int r = 0x7fffffff;
    for (int x = 0; x < 5; ++x) {
        const int v = G.gmem[5 * leader + x];
        if (v >= 0 && v < n_old) { const int rv = G.rank[v]; if (rv < r) r = rv; }
    }
    return r;
}
SXG_HD int group_end(const GraphView& G, int leader, int n_old) {
    int r = -1;
    for (int x = 0; x < 5; ++x) {
        const int v = G.gmem[5 * leader + x];
        if (v >= 0 && v < n_old) { const int rv = G.rank[v]; if (rv > r) r = rv; }
    }
    return r;
}

// S6 + S7.  G.posnode[0..len) holds the aligned node id of every sequence position or -1.
// path_out[0..len) receives the node id of every base (replaces spoa's per-edge labels /
// Node::Successor walk used at src/smooth.cpp:2604-2610).
template <class Ctx>
SXG_HD_PHASE void add_alignment(Ctx& c, const GraphView& G, const uint8_t* seq_, int len,
                          uint32_t weight, int32_t* path_out_) {
    SXG_GP const uint8_t* const seq = (SXG_GP const uint8_t*)seq_;   // both live in HBM
    SXG_GP int32_t* const path_out = (SXG_GP int32_t*)path_out_;
    const int T = c.nthreads(), t = c.tid();
    const int n_old = *G.n_nodes, e_old = *G.n_edges;
    const int BIG = 0x3fffffff;
    c.sync();
    // P1: classify every position: 0 = existing node, 1 = new sibling, 2 = new unaligned
    for (int i = t; i < len; i += T) {
        const int cc = seq[i] > 4 ? 4 : seq[i];
        const int a = G.posnode[i];
        int kind = 2, tg = -1;
        if (a >= 0) {
            const int v = G.gmem[5 * G.leader[a] + cc];
            if (v >= 0) { kind = 0; tg = v; } else kind = 1;
        }
        G.kind[i] = (int8_t)kind;
        G.target[i] = tg;
    }
    c.sync();
    const int n_new = array_excl_sum(c, len, [&](int i) { return G.kind[i] != 0 ? 1 : 0; }, G.newidx);
    array_incl_max(c, len, [&](int i) { return G.kind[i] != 2 ? i : -1; }, G.preva);
    array_suffix_min(c, len, [&](int i) { return G.kind[i] != 2 ? i : BIG; }, G.nexta);
    for (int r = t; r <= n_old; r += T) G.slotadd[r] = 0;
    c.sync();
    // P2: create the new nodes, hand out slots (S7) and final ranks of the new nodes
    for (int i = t; i < len; i += T) {
        const int kind = G.kind[i];
        if (kind == 0) { path_out[i] = G.target[i]; continue; }
        const int cc = seq[i] > 4 ? 4 : seq[i];
        const int k = G.newidx[i];
        const int v = n_old + k;
        G.code[v] = (uint8_t)cc;
        G.in_head[v] = G.in_tail[v] = G.out_head[v] = G.out_tail[v] = -1;
        G.in_deg[v] = G.out_deg[v] = 0;
        for (int x = 0; x < 5; ++x) G.gmem[5 * v + x] = -1;
        int slot;
        if (kind == 1) {
            const int ld = G.leader[G.posnode[i]];
            G.leader[v] = ld;
            slot = group_end(G, ld, n_old) + 1;
            G.xpos[v] = G.xpos[G.posnode[i]];
        } else {
            G.leader[v] = v;
            G.gmem[5 * v + cc] = v;
            const int s = G.nexta[i], p = G.preva[i];
            // band hint: continue the backbone coordinate of the nearest aligned neighbour
            G.xpos[v] = p >= 0 ? G.xpos[G.posnode[p]] + (i - p) : (s < len ? G.xpos[G.posnode[s]] - (s - i) : i + 1);
            if (s < len) {
                const int anchor = G.kind[s] == 0 ? G.target[s] : G.posnode[s];
                slot = group_start(G, G.leader[anchor], n_old);
            } else if (p >= 0) {
                const int anchor = G.kind[p] == 0 ? G.target[p] : G.posnode[p];
                slot = group_end(G, G.leader[anchor], n_old) + 1;
            } else slot = n_old;
        }
        path_out[i] = v;
        G.rank[v] = slot + k;
        G.order_tmp[slot + k] = v;
        c.atomic_add(&G.slotadd[slot], 1);
    }
    c.sync();
    for (int i = t; i < len; i += T) {
        if (G.kind[i] == 0) continue;
        const int v = n_old + G.newidx[i];
        if (G.kind[i] == 1) G.gmem[5 * G.leader[v] + G.code[v]] = v;
        G.target[i] = v;
    }
    // P3: shift the old nodes.  preva is free again: reuse as exclusive slot counts.
    array_excl_sum(c, n_old, [&](int r) { return G.slotadd[r]; }, G.preva);
    c.sync();
    for (int r = t; r < n_old; r += T) {
        const int nr = r + G.preva[r] + G.slotadd[r];
        const int v = G.order[r];
        G.rank[v] = nr;
        G.order_tmp[nr] = v;
    }
    c.sync();
    for (int r = t; r < n_old + n_new; r += T) G.order[r] = G.order_tmp[r];
    // P4: edges between consecutive path nodes, weight += 2*w (S6)
    for (int i = t; i < len; i += T) {
        int isnew = 0;
        if (i >= 1) {
            const int u = G.target[i - 1], v = G.target[i];
            int found = -1;
            if (G.kind[i - 1] == 0 && G.kind[i] == 0)
                for (int e = G.out_head[u]; e >= 0; e = G.e_next_out[e])
                    if (G.e_head[e] == v) { found = e; break; }
            if (found >= 0) G.e_w[found] += 2u * weight; else isnew = 1;
        }
        G.nexta[i] = isnew;
    }
    c.sync();
    const int n_newe = array_excl_sum(c, len, [&](int i) { return G.nexta[i]; }, G.preva);
    c.sync();
    for (int i = t; i < len; i += T) {
        if (!G.nexta[i]) continue;
        const int e = e_old + G.preva[i];
        const int u = G.target[i - 1], v = G.target[i];
        G.e_tail[e] = u; G.e_head[e] = v; G.e_w[e] = 2u * weight;
        G.e_next_in[e] = G.e_next_out[e] = -1;
        // every node is tail of at most one and head of at most one new edge per alignment
        if (G.out_tail[u] >= 0) G.e_next_out[G.out_tail[u]] = e; else G.out_head[u] = e;
        G.out_tail[u] = e; G.out_deg[u] += 1;
        if (G.in_tail[v] >= 0) G.e_next_in[G.in_tail[v]] = e; else G.in_head[v] = e;
        G.in_tail[v] = e; G.in_deg[v] += 1;
    }
    c.sync();
    if (t == 0) { *G.n_nodes = n_old + n_new; *G.n_edges = e_old + n_newe; }
    c.sync();
}

struct RowCaps {
    int rows_cap;   // rows of the traceback plane
    int pool_slots; // row-pool slots
    int step_cap;   // fold steps of the step-mask plane (sum over multi-pred rows of np-1)
};

// Second half of row preparation, shared by the block kernel (graph -> rows) and the
// stand-alone align kernel (caller CSR -> rows).  On entry R.flags holds STORE/SINK bits and
// R.slot[r] the rank of the last reader of row r.  Assigns ring slots of the row pool and
// the step-mask plane offset of multi-pred rows.  Returns a status (same on every thread).
// `hinted`: 1 = packed sweep -- R.tbx already holds the band hint of every row and there is no step-mask plane;
// 2 = banded sweep -- additionally there is no row ring (every row keeps its band in the plane) and the descriptor
// carries the hints of the first two predecessors where the ring slots would be.
template <class Ctx>
SXG_HD_PHASE int finish_rows(Ctx& c, int N, const RowsView& R, const RowCaps& caps, const int hinted = 0) {
    const int T = c.nthreads(), t = c.tid();
    c.sync();
    const int n_store = array_excl_sum(c, N, [&](int r) { return (R.flags[r] & ROW_STORE) ? 1 : 0; }, R.sseq);
    if (t == 0) R.sseq[N] = n_store;
    c.sync();
    int worst = 0;
    for (int r = t; r < N; r += T) {
        if (!(R.flags[r] & ROW_STORE)) { R.slot[r] = -1; continue; }
        const int lu = R.slot[r];
        const int cnt = R.sseq[lu] - R.sseq[r];  // stored rows in [r, lu): must fit the ring
        if (cnt > worst) worst = cnt;
        R.slot[r] = R.sseq[r] % caps.pool_slots;
    }
    worst = c.reduce_max(worst);
    if (worst > caps.pool_slots && hinted != 2) return ST_POOL_OVERFLOW;
    if (!hinted) {
        // multi-pred rows: np-1 fold steps each in the step-mask plane
        const int n_steps = array_excl_sum(c, N, [&](int r) {
            const int d = R.pred_off[r + 1] - R.pred_off[r];
            return d > 1 ? d - 1 : 0; }, R.tbx);
        c.sync();
        if (n_steps > caps.step_cap) return ST_TBX_OVERFLOW;
        for (int r = t; r < N; r += T)
            if (R.pred_off[r + 1] - R.pred_off[r] <= 1) R.tbx[r] = -1;
        c.sync();
    }
    for (int r = t; r < N; r += T) {
        const int pb = R.pred_off[r], np = R.pred_off[r + 1] - pb;
        RowMeta m;
        m.pb = pb;
        m.info = np | ((int)R.code[r] << 16) | ((int)R.flags[r] << 24);
        m.p0 = np >= 1 ? R.preds[pb] : 0;
        m.s0 = (m.p0 >= 1 && m.p0 != r) ? R.slot[m.p0 - 1] : -1;
        m.p1 = np >= 2 ? R.preds[pb + 1] : 0;
        m.s1 = (m.p1 >= 1 && m.p1 != r) ? R.slot[m.p1 - 1] : -1;
        if (hinted == 2) { m.s0 = m.p0 >= 1 ? R.tbx[m.p0 - 1] : 0; m.s1 = m.p1 >= 1 ? R.tbx[m.p1 - 1] : 0; }
        m.slot = R.slot[r];
        m.tbx = R.tbx[r];
        int32_t* d = R.meta + 8 * (size_t)r;
        d[0] = m.pb; d[1] = m.info; d[2] = m.p0; d[3] = m.s0; d[4] = m.p1; d[5] = m.s1; d[6] = m.slot; d[7] = m.tbx;
    }
    c.sync();
    return ST_OK;
}

// Rank-space CSR + per-row DP metadata of the current graph.  Returns a status code
// (identical on every thread).
template <class Ctx>
SXG_HD_PHASE int prep_rows(Ctx& c, const GraphView& G, const RowsView& R, const RowCaps& caps, const int hinted = 0) {
    const int T = c.nthreads(), t = c.tid();
    c.sync();
    const int N = *G.n_nodes;
    if (N > caps.rows_cap) return ST_ROWS_OVERFLOW;
    for (int r = t; r < N; r += T) {
        const int v = G.order[r];
        R.row_node[r] = v;
        R.code[r] = G.code[v];
        if (hinted) R.tbx[r] = G.xpos[v];
    }
    const int E = array_excl_sum(c, N, [&](int r) { return G.in_deg[G.order[r]]; }, R.pred_off);
    if (t == 0) R.pred_off[N] = E;
    c.sync();
    // preds in rank space; store / sink flags; last reader of every row (kept in R.slot)
    for (int r = t; r < N; r += T) {
        const int v = G.order[r];
        int o = R.pred_off[r];
        for (int e = G.in_head[v]; e >= 0; e = G.e_next_in[e]) R.preds[o++] = G.rank[G.e_tail[e]] + 1;
        int store = 0, lu = r;
        for (int e = G.out_head[v]; e >= 0; e = G.e_next_out[e]) {
            const int hr = G.rank[G.e_head[e]];
            if (hr != r + 1) store = 1;
            if (hr > lu) lu = hr;
        }
        R.flags[r] = (uint8_t)((store ? ROW_STORE : 0) | (G.out_deg[v] == 0 ? ROW_SINK : 0));
        R.slot[r] = lu;
    }
    return finish_rows(c, N, R, caps, hinted);
}

// S8: heaviest bundle + branch completion.  One thread; scores in sc (int64), preds in pr.
SXG_HD int branch_completion(const GraphView& G, int N, int r, int64_t* sc, int32_t* pr) {
    const int start = G.order[r];
    for (int e = G.out_head[start]; e >= 0; e = G.e_next_out[e])
        for (int f = G.in_head[G.e_head[e]]; f >= 0; f = G.e_next_in[f])
            if (G.e_tail[f] != start) sc[G.e_tail[f]] = -1;
    int64_t max_score = 0;
    int max_node = -1;
    for (int i = r + 1; i < N; ++i) {
        const int v = G.order[i];
        sc[v] = -1; pr[v] = -1;
        for (int e = G.in_head[v]; e >= 0; e = G.e_next_in[e]) {
            const int tl = G.e_tail[e];
            if (sc[tl] == -1) continue;
            const int64_t w = G.e_w[e];
            if (sc[v] < w || (sc[v] == w && sc[pr[v]] <= sc[tl])) { sc[v] = w; pr[v] = tl; }
        }
        if (pr[v] != -1) sc[v] += sc[pr[v]];
        if (max_score < sc[v]) { max_score = sc[v]; max_node = v; }
    }
    return max_node;
}

// Writes the consensus node ids (forward order) to out; returns the length.  tmp needs N ints.
SXG_HD_PHASE int consensus_serial(const GraphView& G, int64_t* sc, int32_t* pr, int32_t* out) {
    const int N = *G.n_nodes;
    if (N == 0) return 0;
    for (int v = 0; v < N; ++v) { sc[v] = -1; pr[v] = -1; }
    int mx = -1;
    for (int r = 0; r < N; ++r) {
        const int v = G.order[r];
        for (int e = G.in_head[v]; e >= 0; e = G.e_next_in[e]) {
            const int tl = G.e_tail[e];
            const int64_t w = G.e_w[e];
            if (sc[v] < w || (sc[v] == w && sc[pr[v]] <= sc[tl])) { sc[v] = w; pr[v] = tl; }
        }
        if (pr[v] != -1) sc[v] += sc[pr[v]];
        if (mx == -1 || sc[mx] < sc[v]) mx = v;
    }
    while (G.out_deg[mx] != 0) {
        const int nx = branch_completion(G, N, G.rank[mx], sc, pr);
        if (nx < 0) break;
        mx = nx;
    }
    int n = 0;
    for (int v = mx; v != -1; v = pr[v]) out[n++] = v;
    for (int a = 0, b = n - 1; a < b; ++a, --b) { const int32_t x = out[a]; out[a] = out[b]; out[b] = x; }
    return n;
}

}  // namespace sxg
