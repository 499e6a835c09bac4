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
//   int scan_excl_max(int v, int* total);   // block-wide exclusive max-scan in tid order (-0x7fffffff before thread 0)
//   int reduce_max(int v);                  // block-wide max, result on every thread
//   int atomic_add(int32_t* p, int v);
#pragma once
#include "poa_types.h"

namespace sxg {

// The scans below give every thread SCAN_K consecutive elements: the block-wide scan (LDS hop, two barriers) runs
// once per SCAN_K * T elements, and the SCAN_K loads of a thread are independent -- with one wave per block (1 kbp
// blocks, the banded sweep) these loops are bound by the latency of dependent HBM loads, not by instructions.
// SCAN_K and the batch sizes GB / GBH of the loops further down belong to the execution context (round 5): a one-wave
// workgroup takes 16 elements per thread and step, so that a graph of ~1 000 nodes is ONE round of independent loads per stage
// instead of five (graph phases of 8000 x 16 x 1 kbp: 28 % of the slot time with 4 everywhere).

template <class Ctx, class F, class P>
SXG_HD int array_excl_sum(Ctx& c, int n, F get, P out) {
    constexpr int SCAN_K = Ctx::SCAN_K;
    const int T = c.nthreads(), t = c.tid();
    int carry = 0;
    for (int base = 0; base < n; base += SCAN_K * T) {
        const int i0 = base + SCAN_K * t;
        int v[SCAN_K], s = 0;
#pragma unroll
        for (int k = 0; k < SCAN_K; ++k) v[k] = i0 + k < n ? get(i0 + k) : 0;
#pragma unroll
        for (int k = 0; k < SCAN_K; ++k) s += v[k];
        int tot;
        int p = carry + c.scan_excl_add(s, &tot);
#pragma unroll
        for (int k = 0; k < SCAN_K; ++k) {
            if (i0 + k < n) out[i0 + k] = p;
            p += v[k];
        }
        carry += tot;
    }
    return carry;
}

// out[i] = max_{x<=i} get(x)
template <class Ctx, class F, class P>
SXG_HD void array_incl_max(Ctx& c, int n, F get, P out) {
    constexpr int SCAN_K = Ctx::SCAN_K;
    const int T = c.nthreads(), t = c.tid();
    const int NONE = -0x7fffffff;
    int carry = NONE;
    for (int base = 0; base < n; base += SCAN_K * T) {
        const int i0 = base + SCAN_K * t;
        int v[SCAN_K];
#pragma unroll
        for (int k = 0; k < SCAN_K; ++k) v[k] = i0 + k < n ? get(i0 + k) : NONE;
#pragma unroll
        for (int k = 1; k < SCAN_K; ++k) v[k] = v[k] > v[k - 1] ? v[k] : v[k - 1];
        int tot;
        int before = c.scan_excl_max(v[SCAN_K - 1], &tot);
        if (before < carry) before = carry;
#pragma unroll
        for (int k = 0; k < SCAN_K; ++k)
            if (i0 + k < n) out[i0 + k] = v[k] > before ? v[k] : before;
        if (tot > carry) carry = tot;
    }
}

// out[i] = min_{x>=i} get(x)
template <class Ctx, class F, class P>
SXG_HD void array_suffix_min(Ctx& c, int n, F get, P out) {
    constexpr int SCAN_K = Ctx::SCAN_K;
    const int T = c.nthreads(), t = c.tid();
    const int NONE = -0x7fffffff;
    int carry = NONE;
    for (int base = 0; base < n; base += SCAN_K * T) {
        const int r0 = base + SCAN_K * t;          // reversed index of my first element
        int v[SCAN_K];
#pragma unroll
        for (int k = 0; k < SCAN_K; ++k) v[k] = r0 + k < n ? -get(n - 1 - (r0 + k)) : NONE;
#pragma unroll
        for (int k = 1; k < SCAN_K; ++k) v[k] = v[k] > v[k - 1] ? v[k] : v[k - 1];
        int tot;
        int before = c.scan_excl_max(v[SCAN_K - 1], &tot);
        if (before < carry) before = carry;
#pragma unroll
        for (int k = 0; k < SCAN_K; ++k)
            if (r0 + k < n) out[n - 1 - (r0 + k)] = -(v[k] > before ? v[k] : before);
        if (tot > carry) carry = tot;
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
// keep_order = false (round 6): the caller re-sorts the graph right behind this call (decree S7', spoa_resort below, which
// neither reads nor keeps the incrementally kept order): the slots of the new nodes, the shift of the old ones and the copy of
// the order -- the gathers through the groups' ranks and two passes over all nodes -- are left out.
template <class Ctx>
SXG_HD_PHASE void add_alignment(Ctx& c, const GraphView& G_, const uint8_t* seq_, int len,
                          uint32_t weight, int32_t* path_out_, const bool keep_order = true) {
    const GraphView G = sxg_scalar_view(G_);   // (pointers through the scalar unit once: see poa_types.h)
    SXG_GP const uint8_t* const seq = sxg_scalar_ptr((SXG_GP const uint8_t*)seq_);   // both live in HBM
    SXG_GP int32_t* const path_out = sxg_scalar_ptr((SXG_GP int32_t*)path_out_);
    const int T = c.nthreads(), t = c.tid();
    const int n_old = *G.n_nodes, e_old = *G.n_edges;
    const int BIG = 0x3fffffff;
    c.sync();
    // P1: classify every position: 0 = existing node, 1 = new sibling, 2 = new unaligned
    constexpr int GB = Ctx::GB;   // positions per thread and iteration, stage by stage (see prep_rows)
    for (int i0 = t; i0 < len; i0 += GB * T) {
        int cc[GB], a[GB], ld[GB], tg[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int i = i0 + u * T;
            cc[u] = i < len ? (seq[i] > 4 ? 4 : (int)seq[i]) : 0;
            a[u] = i < len ? G.posnode[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < GB; ++u) ld[u] = a[u] >= 0 ? G.leader[a[u]] : -1;
#pragma unroll
        for (int u = 0; u < GB; ++u) tg[u] = ld[u] >= 0 ? G.gmem[5 * ld[u] + cc[u]] : -1;
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int i = i0 + u * T;
            if (i >= len) continue;
            G.kind[i] = (int8_t)(a[u] < 0 ? 2 : (tg[u] >= 0 ? 0 : 1));
            G.target[i] = a[u] >= 0 && tg[u] >= 0 ? tg[u] : -1;
        }
    }
    c.sync();
    const int n_new = array_excl_sum(c, len, [&](int i) { return G.kind[i] != 0 ? 1 : 0; }, G.newidx);
    array_incl_max(c, len, [&](int i) { return G.kind[i] != 2 ? i : -1; }, G.preva);
    array_suffix_min(c, len, [&](int i) { return G.kind[i] != 2 ? i : BIG; }, G.nexta);
    if (keep_order) for (int r = t; r <= n_old; r += T) G.slotadd[r] = 0;
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
        int slot = 0;
        G.via[v] = kind == 1 ? G.posnode[i] : -1;
        if (kind == 1) {
            const int ld = G.leader[G.posnode[i]];
            G.leader[v] = ld;
            if (keep_order) slot = group_end(G, ld, n_old) + 1;
            G.xpos[v] = G.xpos[G.posnode[i]];
        } else {
            G.leader[v] = v;
            G.gmem[5 * v + cc] = v;
            const int s = G.nexta[i], p = G.preva[i];
            // band hint: continue the backbone coordinate of the nearest aligned neighbour
            G.xpos[v] = p >= 0 ? G.xpos[G.posnode[p]] + (i - p) : (s < len ? G.xpos[G.posnode[s]] - (s - i) : i + 1);
            if (!keep_order) slot = 0;
            else if (s < len) {
                const int anchor = G.kind[s] == 0 ? G.target[s] : G.posnode[s];
                slot = group_start(G, G.leader[anchor], n_old);
            } else if (p >= 0) {
                const int anchor = G.kind[p] == 0 ? G.target[p] : G.posnode[p];
                slot = group_end(G, G.leader[anchor], n_old) + 1;
            } else slot = n_old;
        }
        path_out[i] = v;
        if (keep_order) {
            G.rank[v] = slot + k;
            G.order_tmp[slot + k] = v;
            c.atomic_add(&G.slotadd[slot], 1);
        }
    }
    c.sync();
    for (int i = t; i < len; i += T) {
        if (G.kind[i] == 0) continue;
        const int v = n_old + G.newidx[i];
        if (G.kind[i] == 1) G.gmem[5 * G.leader[v] + G.code[v]] = v;
        G.target[i] = v;
    }
    // P3: shift the old nodes.  preva is free again: reuse as exclusive slot counts.
    if (keep_order) {
    array_excl_sum(c, n_old, [&](int r) { return G.slotadd[r]; }, G.preva);
    c.sync();
    for (int r0 = t; r0 < n_old; r0 += GB * T) {
        int nr[GB], v[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int r = r0 + u * T;
            nr[u] = r < n_old ? r + G.preva[r] + G.slotadd[r] : 0;
            v[u] = r < n_old ? G.order[r] : 0;
        }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            if (r0 + u * T >= n_old) continue;
            G.rank[v[u]] = nr[u];
            G.order_tmp[nr[u]] = v[u];
        }
    }
    c.sync();
    for (int r0 = t; r0 < n_old + n_new; r0 += GB * T) {
        int v[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) v[u] = r0 + u * T < n_old + n_new ? G.order_tmp[r0 + u * T] : 0;
#pragma unroll
        for (int u = 0; u < GB; ++u) if (r0 + u * T < n_old + n_new) G.order[r0 + u * T] = v[u];
    }
    }
    // P4: edges between consecutive path nodes, weight += 2*w (S6)
    for (int i0 = t; i0 < len; i0 += GB * T) {
        int u_[GB], v_[GB], both[GB], e0[GB], h0[GB], n0[GB];   // (six values per position: fits GB = 16 in 128 registers)
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int i = i0 + u * T;
            const bool ok = i >= 1 && i < len;
            u_[u] = ok ? G.target[i - 1] : -1;
            v_[u] = ok ? G.target[i] : -1;
            both[u] = ok && G.kind[i - 1] == 0 && G.kind[i] == 0;
        }
#pragma unroll
        for (int u = 0; u < GB; ++u) e0[u] = both[u] ? G.out_head[u_[u]] : -1;
#pragma unroll
        for (int u = 0; u < GB; ++u) {   // the first out-edge (most nodes have one)
            h0[u] = e0[u] >= 0 ? G.e_head[e0[u]] : -1;
            n0[u] = e0[u] >= 0 ? G.e_next_out[e0[u]] : -1;
        }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int i = i0 + u * T;
            if (i >= len) continue;
            int isnew = 0;
            if (i >= 1) {
                int found = -1;
                if (e0[u] >= 0 && h0[u] == v_[u]) found = e0[u];
                else
                    for (int e = n0[u]; e >= 0; e = G.e_next_out[e])
                        if (G.e_head[e] == v_[u]) { found = e; break; }
                if (found >= 0) G.e_w[found] += 2u * weight; else isnew = 1;
            }
            G.nexta[i] = isnew;
        }
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

// S7' -- spoa's topological order AS RECOLLECTED (rvaser/spoa Graph::TopologicalSort is absent from the reference snapshot:
// unverified; an option -- sxg_poa_params::mode | SXG_ORDER_SPOA -- that switches the one known divergence of decree S7 off
// so that it can be priced; restated in oracle/poa_oracle.c::spoa_resort, same text): for every node in id order, depth-first
// with an explicit stack -- a node on top of the stack pushes the tails of its in-edges that are not done (insertion order)
// and, unless it was itself pushed as somebody's aligned node ("ignored"), its aligned nodes that are not done (marking them
// ignored); when nothing had to be pushed it is done and, unless ignored, emitted, followed by its aligned nodes in the
// order of ITS aligned-node list.  That list follows spoa's AddAlignment: a node created aligned to x starts with x's list
// followed by x, and is appended to the list of every member the group already had.
// A chain of dependent loads over the whole graph: ONE lane runs it (the caller's thread 0), after add_alignment.
SXG_HD int spoa_aligned_list(const GraphView& G, int v, int* out) {
    int mem[5], nm = 0;
    const int ld = G.leader[v];
    for (int c = 0; c < 5; ++c) { const int x = G.gmem[5 * ld + c]; if (x >= 0) mem[nm++] = x; }
    for (int a = 1; a < nm; ++a) { const int x = mem[a]; int b = a; while (b > 0 && mem[b - 1] > x) { mem[b] = mem[b - 1]; --b; } mem[b] = x; }   // creation order
    int list[5][5], ln[5];
    for (int k = 0; k < nm; ++k) {
        ln[k] = 0;
        int xi = -1;
        const int via = G.via[mem[k]];
        for (int j = 0; j < k; ++j) if (mem[j] == via) xi = j;
        if (k > 0 && xi < 0) xi = 0;
        if (xi >= 0) { for (int q = 0; q < ln[xi]; ++q) list[k][ln[k]++] = list[xi][q]; list[k][ln[k]++] = mem[xi]; }
        for (int j = 0; j < k; ++j) list[j][ln[j]++] = mem[k];
    }
    for (int k = 0; k < nm; ++k)
        if (mem[k] == v) { for (int q = 0; q < ln[k]; ++q) out[q] = list[k][q]; return ln[k]; }
    return 0;
}
// Round 5: the walk no longer chases the graph's linked lists.  Every thread of the workgroup first writes, for its share of
// the nodes, what the walk needs of a node into ONE 64-byte record -- the number of in-edges and the first three tails
// (insertion order), the aligned-node list, the in-edge tails of the aligned nodes -- so that a visit loads the record and
// then the states of what it names, instead of a dozen dependent loads (in_head -> e_tail -> marks -> e_next_in ...,
// leader -> gmem -> via ...); a node with more than three in-edges walks its list as before.  A node with aligned nodes
// whose tails, and whose aligned nodes' tails, are all done finishes its whole group in the one visit -- what the walk
// would reach by pushing every aligned node, finding each of them valid, popping back and looking at the node again.
//
// Round 6: THE WALK IS PARALLEL.  The sequential algorithm starts a depth-first walk at every node in id order that is
// not done yet; the walk from root s finishes exactly the nodes of s's dependency closure (in-edge tails and aligned
// nodes, transitively) that no earlier root finished.  So node x is finished by the root
//     first(x) = the smallest id among the nodes that reach x through (tail | aligned node) links
//              = the smallest id x reaches through (out-edge head | aligned node) links, x included,
// the order is the concatenation, over the roots s with first(s) = s in ascending order, of the post-order of s's walk,
// and inside that walk "done" is simply  first(x) < s  or finished by this very walk.  Nothing a walk reads is written by
// another walk: the roots are independent.
//   P1  all threads: the records; first(x) := x; jump(x) := head of x's oldest out-edge.
//   P2  all threads, rounds until nothing changes: first(x) := min over x's out-edge heads, aligned nodes and jump(x);
//       jump(x) := jump(jump(x)).  Values only fall and every one of them is the id of a node x reaches, so any
//       interleaving of reads and writes converges to the same fixed point; along the run of nodes a sequence created
//       the jump pointers double their reach per round.  (The nodes of the block's first sequence form the chain
//       0 -> 1 -> ... and everything else has a greater id: their first() is their own id -- `n_static` of them are skipped.)
//   P3  all threads: nodes and stack demand per root (atomic counts), two exclusive sums: where a root's piece of the
//       order and its stack begin.
//   P4  every thread walks ITS roots (t, t + T, ...): same pushes, same pops, same emissions as the sequential walk from
//       that root -- most roots are a single backbone node whose tails are done: one record, one emission.
//   P5  all threads: rank[order[r]] = r.
// The per-node word (first | state << 28; state = mark | ignored << 2) lives in the workgroup's LDS when the graph fits
// (the sweep's rows and windows are dead between two alignments), else in the slot's scratch.  64 x 5 kbp: the one-lane
// walk of round 5 took ~7 ms per re-sort (1.25 x the kernel time for S7'); this one is bounded by the block's longest
// piece -- the structural variant's branch, a few hundred nodes.
#if defined(__HIP_DEVICE_COMPILE__)
#define SXG_WAVE_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define SXG_WAVE_PRIO(p) ((void)0)
#endif
#if defined(__HIPCC__)
SXG_HD __attribute__((address_space(3))) int32_t* sxg_as_i32(__attribute__((address_space(3))) uint8_t* p) { return (__attribute__((address_space(3))) int32_t*)p; }
#endif
SXG_HD int32_t* sxg_as_i32(uint8_t* p) { return (int32_t*)p; }
constexpr int SPOA_REC = 16;   // int32 per record: n_in | n_aligned << 8 | n_member_tails << 16 (255: too many), tails[3], aligned[4], member tails[8]
constexpr int SPOA_FMASK = 0x0fffffff;   // first() of a node in its state word; the state above it
struct alignas(16) SpoaHalf { int x, y, z, w; };

// P4: the depth-first walk from root s0 (first(s0) == s0), emitting into ord[0 ..) and using stk[0 ..) -- both private to the
// root.  W: the per-node words.  Returns the number of nodes emitted.
template <class WP>
SXG_HD_PHASE int spoa_walk_root(const GraphView& G, WP W, const int s0, SXG_GP int32_t* const ord, SXG_GP int32_t* const stk,
                          const SpoaHalf ra, const SpoaHalf rb) {
    SXG_GP const SpoaHalf* const rec = (SXG_GP const SpoaHalf*)G.dfs_rec;
    // state of node x as this walk sees it: finished by an earlier root = done (2); else its own bits (mark | ignored << 2)
    // (bit 3 of the state: the node has no aligned nodes to emit behind it -- what coming back to it needs to know)
    auto stt = [&](const int x) -> int { const int w = (int)W[x]; return (w & SPOA_FMASK) < s0 ? 2 : (w >> 28) & 15; };
    auto set = [&](const int x, const int st) { W[x] = s0 | (st << 28); };   // (x belongs to this root: first(x) == s0)
    int w = 0, sp = 0;
    auto push = [&](const int x) { stk[sp++] = x; };
    push(s0);
    int curr = s0;
    bool fresh = true, at_root = true;   // fresh: curr is the node on top of the stack (just pushed)
    while (sp > 0) {
        if (!fresh) curr = stk[sp - 1];
        bool valid = true;
        int last = curr;
        const int sc = at_root ? 0 : stt(curr);   // (a root is entered with state 0)
        if ((sc & 3) == 1) {
            // Back on top of the stack: whatever this node pushed was popped above it, i.e. finished (a node is only popped when it
            // is), and what it did not push was finished before -- it is valid without looking at its tails again (the sequential
            // walk re-reads them and finds the same).  Only its aligned-node list has to be read again, when it has one.
            set(curr, (sc & 4) | 2);
            if (!(sc & 4)) {
                ord[w++] = curr;
                if (!(sc & 8)) {
                    const SpoaHalf a = rec[4 * (size_t)curr], b = rec[4 * (size_t)curr + 1];
                    const int na = (a.x >> 8) & 0xff;
                    const int al[4] = {b.x, b.y, b.z, b.w};
                    for (int q = 0; q < na; ++q) ord[w++] = al[q];
                }
            }
        } else if ((sc & 3) != 2) {
            SpoaHalf a = ra, b = rb;
            if (!at_root) { a = rec[4 * (size_t)curr]; b = rec[4 * (size_t)curr + 1]; }
            const bool ign = (sc & 4) != 0;
            const int ni = a.x & 0xff, na = ign ? 0 : (a.x >> 8) & 0xff, nmt = (a.x >> 16) & 0xff;
            const int al[4] = {b.x, b.y, b.z, b.w};
            if (ni <= 3) {
                const int m0 = ni > 0 ? stt(a.y) : 2, m1 = ni > 1 ? stt(a.z) : 2, m2 = ni > 2 ? stt(a.w) : 2;
                int ma[4];
                for (int q = 0; q < 4; ++q) ma[q] = q < na ? stt(al[q]) : 2;
                if (na > 0 && nmt <= 8 && (m0 & 3) == 2 && (m1 & 3) == 2 && (m2 & 3) == 2) {
                    // the whole group at once: my tails are done -- are those of my aligned nodes?
                    const SpoaHalf mc = rec[4 * (size_t)curr + 2], md = rec[4 * (size_t)curr + 3];
                    const int mt[8] = {mc.x, mc.y, mc.z, mc.w, md.x, md.y, md.z, md.w};
                    bool all = true;
                    for (int q = 0; q < 8; ++q) if (q < nmt && (stt(mt[q]) & 3) != 2) all = false;
                    if (all) {
                        for (int q = 0; q < 4; ++q) if (q < na && (ma[q] & 3) != 2) { set(al[q], ma[q] | 4 | 2); ma[q] = 2; }
                    }
                }
                if ((m0 & 3) != 2) { push(a.y); last = a.y; valid = false; }
                if ((m1 & 3) != 2) { push(a.z); last = a.z; valid = false; }
                if ((m2 & 3) != 2) { push(a.w); last = a.w; valid = false; }
                for (int q = 0; q < 4; ++q)
                    if ((ma[q] & 3) != 2) { push(al[q]); set(al[q], ma[q] | 4); last = al[q]; valid = false; }
            } else {
                for (int e = G.in_head[curr]; e >= 0; e = G.e_next_in[e]) {
                    const int tl = G.e_tail[e];
                    if ((stt(tl) & 3) != 2) { push(tl); last = tl; valid = false; }
                }
                for (int q = 0; q < na; ++q) {
                    const int m = stt(al[q]);
                    if ((m & 3) != 2) { push(al[q]); set(al[q], m | 4); last = al[q]; valid = false; }
                }
            }
            if (valid) {
                set(curr, (sc & 4) | 2);
                if (!ign) {
                    ord[w++] = curr;
                    for (int q = 0; q < na; ++q) ord[w++] = al[q];
                }
            } else set(curr, (sc & 4) | 1 | (na == 0 ? 8 : 0));
        }
        at_root = false;
        if (valid) { --sp; fresh = false; } else { curr = last; fresh = true; }
    }
    return w;
}

#if defined(SXG_RESORT_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define SXG_RSP(k) do { if (t == 0 && rprof) { const unsigned long long now_ = (unsigned long long)clock64(); rprof[k] += now_ - rt0_; rt0_ = now_; } } while (0)
#define SXG_RSP_DECL unsigned long long rt0_ = (unsigned long long)clock64()
#else
#define SXG_RSP(k) ((void)0)
#define SXG_RSP_DECL ((void)0)
#endif
// One record (see above) of node v, from the graph's lists.  The loads are laid out in stages -- (first in-edge, group) ->
// (its tail and successor, the group's members) -> (the other member's in-list) -> ... -- so that a record is four round trips,
// not the ten of "own list, then the aligned-node list, then every aligned node's list"; groups of one and two (all but a
// handful) take their aligned-node list from the members at hand.
SXG_HD void spoa_build_record(const GraphView& G, const int v) {
    const int e0 = G.in_head[v], ld = G.leader[v];
    int m[5];
#pragma unroll
    for (int x = 0; x < 5; ++x) m[x] = G.gmem[5 * ld + x];
    int t0 = -1, e1 = -1;
    if (e0 >= 0) { t0 = G.e_tail[e0]; e1 = G.e_next_in[e0]; }
    int nm = 0, other = -1;
#pragma unroll
    for (int x = 0; x < 5; ++x) if (m[x] >= 0) { ++nm; if (m[x] != v) other = m[x]; }
    int al[5] = {-1, -1, -1, -1, -1};
    int na;
    if (nm <= 2) { na = nm - 1; al[0] = nm == 2 ? other : -1; }
    else na = spoa_aligned_list(G, v, al);
    const int oe0 = na > 0 ? G.in_head[al[0]] : -1;
    int tl[3] = {-1, -1, -1}, ni = 0;
    if (e0 >= 0) {
        tl[0] = t0; ni = 1;
        for (int e = e1; e >= 0; e = G.e_next_in[e]) { if (ni < 3) tl[ni] = G.e_tail[e]; ++ni; }
    }
    int mt[8] = {-1, -1, -1, -1, -1, -1, -1, -1}, nmt = 0;
    for (int q = 0; q < na; ++q)
        for (int e = q == 0 ? oe0 : G.in_head[al[q]]; e >= 0; e = G.e_next_in[e]) { if (nmt < 8) mt[nmt] = G.e_tail[e]; ++nmt; }
    SXG_GP int32_t* const r = G.dfs_rec + (size_t)SPOA_REC * v;
    r[0] = (ni < 255 ? ni : 255) | (na << 8) | ((nmt <= 8 ? nmt : 255) << 16);
    r[1] = tl[0]; r[2] = tl[1]; r[3] = tl[2];
    r[4] = al[0]; r[5] = al[1]; r[6] = al[2]; r[7] = al[3];
    for (int q = 0; q < 8; ++q) r[8 + q] = mt[q];
}

// Round 6, second step: ONLY WHAT THE ALIGNMENT TOUCHED IS SORTED AGAIN.  Inside a block the graph only grows, so first() of a
// node never rises, and the piece of the order that root s's walk produces is a function of (a) the records of the nodes it
// finishes and (b) which of their tails and aligned nodes belong to s and which to smaller roots.  A piece is therefore the
// same as after the previous re-sort when
//     no node of it is new, none has a new in-edge, a new aligned node or an aligned node with a new in-edge ("touched": their
//     records are the only ones rebuilt), every node of it belonged to s before, and s finishes as many nodes as before
// (a node that left went to a smaller root, which fails the second test there).  Such a piece is not walked: node x goes to
// base(s) + (its previous spoa rank - the smallest previous rank of the piece) -- pieces are contiguous in the order.  What
// is kept from re-sort to re-sort: the records, the spoa rank and first() of every node, the size of every root's piece.
// After AddAlignment of a sequence of `len` letters G.target / G.kind / G.nexta still say, per letter, the node it went to,
// whether that node is new and whether the edge into it is new; `n_prev` nodes existed before it.  `full`: nothing is kept
// (the block's first re-sort).
template <class Ctx, class WP>
SXG_HD void spoa_resort_par(Ctx& c, const GraphView& G, WP W, const int n, const int n_static, const int len, const int n_prev_,
                            const bool full, unsigned long long* rprof) {
    const int T = c.nthreads(), t = c.tid();
    SXG_RSP_DECL;
    const int n_prev = full ? 0 : n_prev_;
    const int BIG = 0x3fffffff;
    SXG_GP int32_t* const rec = G.dfs_rec;
    SXG_GP int32_t* const touched = G.posnode;   // [n] the node's record has to be rebuilt (free until the next alignment starts)
    SXG_GP int32_t* const jump = G.nexta;
    SXG_GP int32_t* const cnt = G.preva;      // nodes a root finishes -> where its piece of the order begins
    SXG_GP int32_t* const sdem = G.slotadd;   // stack entries a root can need -> where its stack begins
    SXG_GP int32_t* const pmin = G.target;    // smallest previous spoa rank of a root's piece
    SXG_GP int8_t* const walk = G.kind;       // the root's piece has to be walked again
    // ---- P1: records of the touched nodes (listed first: rebuilt where they stand in the loop over all nodes, a wave would run
    //      one record's chain of dependent loads after the other for nearly every node it holds)
    constexpr int GB = Ctx::GB;   // (nodes per thread and step, every stage of dependent loads issued for all of them before the next)
    SXG_GP int32_t* const list = G.order_tmp;
    for (int v = t; v < n; v += T) touched[v] = v >= n_prev ? 1 : 0;
    if (t == 0) { sdem[n] = 0; sdem[n + 1] = 0; sdem[n + 2] = 0; }   // (the walks' stacks are handed out from [n]; [n + 1]: touched nodes; [n + 2]: roots to walk)
    c.sync();
    if (!full) {
        for (int i0 = t; i0 < len; i0 += GB * T) {
            int kd[GB], nw[GB], tg[GB];
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const int i = i0 + u * T < len ? i0 + u * T : len - 1;
                kd[u] = G.kind[i]; nw[u] = i > 0 ? G.nexta[i] : 0; tg[u] = G.target[i];
            }
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                if (i0 + u * T >= len || (kd[u] == 0 && nw[u] == 0)) continue;
                const int ld = G.leader[tg[u]];   // (a sequence passes through a group once: no two letters list the same nodes)
                for (int x = 0; x < 5; ++x) {
                    const int mbr = G.gmem[5 * ld + x];
                    if (mbr >= 0) { touched[mbr] = 1; list[c.atomic_add((int32_t*)(sdem + n + 1), 1)] = mbr; }
                }
            }
        }
        c.sync();
        const int nt = sdem[n + 1];
        // (an alignment that added neither a node nor an edge -- a sequence some earlier one spelt -- leaves the order as it is:
        //  AddAlignment's own bookkeeping moved nothing either)
        if (nt == 0 && n == n_prev) return;
        for (int k = t; k < nt; k += T) spoa_build_record(G, list[k]);
    } else {
        for (int v = t; v < n; v += T) spoa_build_record(G, v);
    }
    for (int v0 = t; v0 < n; v0 += GB * T) {
        int eo[GB], f0[GB], hd[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int v = v0 + u * T < n ? v0 + u * T : n - 1;
            eo[u] = G.out_head[v];
            f0[u] = v >= n_prev ? v : G.sp_first[v];   // (still the id of a node v reaches)
        }
#pragma unroll
        for (int u = 0; u < GB; ++u) { const int v = v0 + u * T < n ? v0 + u * T : n - 1; hd[u] = eo[u] >= 0 ? G.e_head[eo[u]] : v; }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int v = v0 + u * T;
            if (v >= n) continue;
            W[v] = f0[u];
            jump[v] = hd[u];
            cnt[v] = 0; sdem[v] = 0; pmin[v] = BIG; walk[v] = full ? 1 : 0;
        }
    }
    c.sync();
    SXG_RSP(0);
    // ---- P2: first()
    for (int round = 0; round < n + 2; ++round) {
        int changed = 0;
        for (int x0 = n_static + t; x0 < n; x0 += GB * T) {
            int old[GB], eo[GB], r0[GB], j[GB], h0[GB], ne[GB], jj[GB], v[GB];
            SpoaHalf al[GB];
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const int x = x0 + u * T < n ? x0 + u * T : n - 1;
                old[u] = (int)W[x]; eo[u] = G.out_head[x]; r0[u] = rec[(size_t)SPOA_REC * x]; j[u] = jump[x];
            }
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const int x = x0 + u * T < n ? x0 + u * T : n - 1;
                h0[u] = eo[u] >= 0 ? G.e_head[eo[u]] : x;
                ne[u] = eo[u] >= 0 ? G.e_next_out[eo[u]] : -1;
                jj[u] = jump[j[u]];
                if ((r0[u] >> 8) & 0xff) al[u] = ((SXG_GP const SpoaHalf*)rec)[4 * (size_t)x + 1];
                else al[u] = SpoaHalf{x, x, x, x};
            }
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const int x = x0 + u * T < n ? x0 + u * T : n - 1;
                const int na = (r0[u] >> 8) & 0xff;
                int m = old[u];
                { const int f = (int)W[h0[u]]; m = f < m ? f : m; }
                { const int f = (int)W[j[u]]; m = f < m ? f : m; }
                { const int f = (int)W[na > 0 ? al[u].x : x]; m = f < m ? f : m; }
                { const int f = (int)W[na > 1 ? al[u].y : x]; m = f < m ? f : m; }
                { const int f = (int)W[na > 2 ? al[u].z : x]; m = f < m ? f : m; }
                { const int f = (int)W[na > 3 ? al[u].w : x]; m = f < m ? f : m; }
                v[u] = m;
            }
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const int x = x0 + u * T;
                if (x >= n) continue;
                int m = v[u];
                for (int e = ne[u]; e >= 0; e = G.e_next_out[e]) { const int f = (int)W[G.e_head[e]]; m = f < m ? f : m; }
                jump[x] = jj[u];
                if (m < old[u]) { W[x] = m; changed = 1; }
            }
        }
        changed = c.reduce_max(changed);   // (also the barrier between the rounds)
#if defined(SXG_RESORT_PROF) && defined(__HIP_DEVICE_COMPILE__)
        if (t == 0 && rprof) rprof[5] += 1;
#endif
        if (!changed) break;
    }
    SXG_RSP(1);
    // ---- P3: the roots' pieces
    for (int x0 = t; x0 < n; x0 += GB * T) {
        int s[GB], idg[GB], tch[GB], spf[GB], spr[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int x = x0 + u * T < n ? x0 + u * T : n - 1;
            s[u] = (int)W[x]; idg[u] = G.in_deg[x]; tch[u] = touched[x];
            spf[u] = full ? 0 : G.sp_first[x]; spr[u] = full ? 0 : G.sp_rank[x];
        }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            if (x0 + u * T >= n) continue;
            c.atomic_add((int32_t*)(cnt + s[u]), 1);
            c.atomic_add((int32_t*)(sdem + s[u]), idg[u] + 6);
            if (!full) {
                if (tch[u] || spf[u] != s[u]) walk[s[u]] = 1;
                else c.atomic_min((int32_t*)(pmin + s[u]), spr[u]);
            }
        }
    }
    c.sync();
    for (int s0 = t; s0 < n; s0 += GB * T) {   // (the roots whose pieces are walked again, listed: a handful of several thousand)
        int k[GB], ko[GB], wk[GB], ws[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int s = s0 + u * T < n ? s0 + u * T : n - 1;
            k[u] = cnt[s]; ko[u] = full ? 0 : G.sp_cnt[s]; wk[u] = walk[s]; ws[u] = (int)W[s];
        }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int s = s0 + u * T;
            if (s >= n) continue;
            const bool again = wk[u] != 0 || (!full && k[u] != ko[u]);
            if (again && !wk[u]) walk[s] = 1;
            G.sp_cnt[s] = k[u];
            if (again && ws[u] == s) list[c.atomic_add((int32_t*)(sdem + n + 2), 1)] = s;
        }
    }
    array_excl_sum(c, n, [&](int s) { return cnt[s]; }, cnt);
    c.sync();
    SXG_RSP(2);
    // ---- P4: kept pieces are placed, the others walked
    if (!full) {
        for (int x0 = t; x0 < n; x0 += GB * T) {
            int s[GB], rk[GB], wk[GB], b0[GB], p0[GB];
#pragma unroll
            for (int u = 0; u < GB; ++u) { const int x = x0 + u * T < n ? x0 + u * T : n - 1; s[u] = (int)W[x] & SPOA_FMASK; rk[u] = G.sp_rank[x]; }
#pragma unroll
            for (int u = 0; u < GB; ++u) { wk[u] = walk[s[u]]; b0[u] = cnt[s[u]]; p0[u] = pmin[s[u]]; }
#pragma unroll
            for (int u = 0; u < GB; ++u) { const int x = x0 + u * T; if (x < n && !wk[u]) G.order[b0[u] + rk[u] - p0[u]] = x; }
        }
    }
    {
        SXG_GP const SpoaHalf* const rh = (SXG_GP const SpoaHalf*)rec;
        const int nw = sdem[n + 2];
        for (int k = t; k < nw; k += T) {
            const int v = list[k];
            const SpoaHalf ra = rh[4 * (size_t)v], rb = rh[4 * (size_t)v + 1];
            const int ob = cnt[v], sb = sdem[v];
            spoa_walk_root(G, W, v, G.order + ob, G.dfs_stack + c.atomic_add((int32_t*)(sdem + n), sb), ra, rb);
        }
    }
    c.sync();
    SXG_RSP(3);
    for (int r0 = t; r0 < n; r0 += GB * T) {
        int v[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) v[u] = r0 + u * T < n ? G.order[r0 + u * T] : 0;
#pragma unroll
        for (int u = 0; u < GB; ++u) { const int r = r0 + u * T; if (r < n) { G.rank[v[u]] = r; G.sp_rank[v[u]] = r; } }
    }
    for (int x = t; x < n; x += T) G.sp_first[x] = (int)W[x] & SPOA_FMASK;
    SXG_RSP(4);
}

// lst / lst_cap: bytes of the workgroup's LDS the re-sort may use (the per-node words); n_static: ids below it are the
// block's first sequence (a chain) -- 0 when the caller does not know; len, n_prev, full: see spoa_resort_par.
template <class Ctx, class LdsP>
SXG_HD_PHASE void spoa_resort(Ctx& c, const GraphView& G_, LdsP lst, const int lst_cap, const int n_static, const int len, const int n_prev,
                              const bool full, unsigned long long* rprof = nullptr) {
    const GraphView G = sxg_scalar_view(G_);
    const int n = *G.n_nodes;
    const int ns = n_static < n ? (n_static > 0 ? n_static : 0) : n;
    if (lst_cap >= 4 * (n + 1)) spoa_resort_par(c, G, sxg_as_i32(lst), n, ns, len, n_prev, full, rprof);
    else spoa_resort_par(c, G, G.newidx, n, ns, len, n_prev, full, rprof);
}

struct RowCaps {
    int rows_cap;   // rows of the traceback plane
    int pool_slots; // row-pool slots
    int step_cap;   // fold steps of the step-mask plane (sum over multi-pred rows of np-1)
    int lds_rows;   // packed sweep: stored rows the workgroup can keep in LDS at a time (0: every stored row goes to the ring)
};

// Second half of row preparation, shared by the block kernel (graph -> rows) and the
// stand-alone align kernel (caller CSR -> rows).  On entry R.flags holds STORE/SINK bits and
// R.slot[r] the rank of the last reader of row r.  Assigns ring slots of the row pool and
// the step-mask plane offset of multi-pred rows.  Returns a status (same on every thread).
// `hinted`: 1 = packed sweep -- R.tbx already holds the band hint of every row and there is no step-mask plane;
// 2 = banded sweep -- additionally there is no row ring (every row keeps its band in the plane) and the descriptor
// carries the hints of the first two predecessors where the ring slots would be;
// 3 = banded sweep with the ADAPTIVE band (decree B4) -- R.tbx holds remain() of every row; the bands follow from the
// sweep itself, which leaves every finished row's band and best-cell columns in words 6 and 7 of its descriptor.
template <class Ctx>
// (finish_rows keeps reading its view through the reference: with the scalar copy the 32-bit kernels trip the AMDGPU back-end
//  assertion on the shared-aperture null check that WgCtx's comment in poa_dp.hip.h describes)
SXG_HD_PHASE int finish_rows(Ctx& c, int N, const RowsView& R, const RowCaps& caps, const int hinted = 0) {
    const int T = c.nthreads(), t = c.tid();
    c.sync();
    const int n_store = array_excl_sum(c, N, [&](int r) { return (R.flags[r] & ROW_STORE) ? 1 : 0; }, R.sseq);
    if (t == 0) R.sseq[N] = n_store;
    c.sync();
    constexpr int GB = Ctx::GB;
    int worst = 0;
    for (int r0 = t; r0 < N; r0 += GB * T) {
        int fl[GB], lu[GB], sr[GB], sl[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int r = r0 + u * T;
            fl[u] = r < N ? (int)R.flags[r] : 0;
            lu[u] = r < N ? R.slot[r] : 0;
            sr[u] = r < N ? R.sseq[r] : 0;
        }
#pragma unroll
        for (int u = 0; u < GB; ++u) sl[u] = (fl[u] & ROW_STORE) ? R.sseq[lu[u]] : 0;
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int r = r0 + u * T;
            if (r >= N) continue;
            if (!(fl[u] & ROW_STORE)) { R.slot[r] = -1; continue; }
            const int cnt = sl[u] - sr[u];  // stored rows in [r, lu): must fit the ring
            if (cnt > worst) worst = cnt;
            // A row whose last reader comes before lds_rows more rows are stored never leaves the chip: stored rows take the
            // on-chip copies in turn (copy = stored-row number mod lds_rows), so the copy this row is written to at the end of
            // its sweep is next written by the stored row lds_rows later -- which is >= lu, and a row reads its predecessors
            // before it writes itself.  Slot -2 - copy; every word of a copy is written and read by the same lane.
            if (hinted == 1 && caps.lds_rows > 0 && cnt <= caps.lds_rows) R.slot[r] = -2 - sr[u] % caps.lds_rows;
            else R.slot[r] = sr[u] % caps.pool_slots;
        }
    }
    worst = c.reduce_max(worst);
    if (worst > caps.pool_slots && hinted < 2) return ST_POOL_OVERFLOW;
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
    constexpr int GBH = Ctx::GBH;   // (nine values per row live across three stages of loads)
    for (int r0 = t; r0 < N; r0 += GBH * T) {
        int pb[GBH], np[GBH], cf[GBH], p0[GBH], p1[GBH], q0[GBH], q1[GBH], ms[GBH], mt[GBH];
#pragma unroll
        for (int u = 0; u < GBH; ++u) {
            const int r = r0 + u * T;
            pb[u] = r < N ? R.pred_off[r] : 0;
            np[u] = r < N ? R.pred_off[r + 1] - pb[u] : 0;
            cf[u] = r < N ? ((int)R.code[r] << 16) | ((int)R.flags[r] << 24) : 0;
            ms[u] = r < N ? R.slot[r] : 0;
            mt[u] = r < N ? R.tbx[r] : 0;
        }
#pragma unroll
        for (int u = 0; u < GBH; ++u) {
            p0[u] = np[u] >= 1 ? R.preds[pb[u]] : 0;
            p1[u] = np[u] >= 2 ? R.preds[pb[u] + 1] : 0;
        }
#pragma unroll
        for (int u = 0; u < GBH; ++u) {
            const int r = r0 + u * T;
            if (hinted == 3) { q0[u] = 0; q1[u] = 0; }
            else if (hinted == 2) {
                q0[u] = p0[u] >= 1 ? R.tbx[p0[u] - 1] : 0;
                q1[u] = p1[u] >= 1 ? R.tbx[p1[u] - 1] : 0;
            } else {
                q0[u] = (p0[u] >= 1 && p0[u] != r) ? R.slot[p0[u] - 1] : -1;
                q1[u] = (p1[u] >= 1 && p1[u] != r) ? R.slot[p1[u] - 1] : -1;
            }
        }
#pragma unroll
        for (int u = 0; u < GBH; ++u) {
            const int r = r0 + u * T;
            if (r >= N) continue;
            int32_t* d = R.meta + 8 * (size_t)r;
            d[0] = pb[u]; d[1] = np[u] | cf[u]; d[2] = p0[u]; d[3] = q0[u]; d[4] = p1[u]; d[5] = q1[u]; d[6] = ms[u]; d[7] = mt[u];
        }
    }
    c.sync();
    return ST_OK;
}

// Decree B4: remain() of the node of every rank -- the number of edges of the walk that follows, from node to node, the
// heaviest out-edge (the first of greatest weight in out-list order) down to a node without out-edges.  A chain over
// ranks when done serially; here: every rank points to the rank of its heaviest successor, and pointer jumping doubles
// the distance covered per round (ceil(log2 N) rounds of N / T steps each; two pairs of scratch arrays -- free between
// two add_alignment calls -- are read and written alternately, so a round never reads what it writes).
template <class Ctx>
SXG_HD_PHASE void rows_remain(Ctx& c, const GraphView& G, int N, SXG_GP int32_t* out) {
    const int T = c.nthreads(), t = c.tid();
    SXG_GP int32_t* nx[2] = {G.preva, G.newidx};
    SXG_GP int32_t* ds[2] = {G.nexta, G.target};
    c.sync();
    constexpr int GB = Ctx::GB;
    for (int r0 = t; r0 < N; r0 += GB * T) {
        int e[GB], best[GB];
        uint32_t bw[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) { e[u] = r0 + u * T < N ? G.out_head[G.order[r0 + u * T]] : -1; best[u] = -1; bw[u] = 0; }
#pragma unroll
        for (int u = 0; u < GB; ++u)
            for (int x = e[u]; x >= 0; x = G.e_next_out[x]) {
                const uint32_t w = G.e_w[x];
                if (best[u] < 0 || w > bw[u]) { best[u] = G.e_head[x]; bw[u] = w; }
            }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int r = r0 + u * T;
            if (r >= N) continue;
            nx[0][r] = best[u] >= 0 ? G.rank[best[u]] : -1;
            ds[0][r] = best[u] >= 0 ? 1 : 0;
        }
    }
    c.sync();
    int cur = 0;
    for (int round = 0; round < 32; ++round) {
        int open = 0;
        for (int r0 = t; r0 < N; r0 += GB * T) {
            int n1[GB], d1[GB], n2[GB], d2[GB];
#pragma unroll
            for (int u = 0; u < GB; ++u) { const int r = r0 + u * T; n1[u] = r < N ? nx[cur][r] : -1; d1[u] = r < N ? ds[cur][r] : 0; }
#pragma unroll
            for (int u = 0; u < GB; ++u) { n2[u] = n1[u] >= 0 ? nx[cur][n1[u]] : -1; d2[u] = n1[u] >= 0 ? ds[cur][n1[u]] : 0; }
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const int r = r0 + u * T;
                if (r >= N) continue;
                nx[cur ^ 1][r] = n2[u];
                ds[cur ^ 1][r] = d1[u] + d2[u];
                open |= (int)(n2[u] >= 0);
            }
        }
        cur ^= 1;
        open = c.reduce_max(open);   // (also the barrier between the rounds)
        if (!open) break;
    }
    for (int r = t; r < N; r += T) out[r] = ds[cur][r];
    c.sync();
}

// Rank-space CSR + per-row DP metadata of the current graph.  Returns a status code
// (identical on every thread).
template <class Ctx>
SXG_HD_PHASE int prep_rows(Ctx& c, const GraphView& G_, const RowsView& R_, const RowCaps& caps, const int hinted = 0) {
    const GraphView G = sxg_scalar_view(G_);
    const RowsView R = sxg_scalar_view(R_);
    const int T = c.nthreads(), t = c.tid();
    c.sync();
    const int N = *G.n_nodes;
    if (N > caps.rows_cap) return ST_ROWS_OVERFLOW;
    // (GB rows per thread and iteration, every stage of dependent loads issued for all of them before the next: the chains
    // order -> in_head -> e_tail -> rank are four HBM round trips deep, and one wave per block has nothing else to hide them)
    constexpr int GB = Ctx::GB;
    for (int r0 = t; r0 < N; r0 += GB * T) {
        int v[GB], cd[GB], xp[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) v[u] = r0 + u * T < N ? G.order[r0 + u * T] : 0;
#pragma unroll
        for (int u = 0; u < GB; ++u) { cd[u] = G.code[v[u]]; xp[u] = hinted ? G.xpos[v[u]] : 0; }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int r = r0 + u * T;
            if (r >= N) continue;
            R.row_node[r] = v[u];
            R.code[r] = (uint8_t)cd[u];
            if (hinted && hinted != 3) R.tbx[r] = xp[u];
        }
    }
    if (hinted == 3) rows_remain(c, G, N, R.tbx);
    const int E = array_excl_sum(c, N, [&](int r) { return G.in_deg[G.order[r]]; }, R.pred_off);
    if (t == 0) R.pred_off[N] = E;
    c.sync();
    // preds in rank space; store / sink flags; last reader of every row (kept in R.slot)
    constexpr int GBH = Ctx::GBH;   // (eleven values per row)
    for (int r0 = t; r0 < N; r0 += GBH * T) {
        int v[GBH], o[GBH], ei[GBH], eo[GBH], od[GBH], ti[GBH], ho[GBH], ni[GBH], no[GBH], ri[GBH], ro[GBH];
#pragma unroll
        for (int u = 0; u < GBH; ++u) {
            const int r = r0 + u * T;
            v[u] = r < N ? G.order[r] : -1;
            o[u] = r < N ? R.pred_off[r] : 0;
        }
#pragma unroll
        for (int u = 0; u < GBH; ++u) {
            ei[u] = v[u] >= 0 ? G.in_head[v[u]] : -1;
            eo[u] = v[u] >= 0 ? G.out_head[v[u]] : -1;
            od[u] = v[u] >= 0 ? G.out_deg[v[u]] : 0;
        }
#pragma unroll
        for (int u = 0; u < GBH; ++u) {   // the first edge of either list (most nodes have one of each)
            ti[u] = ei[u] >= 0 ? G.e_tail[ei[u]] : -1;
            ni[u] = ei[u] >= 0 ? G.e_next_in[ei[u]] : -1;
            ho[u] = eo[u] >= 0 ? G.e_head[eo[u]] : -1;
            no[u] = eo[u] >= 0 ? G.e_next_out[eo[u]] : -1;
        }
#pragma unroll
        for (int u = 0; u < GBH; ++u) {
            ri[u] = ti[u] >= 0 ? G.rank[ti[u]] : -1;
            ro[u] = ho[u] >= 0 ? G.rank[ho[u]] : -1;
        }
#pragma unroll
        for (int u = 0; u < GBH; ++u) {
            const int r = r0 + u * T;
            if (r >= N) continue;
            int oo = o[u], regpred = 0;
            if (ei[u] >= 0) { R.preds[oo++] = ri[u] + 1; regpred |= (int)(ri[u] + 1 == r); }
            for (int e = ni[u]; e >= 0; e = G.e_next_in[e]) { const int pr = G.rank[G.e_tail[e]] + 1; R.preds[oo++] = pr; regpred |= (int)(pr == r); }
            int store = 0, lu = r;
            if (eo[u] >= 0) { if (ro[u] != r + 1) store = 1; if (ro[u] > lu) lu = ro[u]; }
            for (int e = no[u]; e >= 0; e = G.e_next_out[e]) {
                const int hr = G.rank[G.e_head[e]];
                if (hr != r + 1) store = 1;
                if (hr > lu) lu = hr;
            }
            R.flags[r] = (uint8_t)((store ? ROW_STORE : 0) | (od[u] == 0 ? ROW_SINK : 0) | (regpred ? ROW_REGPRED : 0));
            R.slot[r] = lu;
        }
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
