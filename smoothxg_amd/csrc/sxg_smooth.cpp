// sxg_smooth.cpp -- host side of the smoothing iteration around the blocked-POA engine
// (C ABI: include/sxg_smooth.h).  Plain C++17, no HIP: the POA is a callback.
//
// Follows, row by row:
//   A2  append_to_sequence            src/smooth.cpp:75-126
//   A3  collect / orient / dedup      src/smooth.cpp:676-743
//   A4  padding size                  src/smooth.cpp:1946-1970
//   A9  build_odgi_SPOA               src/smooth.cpp:2576-2654
//   A10 unchop, order, re-copy        src/smooth.cpp:935-1010
//   8f-1 lacing, validation, writer   src/main.cpp:599-1061
// odgi's unchop / topological_order / to_gfa are absent from the reference snapshot; they are
// restated by decree and mirrored line for line by oracle/smooth_oracle.py.
#include "../../include/sxg_smooth.h"

#include <omp.h>
#include <sys/mman.h>
#include <algorithm>
#include <thread>
#include <atomic>
#include <parallel/algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <queue>
#include <set>
#include <cmath>
#include <deque>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& m) { g_err = m; return code; }

typedef uint64_t handle_t;  // node index << 1 | is_reverse
inline handle_t mk(uint64_t n, bool rev) { return (n << 1) | (rev ? 1u : 0u); }
inline uint64_t nid(handle_t h) { return h >> 1; }
inline bool rev(handle_t h) { return h & 1; }
inline handle_t flip(handle_t h) { return h ^ 1; }

char comp(char c) {
    switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; default: return 'N'; }
}
std::string revcomp(const std::string& s) {
    std::string r(s.size(), 'N');
    for (size_t i = 0; i < s.size(); ++i) r[s.size() - 1 - i] = comp(s[i]);
    return r;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// input graph (what the path needs of XG: src/xg.hpp:271-370)
struct sxg_graph {
    std::vector<int64_t> ids;            // sorted node ids; index = rank
    std::vector<std::string> seq;        // by rank
    std::vector<std::string> pname;
    std::vector<std::vector<handle_t>> steps;
    std::vector<std::vector<uint64_t>> pos;  // bp offset of every step (+ total length at the end)
    // what block discovery needs on top (src/blocks.cpp): the L lines as oriented neighbour lists, the steps on
    // every node, and the offset of every node in the concatenated node sequences (XG's vectorised order)
    std::vector<std::vector<handle_t>> right_of, left_of;          // by node rank: follow_edges(n+, false / true)
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> on_node;  // (path, step) of every visit, path-major
    std::vector<uint64_t> vec_off;
    std::string sequence(handle_t h) const { return rev(h) ? revcomp(seq[nid(h)]) : seq[nid(h)]; }
    std::string path_sequence(size_t p) const {
        std::string s;
        for (handle_t h : steps[p]) s += sequence(h);
        return s;
    }
};

struct path_range_t { uint64_t path, begin, end, length; };  // steps [begin,end), length in bp (src/blocks.hpp:29-33)
struct sxg_blockset { std::vector<std::vector<path_range_t>> blocks; };

namespace {

// ---------------------------------------------------------------------------------------------
// A2: src/smooth.cpp:75-126.  Quirks kept: the left walk starts AT the range's first step and
// stops before step 0; every visited node contributes its LAST characters on both sides.
void append_to_sequence(const sxg_graph& g, uint64_t path, uint64_t starting_step, std::string& seq,
                        uint64_t& fwd_bp, uint64_t& rev_bp, int poa_padding, bool on_the_left) {
    uint64_t step = starting_step;
    const uint64_t final_step = on_the_left ? 0 : g.steps[path].size();
    uint64_t to_add = (uint64_t)poa_padding;
    std::string tmp;
    while (step != final_step && to_add > 0) {
        const handle_t h = g.steps[path][step];
        const std::string s = g.sequence(h);
        const uint64_t l = s.size();
        uint64_t added;
        if (l <= to_add) { tmp.append(s); added = l; }
        else { tmp.append(s.substr(s.size() - to_add)); added = to_add; }
        if (rev(h)) rev_bp += added; else fwd_bp += added;
        to_add -= added;
        step = on_the_left ? step - 1 : step + 1;
    }
    if (on_the_left) { seq.append(to_add, 'N'); seq.append(tmp); }
    else { seq.append(tmp); seq.append(to_add, 'N'); }
}

// A4: src/smooth.cpp:1946-1970 (float accumulation as in the reference)
int padding_size(const sxg_graph& g, const std::vector<path_range_t>& ranges, const sxg_smooth_params& p) {
    int poa_padding = 0;
    if (p.poa_padding_fraction > 0) {
        if (ranges.size() <= p.max_block_depth_for_padding_more) poa_padding = 311;
        float average_seq_len = 0.0f;
        for (auto& r : ranges)
            for (uint64_t s = r.begin; s != r.end; ++s) average_seq_len += (float)g.seq[nid(g.steps[r.path][s])].size();
        average_seq_len /= (float)ranges.size();
        poa_padding = std::max((int)(average_seq_len * p.poa_padding_fraction), poa_padding);
    }
    return poa_padding;
}

struct collected_t {
    int poa_padding = 0;
    std::vector<std::string> seqs;
    std::vector<uint32_t> weights;
    std::vector<std::vector<bool>> dup_is_revs;
    std::vector<std::vector<std::string>> dup_seq_names;
    std::vector<std::vector<uint64_t>> dup_rank_in_path_ranges;
    std::vector<std::string> all_names_in_original_order;
};

uint64_t xxh64(const void* data, uint64_t len, uint64_t seed);

// A3: src/smooth.cpp:676-743
collected_t collect(const sxg_graph& g, const std::vector<path_range_t>& ranges, const sxg_smooth_params& p) {
    collected_t c;
    if (ranges.empty()) return c;
    c.poa_padding = padding_size(g, ranges, p);
    std::unordered_map<uint64_t, uint64_t> seq_to_rank;
    for (uint64_t i = 0; i < ranges.size(); ++i) {
        const path_range_t& r = ranges[i];
        std::string seq;
        uint64_t fwd_bp = 0, rev_bp = 0;
        append_to_sequence(g, r.path, r.begin, seq, fwd_bp, rev_bp, c.poa_padding, true);
        for (uint64_t s = r.begin; s != r.end; ++s) {
            const handle_t h = g.steps[r.path][s];
            const std::string& ns = g.seq[nid(h)];
            if (rev(h)) { const size_t a0 = seq.size(); seq.resize(a0 + ns.size()); for (size_t y = 0; y < ns.size(); ++y) seq[a0 + y] = comp(ns[ns.size() - 1 - y]); rev_bp += ns.size(); }
            else { seq.append(ns); fwd_bp += ns.size(); }
        }
        append_to_sequence(g, r.path, r.end, seq, fwd_bp, rev_bp, c.poa_padding, false);
        const bool is_rev = rev_bp > fwd_bp;
        if (is_rev) seq = revcomp(seq);
        const std::string name = g.pname[r.path] + "_" + std::to_string(g.pos[r.path][r.begin]);
        const uint64_t hash = xxh64(seq.data(), seq.size(), 0);
        auto it = seq_to_rank.find(hash);
        if (it == seq_to_rank.end()) {
            seq_to_rank[hash] = c.seqs.size();
            c.seqs.push_back(std::move(seq));
            c.weights.push_back(1);
            c.dup_is_revs.push_back({is_rev});
            c.dup_seq_names.push_back({name});
            c.dup_rank_in_path_ranges.push_back({i});
        } else {
            const uint64_t rank = it->second;
            c.weights[rank] += 1;
            c.dup_is_revs[rank].push_back(is_rev);
            c.dup_seq_names[rank].push_back(name);
            c.dup_rank_in_path_ranges[rank].push_back(i);
        }
        c.all_names_in_original_order.push_back(name);
    }
    size_t mx = 0;
    for (auto& s : c.seqs) mx = std::max(mx, s.size());
    if (mx == 0) { collected_t e; e.poa_padding = c.poa_padding; return e; }  // "the graph would be empty"
    return c;
}

// XXH64 (published xxHash specification), the dedup key of src/smooth.cpp:716; same restatement as
// sxg_xxh64 in sxg_poa.hip, repeated here so that this library does not depend on the HIP one.
inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
uint64_t xxh64(const void* data, uint64_t len, uint64_t seed) {
    const uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL,
                   P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
    auto rd64 = [](const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; };
    auto rd32 = [](const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; };
    auto round = [&](uint64_t acc, uint64_t in) { return rotl64(acc + in * P2, 31) * P1; };
    auto merge = [&](uint64_t hh, uint64_t v) { return (hh ^ round(0, v)) * P1 + P4; };
    const uint8_t *p = (const uint8_t*)data, *end = p + len;
    uint64_t hh;
    if (len >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t* lim = end - 32;
        do { v1 = round(v1, rd64(p)); v2 = round(v2, rd64(p + 8)); v3 = round(v3, rd64(p + 16)); v4 = round(v4, rd64(p + 24)); p += 32; } while (p <= lim);
        hh = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        hh = merge(hh, v1); hh = merge(hh, v2); hh = merge(hh, v3); hh = merge(hh, v4);
    } else hh = seed + P5;
    hh += len;
    while (p + 8 <= end) { hh ^= round(0, rd64(p)); hh = rotl64(hh, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { hh ^= (uint64_t)rd32(p) * P1; hh = rotl64(hh, 23) * P2 + P3; p += 4; }
    while (p < end) { hh ^= (*p) * P5; hh = rotl64(hh, 11) * P1; ++p; }
    hh ^= hh >> 33; hh *= P2; hh ^= hh >> 29; hh *= P3; hh ^= hh >> 32;
    return hh;
}

inline uint8_t code_of(char ch) {
    switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}

// ---------------------------------------------------------------------------------------------
// output-side graph (odgi::graph_t's role): nodes, bidirected edges, named paths
struct edge_t {   // (a pair whose default constructor does not zero-fill: see uvec below)
    handle_t first, second;
    edge_t() {}
    edge_t(handle_t a, handle_t b) : first(a), second(b) {}
    bool operator<(const edge_t& o) const { return first < o.first || (first == o.first && second < o.second); }
    bool operator==(const edge_t& o) const { return first == o.first && second == o.second; }
};
// graphs above this many nodes take the OpenMP forms of unchop / sort / GFA text (SXG_SMOOTH_PAR_MIN: tests force them on small graphs)
inline size_t par_min() {
    static const size_t v = getenv("SXG_SMOOTH_PAR_MIN") ? (size_t)atoll(getenv("SXG_SMOOTH_PAR_MIN")) : 100000;
    return v;
}
// SXG_SMOOTH_TIMING=2: laps inside the phases of the laced graph (stderr)
inline void sublap(const char* what) {
    static const bool on = getenv("SXG_SMOOTH_TIMING") && atoi(getenv("SXG_SMOOTH_TIMING")) >= 2;
    static auto T0 = std::chrono::steady_clock::now();
    if (!on) return;
    const auto T1 = std::chrono::steady_clock::now();
    if (what) fprintf(stderr, "[sxg_smooth]   . %-28s %.3f s\n", what, std::chrono::duration<double>(T1 - T0).count());
    T0 = T1;
}
// Vectors of the laced graph (1e7-1e8 elements) are allocated WITHOUT the serial zero-fill of std::vector(n) and
// filled by the OpenMP loop that computes them: on the headline workload the fills and the sortedness checks
// were a third of unchop's time.
// Allocations of 8 MiB and more are 2 MiB-aligned and marked for transparent huge pages: the iteration touches
// ~10 GB of fresh memory on the headline workload, and 4 KiB first-touch faults (and their unmapping) are a
// measurable part of every phase.  free() releases either kind.
inline void* big_alloc(size_t bytes) {
    if (bytes < ((size_t)8 << 20)) return malloc(bytes ? bytes : 1);
    void* q = nullptr;
    if (posix_memalign(&q, (size_t)2 << 20, bytes)) return nullptr;
    static const bool thp = getenv("SXG_SMOOTH_NO_THP") == nullptr;   // (for A/B measurements)
    if (thp) madvise(q, bytes, MADV_HUGEPAGE);
    return q;
}
template <class T> struct noinit_alloc : std::allocator<T> {
    template <class U> struct rebind { typedef noinit_alloc<U> other; };
    T* allocate(size_t n) {
        void* q = big_alloc(n * sizeof(T));
        if (!q) throw std::bad_alloc();
        return (T*)q;
    }
    void deallocate(T* q, size_t) noexcept { free(q); }
    template <class U> void construct(U* q) noexcept { ::new ((void*)q) U; }   // default-init: no store for trivial U
    template <class U, class... A> void construct(U* q, A&&... a) { ::new ((void*)q) U(std::forward<A>(a)...); }
};
template <class T> using uvec = std::vector<T, noinit_alloc<T>>;
typedef uvec<handle_t> steps_t;   // the steps of an output path
template <class T> inline uvec<T> filled(size_t n, T v, bool par) {
    uvec<T> o(n);
#pragma omp parallel for schedule(static) if (par)
    for (int64_t i = 0; i < (int64_t)n; ++i) o[(size_t)i] = v;
    return o;
}
struct ograph_t {
    uvec<std::string> seq;                       // node i has id i+1
    uvec<edge_t> edges;                                 // canonical form, sorted, unique
    std::vector<std::pair<std::string, steps_t>> paths;
    static edge_t canon(handle_t a, handle_t b) {
        const edge_t x(a, b), y(flip(b), flip(a));
        return y < x ? y : x;
    }
    void sort_edges() {
        const int64_t ne = (int64_t)edges.size();
        const bool par = edges.size() > 2 * par_min();
        // one (parallel) pass: already strictly increasing = sorted and unique, the usual case
        int64_t unsorted = 0, dups = 0;
#pragma omp parallel for schedule(static) reduction(+ : unsorted, dups) if (par)
        for (int64_t x = 1; x < ne; ++x) {
            if (edges[(size_t)x] < edges[(size_t)x - 1]) ++unsorted;
            else if (edges[(size_t)x] == edges[(size_t)x - 1]) ++dups;
        }
        if (unsorted) {
            if (par) __gnu_parallel::sort(edges.begin(), edges.end());
            else std::sort(edges.begin(), edges.end());
        }
        if (unsorted || dups) edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
    }
};

// Edge set under construction.  The edges of a block graph are the consecutive step pairs of its paths
// (src/smooth.cpp:980-994): 64 paths x 5 kbp walk the same ~14 k edges 320 k times, so duplicates are
// filtered on insertion with a short per-handle list instead of a tree (the round-1 std::set) or a sort.
struct edge_acc_t {
    struct item_t { handle_t to; int32_t next; };
    std::vector<int32_t> head;
    std::vector<item_t> pool;
    explicit edge_acc_t(size_t n_nodes) : head(2 * n_nodes, -1) {}
    void add(handle_t a, handle_t b) {
        const edge_t e = ograph_t::canon(a, b);
        for (int32_t k = head[e.first]; k >= 0; k = pool[(size_t)k].next)
            if (pool[(size_t)k].to == e.second) return;
        pool.push_back(item_t{e.second, head[e.first]});
        head[e.first] = (int32_t)pool.size() - 1;
    }
    void into(uvec<edge_t>& out) const {
        out.clear();
        out.reserve(pool.size());
        for (size_t h = 0; h < head.size(); ++h) {
            const size_t first = out.size();
            for (int32_t k = head[h]; k >= 0; k = pool[(size_t)k].next) out.emplace_back((handle_t)h, pool[(size_t)k].to);
            std::sort(out.begin() + (long)first, out.end());
        }
    }
};

// unchop, by decree (odgi::algorithms::unchop is absent): merge u+ -> v+ when the right side of u
// has the single edge to v+, the left side of v the single edge from u+, u != v, and no path starts
// or ends inside the link.  Merged nodes are numbered by their chain head, in head order.
// Only the out-degree of every handle and its neighbour when that degree is 1 are needed; paths are
// rewritten in parallel (the laced graph of the headline workload walks 1.6e8 steps).
void unchop(ograph_t& G) {
    const size_t n = G.seq.size();
    const int64_t ne = (int64_t)G.edges.size();
    const bool par = n > par_min();   // (the laced graph of the headline workload: 1.4e7 nodes, 2e7 edges, 1.6e8 steps)
    if (par) sublap(nullptr);
    uvec<uint32_t> deg = filled<uint32_t>(2 * n, 0, par);
    uvec<handle_t> only = filled<handle_t>(2 * n, 0, par);
    if (par) sublap("unchop: alloc deg/only");
#pragma omp parallel for schedule(static) if (par)
    for (int64_t x = 0; x < ne; ++x) {
        const edge_t& e = G.edges[(size_t)x];
        const handle_t a = e.first, b = flip(e.second);
#pragma omp atomic
        deg[a]++;
#pragma omp atomic
        deg[b]++;
        // (`only` is read only where the degree is 1, i.e. where exactly one edge wrote it)
#pragma omp atomic write
        only[a] = e.second;
#pragma omp atomic write
        only[b] = flip(e.first);
    }
    if (par) sublap("unchop: degrees");
    uvec<char> start_at = filled<char>(2 * n, 0, par), end_at = filled<char>(2 * n, 0, par);
    for (auto& p : G.paths) {
        if (p.second.empty()) continue;
        start_at[p.second.front()] = 1; end_at[flip(p.second.front())] = 1;
        end_at[p.second.back()] = 1; start_at[flip(p.second.back())] = 1;
    }
    uvec<int64_t> next = filled<int64_t>(n, -1, par), prev = filled<int64_t>(n, -1, par);
    // (a node v is the target of at most one u: v's left side has the single edge from u+)
#pragma omp parallel for schedule(static) if (par)
    for (int64_t uu = 0; uu < (int64_t)n; ++uu) {
        const size_t u = (size_t)uu;
        const handle_t uf = mk(u, false);
        if (deg[uf] != 1) continue;
        const handle_t vf = only[uf];
        if (rev(vf) || nid(vf) == u) continue;
        const size_t v = nid(vf);
        if (deg[flip(vf)] != 1 || only[flip(vf)] != flip(uf)) continue;
        if (end_at[uf] || start_at[vf] || end_at[flip(vf)] || start_at[flip(uf)]) continue;
        next[u] = (int64_t)v; prev[v] = (int64_t)u;
    }
    // chains hang off their heads (no predecessor); what no head reaches is a pure cycle, broken at its smallest member
    if (par) sublap("unchop: links");
    uvec<int64_t> chain_of = filled<int64_t>(n, -1, par);
    uvec<size_t> heads;
    {
        uvec<char> reached = filled<char>(n, 0, par);
#pragma omp parallel for schedule(dynamic, 4096) if (par)
        for (int64_t uu = 0; uu < (int64_t)n; ++uu) {
            if (prev[(size_t)uu] >= 0) continue;
            for (size_t x = (size_t)uu;; x = (size_t)next[x]) { reached[x] = 1; if (next[x] < 0) break; }
        }
        for (size_t u = 0; u < n; ++u) {
            if (reached[u]) continue;
            size_t m = u;   // u is the smallest member: smaller ones would have been met (and marked) first
            for (size_t x = (size_t)next[u]; x != u; x = (size_t)next[x]) reached[x] = 1;
            reached[u] = 1;
            next[(size_t)prev[m]] = -1; prev[m] = -1;
        }
        // heads in id order: counted and written per chunk
        const int64_t CHK = 1 << 16, nch = ((int64_t)n + CHK - 1) / CHK;
        std::vector<size_t> cnt((size_t)nch + 1, 0);
#pragma omp parallel for schedule(static) if (par)
        for (int64_t q = 0; q < nch; ++q) {
            size_t k = 0;
            for (int64_t u = q * CHK; u < std::min((int64_t)n, (q + 1) * CHK); ++u) k += prev[(size_t)u] < 0 ? 1 : 0;
            cnt[(size_t)q + 1] = k;
        }
        for (int64_t q = 0; q < nch; ++q) cnt[(size_t)q + 1] += cnt[(size_t)q];
        heads.resize(cnt[(size_t)nch]);
#pragma omp parallel for schedule(static) if (par)
        for (int64_t q = 0; q < nch; ++q) {
            size_t w = cnt[(size_t)q];
            for (int64_t u = q * CHK; u < std::min((int64_t)n, (q + 1) * CHK); ++u) if (prev[(size_t)u] < 0) heads[w++] = (size_t)u;
        }
    }
    if (par) sublap("unchop: heads");
    const int64_t nc = (int64_t)heads.size();
    uvec<int64_t> first_of((size_t)nc), last_of((size_t)nc);
    uvec<std::string> nseq((size_t)nc);
#pragma omp parallel for schedule(dynamic, 4096) if (par)
    for (int64_t c = 0; c < nc; ++c) {
        const size_t u = heads[(size_t)c];
        size_t x = u, last = u;
        if (next[u] < 0) { chain_of[u] = c; nseq[(size_t)c].swap(G.seq[u]); }
        else {
            std::string s;
            while (true) { chain_of[x] = c; s += G.seq[x]; last = x; if (next[x] < 0) break; x = (size_t)next[x]; }
            nseq[(size_t)c].swap(s);
        }
        first_of[(size_t)c] = (int64_t)u; last_of[(size_t)c] = (int64_t)last;
    }
    if (par) sublap("unchop: chain sequences");
    auto map_handle = [&](handle_t h) { return mk((uint64_t)chain_of[nid(h)], rev(h)); };
    // surviving edges, mapped: counted and written per chunk so that the order of G.edges is kept
    uvec<edge_t> nedges;
    {
        const int64_t CHK = 1 << 16, nch = (ne + CHK - 1) / CHK;
        std::vector<size_t> cnt((size_t)nch + 1, 0);
        // the merged link u+ -> v+ in whichever canonical form it is stored: (u+, v+) or, when u > v, (v-, u-)
        // (round 4: the second form used to survive as a self loop of the merged node)
        auto interior = [&](const edge_t& e) {
            return (!rev(e.first) && !rev(e.second) && next[nid(e.first)] == (int64_t)nid(e.second)) ||
                   (rev(e.first) && rev(e.second) && next[nid(e.second)] == (int64_t)nid(e.first));
        };
#pragma omp parallel for schedule(static) if (par)
        for (int64_t q = 0; q < nch; ++q) {
            size_t k = 0;
            for (int64_t x = q * CHK; x < std::min(ne, (q + 1) * CHK); ++x) k += interior(G.edges[(size_t)x]) ? 0 : 1;
            cnt[(size_t)q + 1] = k;
        }
        for (int64_t q = 0; q < nch; ++q) cnt[(size_t)q + 1] += cnt[(size_t)q];
        nedges.resize(cnt[(size_t)nch]);
#pragma omp parallel for schedule(static) if (par)
        for (int64_t q = 0; q < nch; ++q) {
            size_t w = cnt[(size_t)q];
            for (int64_t x = q * CHK; x < std::min(ne, (q + 1) * CHK); ++x) {
                const edge_t& e = G.edges[(size_t)x];
                if (!interior(e)) nedges[w++] = ograph_t::canon(map_handle(e.first), map_handle(e.second));
            }
        }
    }
    if (par) sublap("unchop: edges");
    const int64_t np = (int64_t)G.paths.size();
#pragma omp parallel for schedule(dynamic, 1) if (par)
    for (int64_t q = 0; q < np; ++q) {
        auto& st = G.paths[(size_t)q].second;
        size_t w = 0;
        for (size_t r = 0; r < st.size(); ++r) {
            const handle_t h = st[r];
            const size_t x = nid(h);
            const int64_t c = chain_of[x];
            if (!rev(h)) { if (first_of[(size_t)c] == (int64_t)x) st[w++] = mk((uint64_t)c, false); }
            else { if (last_of[(size_t)c] == (int64_t)x) st[w++] = mk((uint64_t)c, true); }
        }
        st.resize(w);
    }
    if (par) sublap("unchop: paths");
    G.seq.swap(nseq);
    G.edges.swap(nedges);
    G.sort_edges();
    if (par) sublap("unchop: sort edges");
}

// topological order, by decree (odgi::algorithms::topological_order is absent): Kahn over the edges between
// forward nodes in their walking direction -- (a+, b+) is a -> b, the canonical form (a-, b-) is b -> a (round 4: those
// used to be left out, so the order was not topological whenever an edge ran from a higher to a lower id) --,
// smallest node first; leftovers (cycles) in id order.  Renumbers.
void topo_renumber(ograph_t& G) {
    const size_t n = G.seq.size();
    std::vector<uint32_t> off(n + 1, 0);
    std::vector<int> indeg(n, 0);
    auto fwd = [](const edge_t& e) { return rev(e.first) == rev(e.second) && nid(e.first) != nid(e.second); };
    auto tail = [](const edge_t& e) { return rev(e.first) ? nid(e.second) : nid(e.first); };
    auto head = [](const edge_t& e) { return rev(e.first) ? nid(e.first) : nid(e.second); };
    for (auto& e : G.edges) if (fwd(e)) { off[tail(e) + 1]++; indeg[head(e)]++; }
    for (size_t u = 0; u < n; ++u) off[u + 1] += off[u];
    std::vector<uint32_t> succ(off[n]), fill(off.begin(), off.end() - 1);
    for (auto& e : G.edges) if (fwd(e)) succ[fill[tail(e)]++] = (uint32_t)head(e);
    std::priority_queue<size_t, std::vector<size_t>, std::greater<size_t>> q;
    for (size_t u = 0; u < n; ++u) if (!indeg[u]) q.push(u);
    std::vector<int64_t> newid(n, -1);
    size_t k = 0;
    while (!q.empty()) {
        const size_t u = q.top(); q.pop();
        newid[u] = (int64_t)k++;
        for (uint32_t x = off[u]; x < off[u + 1]; ++x) if (--indeg[succ[x]] == 0) q.push(succ[x]);
    }
    for (size_t u = 0; u < n; ++u) if (newid[u] < 0) newid[u] = (int64_t)k++;
    uvec<std::string> nseq(n);
    for (size_t u = 0; u < n; ++u) nseq[(size_t)newid[u]].swap(G.seq[u]);
    for (auto& e : G.edges) e = ograph_t::canon(mk((uint64_t)newid[nid(e.first)], rev(e.first)), mk((uint64_t)newid[nid(e.second)], rev(e.second)));
    G.sort_edges();
    for (auto& p : G.paths) for (auto& h : p.second) h = mk((uint64_t)newid[nid(h)], rev(h));
    G.seq.swap(nseq);
}

// GFA1 text in the convention the in-tree XG::to_gfa shows (src/xg.cpp:1532-1580) minus its tags;
// S by id, L sorted by (from, to), P in path order.  odgi::to_gfa is absent: byte parity unpinned.
inline void put_u64(std::string& o, uint64_t v) {
    char buf[24];
    int k = 24;
    do { buf[--k] = (char)('0' + v % 10); v /= 10; } while (v);
    o.append(buf + k, (size_t)(24 - k));
}
// Pieces (node chunks, edge chunks, one per path) are formatted in parallel and copied in parallel into ONE
// malloc'ed buffer, which is what the C ABI hands out: the laced graph of the headline workload is 2.7 GB of
// text, and building it as one std::string plus a copy for the caller took 2.4 s of the iteration.
inline size_t digits_u64(uint64_t v) {
    size_t d = 1;
    while (v >= 10) { v /= 10; ++d; }
    return d;
}
inline char* put_u64_at(char* o, uint64_t v) {
    char buf[24];
    int k = 24;
    do { buf[--k] = (char)('0' + v % 10); v /= 10; } while (v);
    memcpy(o, buf + k, (size_t)(24 - k));
    return o + (24 - k);
}
// consume = true: the steps of a path are released by the thread that has just written its P line (the laced graph of
// the headline workload holds 1.3 GB of steps; unmapping them in 64 parallel pieces instead of one serial teardown)
char* to_gfa_c(const ograph_t& G, size_t* out_len, bool consume = false) {
    const size_t n = G.seq.size(), ne = G.edges.size(), np = G.paths.size();
    const size_t CH = 65536;
    const size_t nch = (n + CH - 1) / CH, ech = (ne + CH - 1) / CH;
    const size_t pieces = 1 + nch + ech + np;
    // pass 1: the exact size of every piece (counting digits touches no output memory)
    std::vector<size_t> off(pieces + 1, 0);
    const bool par = n > par_min();
    static const char head[] = "H\tVN:Z:1.0\n";
    off[1] = sizeof(head) - 1;
#pragma omp parallel for schedule(dynamic, 1) if (par)
    for (int64_t q = 1; q < (int64_t)pieces; ++q) {
        const size_t x = (size_t)q - 1;
        size_t bytes = 0;
        if (x < nch) {
            const size_t lo = x * CH, hi = std::min(n, lo + CH);
            for (size_t i = lo; i < hi; ++i) bytes += 4 + digits_u64(i + 1) + G.seq[i].size();          // "S\t" id "\t" seq "\n"
        } else if (x < nch + ech) {
            const size_t lo = (x - nch) * CH, hi = std::min(ne, lo + CH);
            for (size_t i = lo; i < hi; ++i) bytes += 11 + digits_u64(nid(G.edges[i].first) + 1) + digits_u64(nid(G.edges[i].second) + 1);
        } else {
            const auto& p = G.paths[x - nch - ech];
            bytes = 3 + p.first.size() + 3;                                                           // "P\t" name "\t" ... "\t*\n"
            for (size_t k = 0; k < p.second.size(); ++k) bytes += digits_u64(nid(p.second[k]) + 1) + 1 + (k ? 1 : 0);
        }
        off[(size_t)q + 1] = bytes;
    }
    for (size_t q = 0; q < pieces; ++q) off[q + 1] += off[q];
    char* buf = (char*)big_alloc(off.back() + 1);
    if (!buf) return nullptr;
    memcpy(buf, head, sizeof(head) - 1);
    // pass 2: every piece is written in place
#pragma omp parallel for schedule(dynamic, 1) if (par)
    for (int64_t q = 1; q < (int64_t)pieces; ++q) {
        const size_t x = (size_t)q - 1;
        char* o = buf + off[(size_t)q];
        if (x < nch) {
            const size_t lo = x * CH, hi = std::min(n, lo + CH);
            for (size_t i = lo; i < hi; ++i) {
                *o++ = 'S'; *o++ = '\t'; o = put_u64_at(o, i + 1); *o++ = '\t';
                memcpy(o, G.seq[i].data(), G.seq[i].size()); o += G.seq[i].size(); *o++ = '\n';
            }
        } else if (x < nch + ech) {
            const size_t lo = (x - nch) * CH, hi = std::min(ne, lo + CH);
            for (size_t i = lo; i < hi; ++i) {
                const edge_t& e = G.edges[i];
                *o++ = 'L'; *o++ = '\t'; o = put_u64_at(o, nid(e.first) + 1); *o++ = '\t'; *o++ = rev(e.first) ? '-' : '+'; *o++ = '\t';
                o = put_u64_at(o, nid(e.second) + 1); *o++ = '\t'; *o++ = rev(e.second) ? '-' : '+'; memcpy(o, "\t0M\n", 4); o += 4;
            }
        } else {
            const auto& p = G.paths[x - nch - ech];
            *o++ = 'P'; *o++ = '\t'; memcpy(o, p.first.data(), p.first.size()); o += p.first.size(); *o++ = '\t';
            for (size_t k = 0; k < p.second.size(); ++k) {
                if (k) *o++ = ',';
                o = put_u64_at(o, nid(p.second[k]) + 1);
                *o++ = rev(p.second[k]) ? '-' : '+';
            }
            memcpy(o, "\t*\n", 3); o += 3;
            if (consume) { steps_t none; none.swap(const_cast<steps_t&>(p.second)); }
        }
        if (o != buf + off[(size_t)q + 1]) abort();   // (size pass and write pass must agree)
    }
    buf[off.back()] = 0;
    if (out_len) *out_len = off.back();
    return buf;
}
std::string to_gfa(const ograph_t& G) {
    size_t len = 0;
    char* c = to_gfa_c(G, &len);
    std::string o(c ? c : "", c ? len : 0);
    free(c);
    return o;
}

// A9 + A10 for one block.  poa_* describe the block's POA graph (include/sxg_poa.h, block-local).
// abpoa: the consensus path keeps only nodes that some sequence path visits (build_odgi_abPOA, src/smooth.cpp:2542-2548;
// build_odgi_SPOA appends every consensus node, :2624-2627)
ograph_t build_block_graph(const collected_t& c, const uint8_t* node_code, int64_t n_nodes,
                           const std::vector<const int32_t*>& seq_paths, const int32_t* cons, int64_t n_cons,
                           const std::string& consensus_name, const bool abpoa = false) {
    static const char dec[5] = {'A', 'C', 'G', 'T', 'N'};
    ograph_t G;
    // A9 (src/smooth.cpp:2583-2637): one node per POA node; a path per duplicate name, padding steps
    // trimmed at both ends, reversed and flipped when the range was collected in reverse
    std::vector<std::pair<std::string, steps_t>> by_name;
    for (size_t i = 0; i < c.seqs.size(); ++i) {
        const int64_t len = (int64_t)c.seqs[i].size();
        for (size_t j = 0; j < c.dup_seq_names[i].size(); ++j) {
            steps_t st;
            st.reserve((size_t)std::max<int64_t>(0, len - 2 * (int64_t)c.poa_padding));
            for (int64_t k = c.poa_padding; k < len - c.poa_padding; ++k) st.push_back(mk((uint64_t)seq_paths[i][k], false));
            if (c.dup_is_revs[i][j]) { std::reverse(st.begin(), st.end()); for (auto& h : st) h = flip(h); }
            by_name.emplace_back(c.dup_seq_names[i][j], std::move(st));
        }
    }
    if (!consensus_name.empty()) {
        steps_t st;
        st.reserve((size_t)std::max<int64_t>(0, n_cons));
        std::vector<char> visited;
        if (abpoa) {
            visited.assign((size_t)n_nodes, 0);
            for (auto& pth : by_name) for (handle_t h : pth.second) visited[nid(h)] = 1;
        }
        for (int64_t k = 0; k < n_cons; ++k)
            if (!abpoa || visited[(size_t)cons[k]]) st.push_back(mk((uint64_t)cons[k], false));
        by_name.emplace_back(consensus_name, std::move(st));
    }
    // :2639-2653 drop nodes no path visits; A10 :980-994 keeps only path-supported edges, so the
    // edge set IS the set of consecutive step pairs
    std::vector<int64_t> keep(n_nodes, -1);
    for (auto& p : by_name) for (handle_t h : p.second) keep[nid(h)] = 0;
    for (int64_t v = 0, k = 0; v < n_nodes; ++v) if (keep[v] == 0) { keep[v] = k++; G.seq.push_back(std::string(1, dec[node_code[v] > 4 ? 4 : node_code[v]])); }
    edge_acc_t acc(G.seq.size());
    for (auto& p : by_name) {
        for (auto& h : p.second) h = mk((uint64_t)keep[nid(h)], rev(h));
        for (size_t k = 1; k < p.second.size(); ++k) acc.add(p.second[k - 1], p.second[k]);
    }
    acc.into(G.edges);
    // A10 :996-1010 paths in the order of the input block, consensus last
    std::map<std::string, size_t> idx;
    for (size_t k = 0; k < by_name.size(); ++k) idx[by_name[k].first] = k;
    // (a name occurs once per range -- path name + start position -- so every entry is taken exactly once;
    //  should two ranges ever share a name, the later ones copy)
    std::vector<char> taken(by_name.size(), 0);
    for (auto& nm : c.all_names_in_original_order) {
        const size_t k = idx[nm];
        if (!taken[k] && (consensus_name.empty() || k + 1 != by_name.size())) { G.paths.push_back(std::move(by_name[k])); taken[k] = 1; G.paths.back().first = nm; }
        else G.paths.push_back(taken[k] ? *std::find_if(G.paths.begin(), G.paths.end(), [&](const std::pair<std::string, steps_t>& q) { return q.first == nm; }) : by_name[k]);
    }
    if (!consensus_name.empty()) G.paths.push_back(std::move(by_name.back()));
    unchop(G);          // :935
    topo_renumber(G);   // :947
    return G;
}

struct batch_t {
    std::vector<int32_t> blk_off{0};
    std::vector<int64_t> seq_off{0};
    uvec<uint8_t> bases;   // (320 MB on the headline batch: no serial zero-fill)
    std::vector<uint32_t> weights;
};
void add_to_batch(batch_t& B, const collected_t& c) {
    for (size_t i = 0; i < c.seqs.size(); ++i) {
        for (char ch : c.seqs[i]) B.bases.push_back(code_of(ch));
        B.seq_off.push_back((int64_t)B.bases.size());
        B.weights.push_back(c.weights[i]);
    }
    B.blk_off.push_back((int32_t)(B.seq_off.size() - 1));
}
sxg_poa_params poa_params(const sxg_smooth_params& p) {  // src/smooth.cpp:2098-2106
    sxg_poa_params q;
    // (spoa's node order belongs to the spoa path: smooth_abpoa never calls spoa's sort, the `-A` path keeps the engine's own order)
    q.mode = (uint8_t)((p.local_alignment ? SXG_MODE_LOCAL : SXG_MODE_GLOBAL) | (p.poa_spoa_order && !p.use_abpoa ? SXG_ORDER_SPOA : 0)); q.banded = 0;
    if (!p.use_abpoa) {
        q.m = (int8_t)p.poa_m; q.n = (int8_t)-p.poa_n; q.g = (int8_t)-p.poa_g; q.e = (int8_t)-p.poa_e; q.q = (int8_t)-p.poa_q; q.c = (int8_t)-p.poa_c;
        return q;
    }
    // smooth_abpoa (src/smooth.cpp:2079-2090, 256-297): abPOA takes the CLI values as they are, a gap of k letters costs
    // min(o1 + k e1, o2 + k e2); gap_open1 = 0 selects its linear and gap_open2 = 0 its affine model.  The engine's
    // convention is spoa's -- the first letter of a gap costs g, every further one e -- so g = -(o + e), e = -e.
    // Banded alignment is always on (:2090), with abPOA's adaptive band.
    const int o1 = p.poa_g, e1 = p.poa_e, o2 = p.poa_q, e2 = p.poa_c;
    q.m = (int8_t)p.poa_m; q.n = (int8_t)-p.poa_n;
    if (o1 == 0) { q.g = q.e = q.q = q.c = (int8_t)-e1; }
    else {
        q.g = (int8_t)-(o1 + e1); q.e = (int8_t)-e1;
        if (o2 == 0) { q.q = q.g; q.c = q.e; } else { q.q = (int8_t)-(o2 + e2); q.c = (int8_t)-e2; }
    }
    // the band: always for global alignment; for local alignment unless the caller says that upstream abPOA runs its
    // local mode without one (sxg_smooth_params::abpoa_band_local, include/sxg_smooth.h)
    q.banded = (p.local_alignment && !p.abpoa_band_local) ? 0 : 2;
    return q;
}
// ---------------------------------------------------------------------------------------------
// A14: src/smooth.cpp:1972-2069.  The estimator is restated by decree (see include/sxg_smooth.h).
std::vector<uint64_t> canonical_kmers(const std::string& s, int k) {
    std::vector<uint64_t> v;
    if ((int)s.size() < k) return v;
    const uint64_t mask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    uint64_t fw = 0, rc = 0;
    int run = 0;
    for (char ch : s) {
        int c;
        switch (ch) { case 'A': case 'a': c = 0; break; case 'C': case 'c': c = 1; break;
                      case 'G': case 'g': c = 2; break; case 'T': case 't': c = 3; break; default: c = -1; }
        if (c < 0) { run = 0; fw = rc = 0; continue; }
        fw = ((fw << 2) | (uint64_t)c) & mask;
        rc = (rc >> 2) | ((uint64_t)(3 - c) << (2 * (k - 1)));
        if (++run >= k) v.push_back(fw < rc ? fw : rc);
    }
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    return v;
}
float mash_identity(const std::vector<uint64_t>& a, const std::vector<uint64_t>& b, int k) {
    size_t i = 0, j = 0, inter = 0;
    while (i < a.size() && j < b.size()) {
        if (a[i] < b[j]) ++i; else if (b[j] < a[i]) ++j; else { ++inter; ++i; ++j; }
    }
    const size_t uni = a.size() + b.size() - inter;
    double dist = 1.0;
    if (inter > 0 && uni > 0) {
        const double J = (double)inter / (double)uni;
        dist = -std::log(2.0 * J / (1.0 + J)) / (double)k;
    }
    return (float)(1.0 - dist);
}
// returns the number of sequences that took part; *thr is set when that is > 1
int identity_threshold(const sxg_graph& g, const std::vector<path_range_t>& ranges, int k, float* thr) {
    std::vector<std::vector<uint64_t>> km;
    for (auto& r : ranges) {
        std::string seq;  // :1985-1991, node sequences in step orientation, no padding
        for (uint64_t st = r.begin; st < r.end; ++st) seq += g.sequence(g.steps[r.path][st]);
        if (seq.size() >= (size_t)(8 * k)) km.push_back(canonical_kmers(seq, k));  // :1995
    }
    if (km.size() > 1) {
        std::vector<float> est;
        est.reserve(km.size() * (km.size() - 1) / 2);
        for (size_t i = 0; i < km.size(); ++i)
            for (size_t j = i + 1; j < km.size(); ++j) est.push_back(mash_identity(km[i], km[j], k));
        std::sort(est.begin(), est.end());
        *thr = std::max((float)0.7, est[(size_t)((double)(est.size() - 1) * 0.30)]);  // :2026
    }
    return (int)km.size();
}
void adaptive_scores(float thr, const int32_t in[6], int32_t out[6]) {  // :2032-2069
    static const int32_t tiers[5][6] = {{1, 19, 39, 3, 81, 1}, {1, 13, 31, 3, 51, 1}, {1, 9, 16, 2, 41, 1},
                                        {1, 7, 11, 2, 33, 1}, {1, 4, 6, 2, 26, 1}};
    static const double cut[5] = {0.99, 0.98, 0.97, 0.95, 0.90};
    const int32_t* pick = in;
    for (int t = 0; t < 5; ++t)
        if ((double)thr >= cut[t]) { pick = tiers[t]; break; }
    for (int x = 0; x < 6; ++x) out[x] = pick[x];
}
// the scores block `ranges` is aligned with (set/default, or its adaptive tier)
sxg_poa_params block_poa_params(const sxg_graph& g, const std::vector<path_range_t>& ranges, const sxg_smooth_params& p) {
    sxg_smooth_params q = p;
    if (p.adaptive_poa_params && ranges.size() > 1 && ranges.size() <= p.max_block_depth_for_padding_more) {  // :1982
        float thr = 0;
        if (identity_threshold(g, ranges, p.kmer_size > 0 ? p.kmer_size : 17, &thr) > 1) {
            const int32_t in[6] = {p.poa_m, p.poa_n, p.poa_g, p.poa_e, p.poa_q, p.poa_c};
            int32_t out[6];
            adaptive_scores(thr, in, out);
            q.poa_m = out[0]; q.poa_n = out[1]; q.poa_g = out[2]; q.poa_e = out[3]; q.poa_q = out[4]; q.poa_c = out[5];
        }
    }
    return poa_params(q);
}

// ---------------------------------------------------------------------------------------------
// MSA -> MAF rows: src/smooth.cpp:782-905
struct maf_row_t { std::string src; uint64_t start, size; bool rev; uint64_t src_size; std::string text; };

std::vector<maf_row_t> maf_rows_from_msa(const sxg_graph& g, const std::vector<path_range_t>& ranges, const collected_t& c,
                                         std::vector<std::string> msa, const std::string& consensus_name, size_t consensus_len) {
    std::vector<maf_row_t> rows;
    if (msa.empty()) return rows;
    const bool add_consensus = !consensus_name.empty();
    const size_t msa_l = msa[0].size();
    const uint64_t pad = (uint64_t)c.poa_padding;
    for (auto& r : msa) {  // :791-812 blank the padding
        size_t j = 0;
        for (uint64_t left = pad; left > 0; ++j)
            if (r[j] != '-') { r[j] = '-'; --left; }
        j = msa_l;
        for (uint64_t left = pad; left > 0;) {
            --j;
            if (r[j] != '-') { r[j] = '-'; --left; }
        }
    }
    auto col_has_letter = [&](size_t col) { for (auto& r : msa) if (r[col] != '-') return true; return false; };
    size_t b0 = 0;                       // :815-829
    while (b0 < msa_l && !col_has_letter(b0)) ++b0;
    int64_t e0 = (int64_t)msa_l - 1;     // :832-847
    while (e0 >= 0 && !col_has_letter((size_t)e0)) --e0;
    e0 += 1;
    const size_t num_seqs = msa.size();
    for (size_t rank = 0; rank < num_seqs; ++rank) {
        const bool is_cons = add_consensus && rank == num_seqs - 1;
        const size_t ndup = is_cons ? 1 : c.dup_rank_in_path_ranges[rank].size();  // :772-779 placeholder entry
        for (size_t x = 0; x < ndup; ++x) {
            maf_row_t row;
            if (!is_cons) {  // :861-880
                const path_range_t& pr = ranges[c.dup_rank_in_path_ranges[rank][x]];
                row.src = g.pname[pr.path];
                row.rev = c.dup_is_revs[rank][x];
                row.src_size = g.pos[pr.path].back();
                const uint64_t last = pr.end - 1;
                row.start = row.rev ? row.src_size - g.pos[pr.path][last] - g.seq[nid(g.steps[pr.path][last])].size() : g.pos[pr.path][pr.begin];
                row.size = c.seqs[rank].size() - 2 * pad;
            } else {         // :881-889
                row.src = consensus_name; row.rev = false; row.src_size = consensus_len - 2 * pad; row.start = 0; row.size = row.src_size;
            }
            row.text = (int64_t)b0 < e0 ? msa[rank].substr(b0, (size_t)e0 - b0) : std::string();
            rows.push_back(row);
        }
    }
    return rows;
}
std::string maf_rows_text(const std::vector<maf_row_t>& rows) {
    std::string o;
    for (auto& r : rows)
        o += r.src + "\t" + std::to_string(r.start) + "\t" + std::to_string(r.size) + "\t" + (r.rev ? "-" : "+") + "\t" + std::to_string(r.src_size) +
             "\t" + r.text + "\n";
    return o;
}
std::string maf_block_text(const std::vector<maf_row_t>& rows) {  // src/maf.hpp:35-66
    size_t w_src = 0, w_start = 0, w_size = 0, w_srcsize = 0;
    std::vector<std::string> order;
    for (auto& r : rows) {
        w_src = std::max(w_src, r.src.size());
        w_start = std::max(w_start, std::to_string(r.start).size());
        w_size = std::max(w_size, std::to_string(r.size).size());
        w_srcsize = std::max(w_srcsize, std::to_string(r.src_size).size());
        if (std::find(order.begin(), order.end(), r.src) == order.end()) order.push_back(r.src);
    }
    auto setw = [](const std::string& v, size_t w) { return (v.size() < w ? std::string(w - v.size(), ' ') : std::string()) + v; };
    std::string o;
    for (auto& src : order)
        for (auto& r : rows) {
            if (r.src != src) continue;
            o += "s " + r.src + std::string(w_src - r.src.size(), ' ') + setw(std::to_string(r.start), w_start + 1) +
                 setw(std::to_string(r.size), w_size + 1) + setw(r.rev ? "-" : "+", 2) + setw(std::to_string(r.src_size), w_srcsize + 1) + " " +
                 r.text + "\n";
        }
    return o + "\n";
}

// ---------------------------------------------------------------------------------------------
// A13 + 8f-4: MAF block merging, flip decision, flip rebuild (src/smooth.cpp:1091-1544, 1600-1919,
// 2352-2436).  The reference keeps rows in hash maps and walks them in hash order; here every map is
// in insertion order (first emission), by decree.  Mirrored by oracle/smooth_oracle.py.
struct maf_prow_t { uint64_t start, size; bool rev; uint64_t plen; std::string text; };
struct omap_t {   // insertion-ordered string -> rows
    std::vector<std::pair<std::string, std::vector<maf_prow_t>>> items;
    std::vector<maf_prow_t>* find(const std::string& k) { for (auto& it : items) if (it.first == k) return &it.second; return nullptr; }
    std::vector<maf_prow_t>& get(const std::string& k) { if (auto* f = find(k)) return *f; items.emplace_back(k, std::vector<maf_prow_t>()); return items.back().second; }
    bool empty() const { return items.empty(); }
};
struct maf_group_t { std::vector<uint64_t> block_ids; omap_t rows; std::deque<std::pair<std::string, maf_prow_t>> cons; };
struct merged_group_info_t { std::string ranges; bool inverted; std::vector<std::pair<uint64_t, uint64_t>> intervals; };
struct merge_state_t { std::vector<merged_group_info_t> groups; std::vector<char> in_merged; };

omap_t maf_block_map(const std::vector<maf_row_t>& rows) {   // what smooth_spoa hands to the writer thread (:893-905)
    omap_t m;
    for (auto& r : rows) m.get(r.src).push_back(maf_prow_t{r.start, r.size, r.rev, r.src_size, r.text});
    return m;
}
std::string revcomp_gapped(const std::string& t) {   // odgi::reverse_complement_in_place: '-' stays '-' (src/dna.cpp table)
    std::string r(t.size(), '-');
    for (size_t i = 0; i < t.size(); ++i) { const char c = t[t.size() - 1 - i]; r[i] = c == '-' ? '-' : comp(c); }
    return r;
}
std::string write_maf_rows(const omap_t& maf) {   // src/maf.hpp:35-66 over an ordered map
    size_t w_src = 0, w_start = 0, w_size = 0, w_ps = 0;
    for (auto& it : maf.items)
        for (auto& r : it.second) {
            w_src = std::max(w_src, it.first.size());
            w_start = std::max(w_start, std::to_string(r.start).size());
            w_size = std::max(w_size, std::to_string(r.size).size());
            w_ps = std::max(w_ps, std::to_string(r.plen).size());
        }
    auto setw = [](const std::string& v, size_t w) { return (v.size() < w ? std::string(w - v.size(), ' ') : std::string()) + v; };
    std::string o;
    for (auto& it : maf.items)
        for (auto& r : it.second)
            o += "s " + it.first + std::string(w_src - it.first.size(), ' ') + setw(std::to_string(r.start), w_start + 1) + setw(std::to_string(r.size), w_size + 1) +
                 setw(r.rev ? "-" : "+", 2) + setw(std::to_string(r.plen), w_ps + 1) + " " + r.text + "\n";
    return o + "\n";
}
// src/smooth.cpp:1091-1310
void put_block_in_group(maf_group_t& grp, uint64_t block_id, omap_t& maf, const std::string& consensus_name, bool on_the_left, bool flip) {
    size_t width = grp.block_ids.empty() ? 0 : grp.rows.items.front().second.front().text.size();
    std::string gaps(width, '-');
    for (auto& it : maf.items) {
        if (it.first == consensus_name) continue;
        std::vector<maf_prow_t>* have = grp.rows.find(it.first);
        if (!have) {
            std::vector<maf_prow_t> fresh;
            for (auto& r : it.second) {
                const uint64_t start = flip ? r.plen - (r.start + r.size) : r.start;
                if (flip) r.text = revcomp_gapped(r.text);
                fresh.push_back(maf_prow_t{start, r.size, (bool)(flip ^ r.rev), r.plen, on_the_left ? r.text + gaps : gaps + r.text});
            }
            grp.rows.get(it.first) = std::move(fresh);
        } else {
            std::vector<size_t> unmerged;
            for (size_t rk = 0; rk < it.second.size(); ++rk) {
                maf_prow_t& r = it.second[rk];
                const uint64_t start = flip ? r.plen - (r.start + r.size) : r.start;
                bool merged = false;
                for (auto& m : *have) {
                    if ((bool)(flip ^ r.rev) != m.rev || m.text.size() != width) continue;
                    if (m.rev) {
                        if (m.plen - m.start == r.plen - (start + r.size)) {            // new row on the left
                            m.start -= r.size;
                            if (flip) r.text = revcomp_gapped(r.text);
                            m.text = r.text + m.text; m.size += r.size; merged = true; break;
                        } else if (r.plen - start == m.plen - (m.start + m.size)) {     // new row on the right
                            if (flip) r.text = revcomp_gapped(r.text);
                            m.text += r.text; m.size += r.size; merged = true; break;
                        }
                    } else {
                        if (m.start + m.size == start) {                                 // new row on the right
                            if (flip) r.text = revcomp_gapped(r.text);
                            m.text += r.text; m.size += r.size; merged = true; break;
                        } else if (start + r.size == m.start) {                          // new row on the left
                            m.start -= r.size;
                            if (flip) r.text = revcomp_gapped(r.text);
                            m.text = r.text + m.text; m.size += r.size; merged = true; break;
                        }
                    }
                }
                if (!merged) unmerged.push_back(rk);
            }
            for (size_t rk : unmerged) {
                maf_prow_t& r = it.second[rk];
                const uint64_t start = flip ? r.plen - (r.start + r.size) : r.start;
                if (flip) r.text = revcomp_gapped(r.text);
                have->push_back(maf_prow_t{start, r.size, (bool)(flip ^ r.rev), r.plen, on_the_left ? r.text + gaps : gaps + r.text});
            }
        }
    }
    if (!consensus_name.empty()) {
        maf_prow_t& r = (*maf.find(consensus_name))[0];
        if (flip) r.text = revcomp_gapped(r.text);
        if (on_the_left) grp.cons.emplace_front(consensus_name, r); else grp.cons.emplace_back(consensus_name, r);
    }
    const size_t add = maf.items.front().second.front().text.size();
    width += add;
    gaps.assign(add, '-');
    for (auto& it : grp.rows.items)
        for (auto& m : it.second)
            if (m.text.size() < width) m.text = on_the_left ? gaps + m.text : m.text + gaps;
    if (on_the_left) grp.block_ids.insert(grp.block_ids.begin(), block_id); else grp.block_ids.push_back(block_id);
}
// src/smooth.cpp:1312-1544
std::string write_group(maf_group_t& grp, merge_state_t& st, bool add_consensus, const std::string& base, bool below, bool preserve_unmerged) {
    const auto& ids = grp.block_ids;
    const size_t n = ids.size();
    const uint64_t lo = std::min(ids.front(), ids.back()), hi = std::max(ids.front(), ids.back());
    std::string ranges = std::to_string(lo), full = std::to_string(ids.front());
    if (n > 1) {
        full.clear();
        ranges += "-" + std::to_string(hi);
        const bool inverted = ids.front() > ids.back();
        merged_group_info_t info;
        info.inverted = inverted;
        size_t begin = 0;
        if (add_consensus) st.in_merged[ids[0]] = 1;
        for (size_t i = 1; i < n; ++i) {
            const bool contiguous = inverted ? ids[i - 1] - ids[i] == 1 : ids[i] - ids[i - 1] == 1;
            if (!contiguous) {
                if (inverted) info.intervals.emplace_back(ids[i - 1], ids[begin] + 1); else info.intervals.emplace_back(ids[begin], ids[i - 1] + 1);
                full += std::to_string(ids[begin]);
                if ((i - 1) - begin > 0) full += "-" + std::to_string(ids[i - 1]);
                full += "_";
                begin = i;
            }
            if (add_consensus) st.in_merged[ids[i]] = 1;
        }
        if (inverted) info.intervals.emplace_back(ids[n - 1], ids[begin] + 1); else info.intervals.emplace_back(ids[begin], ids[n - 1] + 1);
        full += std::to_string(ids[begin]);
        if ((n - 1) - begin > 0) full += "-" + std::to_string(ids[n - 1]);
        info.ranges = ranges;
        st.groups.push_back(info);
    }
    bool loops = false;
    omap_t maf;
    for (auto& it : grp.rows.items) { if (it.second.size() > 1) loops = true; maf.get(it.first) = it.second; }
    if (add_consensus) {
        const size_t length = grp.rows.items.front().second.front().text.size();
        size_t pos0 = 0;
        uint64_t m_size = 0, m_plen = 0;
        std::string m_text;
        for (auto& c : grp.cons) {
            if (n == 1 || preserve_unmerged) {
                std::string gapped = std::string(pos0, '-') + c.second.text;
                if (gapped.size() < length) gapped.append(length - gapped.size(), '-');
                maf.get(c.first).push_back(maf_prow_t{c.second.start, c.second.size, c.second.rev, c.second.plen, gapped});
                pos0 += c.second.text.size();
            }
            if (n > 1) { m_size += c.second.size; m_plen += c.second.plen; m_text += c.second.text; }
        }
        if (n > 1) maf.get(base + ranges + " ").push_back(maf_prow_t{grp.cons.front().second.start, m_size, grp.cons.front().second.rev, m_plen, m_text});
    }
    std::string o = "a blocks=" + full + " loops=" + (loops ? "true" : "false");
    if (n > 1) { o += " merged=true"; if (below) o += " below_thresh=true"; }
    return o + "\n" + write_maf_rows(maf);
}
// the in-order MAF consumer of smooth_and_lace, src/smooth.cpp:1600-1919
std::string merge_maf_blocks(std::vector<omap_t>& block_mafs, const std::vector<char>& groom_flips, bool merge_blocks, double jaccard_min,
                             bool add_consensus, const std::string& base, size_t max_groups, bool preserve_unmerged, const char* header,
                             std::vector<char>& flips, merge_state_t& st) {
    const size_t nb = block_mafs.size();
    flips.assign(nb, 0);
    st.groups.clear();
    st.in_merged.assign(nb, 0);
    std::string out = header ? std::string(header) + "\n" : std::string();
    std::deque<maf_group_t> queue;
    for (size_t block_id = 0; block_id < nb; ++block_id) {
        omap_t& maf = block_mafs[block_id];
        if (maf.empty()) continue;   // (a block without sequences has no rows: skipped)
        const std::string cname = add_consensus ? base + std::to_string(block_id) : std::string();
        bool merged = false, below = false, flip_in = false;
        int64_t where = -1;
        int left_in = -1;
        if (merge_blocks) {
            if (queue.empty()) { queue.emplace_back(); where = 0; merged = true; }
            else {
                double best = -1;
                for (size_t gi = 0; gi < queue.size(); ++gi) {
                    maf_group_t& grp = queue[gi];
                    int on_left = grp.block_ids.size() > 1 ? (grp.block_ids.front() > grp.block_ids.back() ? 1 : 0) : -1;
                    for (int fl = 0; fl < 2; ++fl) {
                        const bool flip = fl != 0;
                        bool ok = true;
                        uint64_t ncont = 0;
                        for (auto& it : maf.items) {
                            if (it.first == cname) continue;
                            std::vector<maf_prow_t>* have = grp.rows.find(it.first);
                            if (!have) continue;
                            bool found = false;
                            for (auto& r : it.second) {
                                const uint64_t start = flip ? r.plen - (r.start + r.size) : r.start;
                                for (auto& m : *have) {
                                    if ((bool)(flip ^ r.rev) != m.rev) continue;
                                    if (flip ^ r.rev) {
                                        if (m.plen - m.start == r.plen - (start + r.size)) { if (on_left == -1 || on_left == 1) { on_left = 1; found = true; ++ncont; break; } }
                                        else if (r.plen - start == m.plen - (m.start + m.size)) { if (on_left == -1 || on_left == 0) { on_left = 0; found = true; ++ncont; break; } }
                                    } else {
                                        if (m.start + m.size == start) { if (on_left == -1 || on_left == 0) { on_left = 0; found = true; ++ncont; break; } }
                                        else if (start + r.size == m.start) { if (on_left == -1 || on_left == 1) { on_left = 1; found = true; ++ncont; break; } }
                                    }
                                }
                            }
                            if (!found) { ok = false; break; }
                        }
                        if (ok) {
                            uint64_t n_grp = 0, n_blk = 0;
                            for (auto& it : grp.rows.items) n_grp += it.second.size();
                            for (auto& it : maf.items) n_blk += it.second.size();
                            const double jac = (double)ncont / (double)(n_blk - (add_consensus ? 1 : 0) + n_grp - ncont);
                            if (jac >= jaccard_min && jac > best) { best = jac; flip_in = flip; where = (int64_t)gi; left_in = on_left; }
                        }
                    }
                }
                below = best > -1 && best < jaccard_min;
            }
            merged = where > -1;
        }
        if (merged) {
            put_block_in_group(queue[(size_t)where], block_id, maf, cname, left_in == 1, flip_in);
            if (flip_in) flips[block_id] = 1;
        } else {
            if (queue.size() >= max_groups) { out += write_group(queue.front(), st, add_consensus, base, below, preserve_unmerged); queue.pop_front(); }
            queue.emplace_back();
            put_block_in_group(queue.back(), block_id, maf, cname, false, groom_flips[block_id] != 0);
        }
        omap_t().items.swap(maf.items);
    }
    while (!queue.empty()) { out += write_group(queue.front(), st, add_consensus, base, false, preserve_unmerged); queue.pop_front(); }
    return out;
}
// src/smooth.cpp:2352-2436 -- by decree the INTENDED flip (the reference looks edge endpoints up in a table that
// only holds forward handles): ids kept, sequences reverse-complemented, every edge and path step toggles its
// orientation (paths keep spelling their sequence), the consensus keeps its handles in reversed order.
void flip_block_graph(ograph_t& G, const std::string& consensus_name) {
    for (auto& sq : G.seq) sq = revcomp(sq);
    for (auto& e : G.edges) e = ograph_t::canon(flip(e.first), flip(e.second));
    G.sort_edges();
    for (auto& p : G.paths) {
        if (!consensus_name.empty() && p.first == consensus_name) std::reverse(p.second.begin(), p.second.end());
        else for (auto& h : p.second) h = flip(h);
    }
}
// :1826-1842: is the first step of the block path that belongs to the lowest-ranked input path reversed?
bool groom_flip(const std::unordered_map<std::string, size_t>& rank_of, const ograph_t& G, const std::string& consensus_name) {
    size_t best = (size_t)-1;
    bool fl = false;
    for (auto& p : G.paths) {
        if ((!consensus_name.empty() && p.first == consensus_name) || p.second.empty()) continue;
        auto it = rank_of.find(p.first.substr(0, p.first.find_last_of('_')));
        if (it == rank_of.end()) continue;
        if (it->second < best) { best = it->second; fl = rev(p.second.front()); }
    }
    return fl;
}

std::string cons_name(const sxg_smooth_params& p, int64_t block_id) {
    if (!p.add_consensus) return "";
    return std::string(p.consensus_base_name ? p.consensus_base_name : "Consensus_") + std::to_string(block_id);
}
ograph_t block_graph_from_out(const collected_t& c, const batch_t& B, const sxg_poa_batch_out& out, int64_t slot, const std::string& cname,
                              const bool abpoa = false) {
    std::vector<const int32_t*> sp;
    for (int32_t s = B.blk_off[slot]; s < B.blk_off[slot + 1]; ++s) sp.push_back(out.seq_path_nodes + B.seq_off[s]);
    const int64_t n0 = out.node_off[slot], nn = out.node_off[slot + 1] - n0;
    const int32_t* cons = out.cons_nodes && out.cons_off ? out.cons_nodes + out.cons_off[slot] : nullptr;
    const int64_t nc = out.cons_nodes && out.cons_off ? out.cons_off[slot + 1] - out.cons_off[slot] : 0;
    return build_block_graph(c, out.node_code + n0, nn, sp, cons, nc, cname, abpoa);
}
// ---------------------------------------------------------------------------------------------
// Compact block graphs + the flat lacing path.
//
// A block's normalised graph (A9 + A10) as flat arrays -- node lengths, out-degrees (CSR), edge heads, one path of
// node ids per DEDUP'D sequence, the consensus path -- is what the GPU engine returns when it is asked for block graphs
// (sxg_poa_batch_in::want_block_graph, include/sxg_poa.h): the block's workgroup-sized scans do the trim, the
// path-supported edge filter, the 1-bp chain unchop and the renumbering on the device.  A provider that returns only raw
// POA results (the CPU oracle callback of the tests, a sharded provider) gets the same arrays built here, on the host,
// from build_block_graph's own steps.  Either way the iteration never materialises per-node strings or per-duplicate
// step vectors again: lacing, validation, the global unchop and the GFA text work on these arrays.
struct cstore_t {   // arrays of a block graph built on the host
    uvec<int32_t> len, outdeg, eto, steps, cons;
    uvec<uint8_t> indeg;
    uvec<char> seq;
    uvec<uint32_t> soff, eoff;
};
// A block graph is topologically numbered, so every edge runs forward-forward from a lower to a higher id: the edges
// are listed per tail (CSR), heads ascending -- the block's L lines in their order.
struct cblock_t {
    int64_t n = 0, ne = 0;
    bool has_paths = false;                                   // the block had sequences (its consensus path exists, even if empty)
    const int32_t *len = nullptr, *outdeg = nullptr, *eto = nullptr;
    const uint8_t* indeg = nullptr;                           // saturating at 255
    const char* seq = nullptr;
    std::vector<std::pair<const int32_t*, int64_t>> upath;    // steps of every dedup'd sequence, in alignment order
    const int32_t* cons = nullptr;
    int64_t ncons = 0;
    const uint32_t *soff = nullptr, *eoff = nullptr;          // [n+1] prefix sums of len / outdeg (storage: the chunk's, or `own`)
    std::vector<int32_t> range_useq;                          // path range (rank in the block) -> dedup'd sequence
    std::vector<char> range_rev;                              //                                  -> collected in reverse
    std::unique_ptr<cstore_t> own;
    int validated = 0;                                        // 1 / -1: its ranges were checked (ok / not) when its chunk came back
    void index(const collected_t& c, size_t n_ranges, uint32_t* so, uint32_t* eo) {
        uint32_t a = 0, e = 0;
        for (int64_t v = 0; v < n; ++v) { so[(size_t)v] = a; eo[(size_t)v] = e; a += (uint32_t)len[v]; e += (uint32_t)outdeg[v]; }
        so[(size_t)n] = a; eo[(size_t)n] = e;
        soff = so; eoff = eo;
        ne = e;
        range_useq.assign(n_ranges, -1); range_rev.assign(n_ranges, 0);
        for (size_t i = 0; i < c.dup_rank_in_path_ranges.size(); ++i)
            for (size_t j = 0; j < c.dup_rank_in_path_ranges[i].size(); ++j) {
                range_useq[(size_t)c.dup_rank_in_path_ranges[i][j]] = (int32_t)i;
                range_rev[(size_t)c.dup_rank_in_path_ranges[i][j]] = c.dup_is_revs[i][j] ? 1 : 0;
            }
    }
    bool has_edge(uint64_t v, uint64_t w) const {
        for (uint32_t x = eoff[(size_t)v]; x < eoff[(size_t)v + 1]; ++x) if ((uint64_t)eto[x] == w) return true;
        return false;
    }
};
// A9 + A10 on the host from raw POA results, for ONE path per dedup'd sequence: the duplicates of a sequence walk the
// same nodes (reversed and flipped when collected in reverse, src/smooth.cpp:2612-2615), so they add neither nodes nor
// canonical edges nor path ends to what unchop and the ordering see -- the graph is build_block_graph's.
void cblock_from_raw(cblock_t& B, const collected_t& c, size_t n_ranges, const uint8_t* node_code, int64_t n_nodes,
                     const std::vector<const int32_t*>& seq_paths, const int32_t* cons, int64_t n_cons, bool want_cons, bool abpoa) {
    static const char dec[5] = {'A', 'C', 'G', 'T', 'N'};
    ograph_t G;
    const size_t S = c.seqs.size();
    G.paths.resize(S + (want_cons ? 1 : 0));
    for (size_t i = 0; i < S; ++i) {
        const int64_t len = (int64_t)c.seqs[i].size();
        steps_t& st = G.paths[i].second;
        st.reserve((size_t)std::max<int64_t>(0, len - 2 * (int64_t)c.poa_padding));
        for (int64_t k = c.poa_padding; k < len - c.poa_padding; ++k) st.push_back(mk((uint64_t)seq_paths[i][k], false));
    }
    if (want_cons) {
        steps_t& st = G.paths[S].second;
        std::vector<char> visited;
        if (abpoa) {
            visited.assign((size_t)n_nodes, 0);
            for (size_t i = 0; i < S; ++i) for (handle_t h : G.paths[i].second) visited[nid(h)] = 1;
        }
        for (int64_t k = 0; k < n_cons; ++k)
            if (!abpoa || visited[(size_t)cons[k]]) st.push_back(mk((uint64_t)cons[k], false));
    }
    std::vector<int64_t> keep((size_t)n_nodes, -1);
    for (auto& pth : G.paths) for (handle_t h : pth.second) keep[nid(h)] = 0;
    for (int64_t v = 0, k = 0; v < n_nodes; ++v) if (keep[(size_t)v] == 0) { keep[(size_t)v] = k++; G.seq.push_back(std::string(1, dec[node_code[v] > 4 ? 4 : node_code[v]])); }
    edge_acc_t acc(G.seq.size());
    for (auto& pth : G.paths) {
        for (auto& h : pth.second) h = mk((uint64_t)keep[nid(h)], rev(h));
        for (size_t k = 1; k < pth.second.size(); ++k) acc.add(pth.second[k - 1], pth.second[k]);
    }
    acc.into(G.edges);
    unchop(G);
    topo_renumber(G);
    B.own.reset(new cstore_t());
    cstore_t& O = *B.own;
    const size_t n = G.seq.size();
    O.len.resize(n); O.outdeg.resize(n); O.indeg.resize(n);
    size_t bytes = 0;
    for (size_t v = 0; v < n; ++v) { O.len[v] = (int32_t)G.seq[v].size(); O.outdeg[v] = 0; O.indeg[v] = 0; bytes += G.seq[v].size(); }
    O.seq.resize(bytes + 1);
    { size_t a = 0; for (size_t v = 0; v < n; ++v) { memcpy(O.seq.data() + a, G.seq[v].data(), G.seq[v].size()); a += G.seq[v].size(); } }
    O.eto.resize(G.edges.size() + 1);
    for (size_t x = 0; x < G.edges.size(); ++x) {   // (canonical and sorted: by tail, then head)
        const edge_t& e = G.edges[x];
        if (rev(e.first) || rev(e.second) || nid(e.first) >= nid(e.second)) abort();   // (a topologically numbered DAG has no other kind)
        O.outdeg[nid(e.first)] += 1;
        if (O.indeg[nid(e.second)] < 255) O.indeg[nid(e.second)] += 1;
        O.eto[x] = (int32_t)nid(e.second);
    }
    size_t total = 0;
    for (size_t i = 0; i < S; ++i) total += G.paths[i].second.size();
    O.steps.resize(total + 1);
    B.upath.resize(S);
    size_t a = 0;
    for (size_t i = 0; i < S; ++i) {
        const steps_t& st = G.paths[i].second;
        for (size_t k = 0; k < st.size(); ++k) O.steps[a + k] = (int32_t)nid(st[k]);
        B.upath[i] = std::make_pair(O.steps.data() + a, (int64_t)st.size());
        a += st.size();
    }
    if (want_cons) {
        const steps_t& st = G.paths[S].second;
        O.cons.resize(st.size() + 1);
        for (size_t k = 0; k < st.size(); ++k) O.cons[k] = (int32_t)nid(st[k]);
        B.cons = O.cons.data(); B.ncons = (int64_t)st.size();
    }
    B.n = (int64_t)n; B.len = O.len.data(); B.outdeg = O.outdeg.data(); B.indeg = O.indeg.data(); B.eto = O.eto.data(); B.seq = O.seq.data();
    B.has_paths = S > 0;
    O.soff.resize(n + 1); O.eoff.resize(n + 1);
    B.index(c, n_ranges, O.soff.data(), O.eoff.data());
}
// ... or views of the arrays the provider returned for block `slot` of its batch
// (so / eo: room for the block's n + 1 offsets in the chunk's arrays)
void cblock_from_out(cblock_t& B, const collected_t& c, size_t n_ranges, const batch_t& Bt, const sxg_poa_batch_out& out, int64_t slot, bool want_cons,
                     uint32_t* so, uint32_t* eo) {
    const int64_t n0 = out.bg_node_off[slot];
    B.n = out.bg_node_off[slot + 1] - n0;
    B.len = out.bg_node_len + n0; B.outdeg = out.bg_node_outdeg + n0; B.indeg = out.bg_node_indeg + n0;
    B.seq = out.bg_seq + out.bg_seq_off[slot];
    B.eto = out.bg_edge_to + out.bg_edge_off[slot];
    const int32_t s0 = Bt.blk_off[(size_t)slot], s1 = Bt.blk_off[(size_t)slot + 1];
    B.upath.resize((size_t)(s1 - s0));
    for (int32_t sq = s0; sq < s1; ++sq) B.upath[(size_t)(sq - s0)] = std::make_pair(out.bg_steps + out.bg_step_off[sq], out.bg_step_off[sq + 1] - out.bg_step_off[sq]);
    if (want_cons && out.bg_cons_off && out.bg_cons_steps) { B.cons = out.bg_cons_steps + out.bg_cons_off[slot]; B.ncons = out.bg_cons_off[slot + 1] - out.bg_cons_off[slot]; }
    B.has_paths = s1 > s0;
    B.index(c, n_ranges, so, eo);
}

// Output sinks of the GFA writer: the size pass and the write pass run the same code.
struct count_sink_t {
    size_t n = 0;
    void raw(const char*, size_t len) { n += len; }
    void ch(char) { n += 1; }
    void num(uint64_t v) { n += digits_u64(v); }
};
struct write_sink_t {
    char* o;
    void raw(const char* q, size_t len) { memcpy(o, q, len); o += len; }
    void ch(char c) { *o++ = c; }
    void num(uint64_t v) {   // digits written in place, two at a time from the end
        static const char lut[201] =
            "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263"
            "646566676869707172737475767778798081828384858687888990919293949596979899";
        if (v < 4000000000ull) {
            uint32_t x = (uint32_t)v;
            const int d = x < 100000u ? (x < 100u ? (x < 10u ? 1 : 2) : (x < 10000u ? (x < 1000u ? 3 : 4) : 5))
                                      : (x < 10000000u ? (x < 1000000u ? 6 : 7) : (x < 1000000000u ? (x < 100000000u ? 8 : 9) : 10));
            char* e = o + d;
            while (x >= 100u) { const uint32_t r = x % 100u; x /= 100u; e -= 2; memcpy(e, lut + 2 * r, 2); }
            if (x >= 10u) { e -= 2; memcpy(e, lut + 2 * x, 2); } else *--e = (char)('0' + x);
            o += d;
            return;
        }
        char buf[24];
        int k = 24;
        while (v >= 100) { const unsigned r = (unsigned)(v % 100); v /= 100; buf[--k] = lut[2 * r + 1]; buf[--k] = lut[2 * r]; }
        if (v >= 10) { buf[--k] = lut[2 * v + 1]; buf[--k] = lut[2 * v]; } else buf[--k] = (char)('0' + v);
        memcpy(o, buf + k, (size_t)(24 - k));
        o += 24 - k;
    }
};

// The lacing half of the iteration on compact block graphs (src/main.cpp:599-1061), for the run without MAF merging
// (no flips, every block keeps its own consensus path): fragments by (path, start), validation, links between the
// fragments, the global unchop, GFA text.  Same bytes as the laced-graph path below (ograph_t; SXG_SMOOTH_LEGACY=1 keeps
// taking it, and the MAF / merge iteration always does), without building the laced graph:
//   * a laced path is its list of fragments -- (block, dedup'd sequence, orientation) -- and its steps are read from the
//     block's step array with the block's id offset when they are counted and when they are written;
//   * unchop (decree of DESIGN.md section 9) runs its test on every node -- out-degree from the block's CSR plus the
//     links that attach to the node, in-degree of the one successor, path ends -- but a block graph is already
//     unchopped, so what it finds are the few links and block edges whose path ends went away with lacing: the merged
//     chains live in small per-block tables, node ids shift by the number of removed nodes in front of them;
//   * edges keep their order under that renumbering except those that leave a merged chain and the links, which are
//     sorted apart and merged in by position while the L lines are written.
// Validation of one fragment (src/main.cpp:770-810, per range): the steps of the range's sequence in its block graph spell the
// range's original sequence.  `st` / `cnt` / `rv`: the dedup'd sequence's step list and whether the range was collected in reverse.
// Also checks what the text writer relies on: every step names a node of the block, a sequence's steps ascend strictly.
static bool fragment_spells_its_range(const sxg_graph* g, const path_range_t& r, const cblock_t& B, const int32_t* st, int64_t cnt, bool rv) {
    // the range's original sequence, once, into a thread-local buffer; then the fragment's nodes against it
    static thread_local std::string os;
    os.clear();
    for (uint64_t q = r.begin; q < r.end; ++q) {
        const handle_t h = g->steps[r.path][q];
        const std::string& sq = g->seq[nid(h)];
        if (!rev(h)) os.append(sq);
        else { const size_t a0 = os.size(); os.resize(a0 + sq.size()); for (size_t y = 0; y < sq.size(); ++y) os[a0 + y] = comp(sq[sq.size() - 1 - y]); }
    }
    const char* ob = os.data();
    const size_t on = os.size();
    size_t at = 0;
    bool ok = true;
    const char* bs = B.seq;
    const uint32_t* so = B.soff;
    // (the step lists come from the provider: every step must name a node of its block, and a sequence's steps must be
    //  strictly ascending -- the block graph is topologically numbered and its paths walk forward; the size pass of the
    //  text writer relies on it when it looks steps up by binary search)
    const uint64_t nB = (uint64_t)B.n;
    int64_t prev = -1;
    if (!rv) {
        // (comparing whole runs of consecutive ids with memcmp -- consecutive nodes lie side by side in the block's bytes --
        //  was measured on the box: 0.060 s against 0.042 s for this loop; the runs are a few bases long)
        for (int64_t j = 0; j < cnt; ++j) {
            if ((uint64_t)(uint32_t)st[j] >= nB || (int64_t)st[j] <= prev) { ok = false; break; }
            prev = st[j];
            const uint32_t a0 = so[st[j]], a1 = so[st[j] + 1];
            if (at + (a1 - a0) > on) { ok = false; break; }
            for (uint32_t y = a0; y < a1; ++y) ok &= bs[y] == ob[at++];
        }
    } else {
        prev = (int64_t)nB;
        for (int64_t j = cnt - 1; j >= 0; --j) {
            if ((uint64_t)(uint32_t)st[j] >= nB || (int64_t)st[j] >= prev) { ok = false; break; }
            prev = st[j];
            const uint32_t a0 = so[st[j]], a1 = so[st[j] + 1];
            if (at + (a1 - a0) > on) { ok = false; break; }
            for (uint32_t y = a1; y > a0; --y) ok &= comp(bs[y - 1]) == ob[at++];
        }
    }
    return ok && at == on;
}
// ... of every range of a block (the chunk pipeline calls this while the POA provider works on the next chunk)
static bool block_spells_its_ranges(const sxg_graph* g, const std::vector<path_range_t>& ranges, const cblock_t& B) {
    if (B.n == 0) return true;
    for (size_t ri = 0; ri < ranges.size(); ++ri) {
        const int32_t u = B.range_useq[ri];
        if (!fragment_spells_its_range(g, ranges[ri], B, u >= 0 ? B.upath[(size_t)u].first : nullptr, u >= 0 ? B.upath[(size_t)u].second : 0, B.range_rev[ri] != 0)) return false;
    }
    return true;
}

struct lace_timer_t { std::function<void(const char*)> lap; };
int lace_fast(const sxg_graph* g, const sxg_blockset* b, const sxg_smooth_params* p, std::vector<cblock_t>& cb, char** out_gfa,
              const std::function<void(const char*)>& lap) {
    const int64_t nb = (int64_t)cb.size();
    struct frag_t { uint64_t path, start, end; int64_t target, block; };
    std::vector<frag_t> mapping;
    for (int64_t k = 0; k < nb; ++k) {
        if (cb[(size_t)k].n == 0) continue;
        int64_t path_id = 0;
        for (auto& r : b->blocks[(size_t)k]) mapping.push_back(frag_t{r.path, g->pos[r.path][r.begin], g->pos[r.path][r.end], path_id++, k});
    }
    std::stable_sort(mapping.begin(), mapping.end(), [](const frag_t& a, const frag_t& c) { return a.path < c.path || (a.path == c.path && a.start < c.start); });
    std::vector<uint64_t> id_trans((size_t)nb + 1, 0);
    for (int64_t k = 0; k < nb; ++k) id_trans[(size_t)k + 1] = id_trans[(size_t)k] + (uint64_t)cb[(size_t)k].n;
    const uint64_t n_laced = id_trans[(size_t)nb];
    auto block_of = [&](uint64_t X) { return (int64_t)(std::upper_bound(id_trans.begin(), id_trans.end(), X) - id_trans.begin()) - 1; };
    std::vector<std::pair<size_t, size_t>> runs;   // [a, z) of every path's fragments
    for (size_t a = 0; a < mapping.size();) {
        size_t z = a;
        while (z < mapping.size() && mapping[z].path == mapping[a].path) ++z;
        runs.emplace_back(a, z);
        a = z;
    }
    const size_t nf = mapping.size(), nruns = runs.size();
    // a fragment's steps: the path of its dedup'd sequence, walked backwards with flipped handles when the range was collected in reverse
    struct fview_t { const int32_t* st; int64_t cnt; bool rv; uint64_t o; };
    auto fview = [&](size_t f) {
        const cblock_t& B = cb[(size_t)mapping[f].block];
        const int32_t u = B.range_useq[(size_t)mapping[f].target];
        fview_t v;
        v.st = u >= 0 ? B.upath[(size_t)u].first : nullptr; v.cnt = u >= 0 ? B.upath[(size_t)u].second : 0;
        v.rv = B.range_rev[(size_t)mapping[f].target] != 0; v.o = id_trans[(size_t)mapping[f].block];
        return v;
    };
    auto ffront = [&](const fview_t& v) { return v.rv ? mk(v.o + (uint64_t)v.st[v.cnt - 1], true) : mk(v.o + (uint64_t)v.st[0], false); };
    auto fback = [&](const fview_t& v) { return v.rv ? mk(v.o + (uint64_t)v.st[0], true) : mk(v.o + (uint64_t)v.st[v.cnt - 1], false); };
    // coverage, links between the fragments of a path, the ends of the laced paths (parallel over paths, src/main.cpp:706-764)
    std::vector<std::string> errs(nruns);
    std::vector<std::vector<edge_t>> links(nruns);
    std::vector<handle_t> pfront(nruns, 0), pback(nruns, 0);
    std::vector<char> pany(nruns, 0);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t q = 0; q < (int64_t)nruns; ++q) {
        const size_t a = runs[(size_t)q].first, z = runs[(size_t)q].second;
        uint64_t last_end = 0;
        bool any = false;
        handle_t last = 0;
        for (size_t f = a; f < z; ++f) {
            if (mapping[f].start != last_end) { errs[(size_t)q] = "path " + g->pname[mapping[a].path] + " is not covered by the blocks"; break; }
            const fview_t v = fview(f);
            if (v.cnt > 0) {
                if (any) links[(size_t)q].push_back(ograph_t::canon(last, ffront(v)));
                else pfront[(size_t)q] = ffront(v);
                last = fback(v);
                any = true;
            }
            last_end = mapping[f].end;
        }
        if (errs[(size_t)q].empty() && last_end != g->pos[mapping[a].path].back())
            errs[(size_t)q] = "path " + g->pname[mapping[a].path] + " is not covered to its end";
        pany[(size_t)q] = any ? 1 : 0; pback[(size_t)q] = last;
    }
    for (auto& e : errs) if (!e.empty()) return fail(SXG_E_INVALID, e);
    lap("lacing");
    // validation (src/main.cpp:770-810): every laced path spells its original sequence.  The fragments tile their path
    // (checked above), so that is: every fragment spells its range.
    {
        size_t nonempty = 0;
        for (auto& st : g->steps) if (!st.empty()) ++nonempty;
        if (nruns != nonempty) return fail(SXG_E_INVALID, "path count mismatch between input and smoothed graph");
        int64_t bad = -1;
#pragma omp parallel for schedule(dynamic, 16)
        for (int64_t f = 0; f < (int64_t)nf; ++f) {
            const cblock_t& B = cb[(size_t)mapping[(size_t)f].block];
            if (B.validated == 1) continue;   // (done per block while the provider worked on a later chunk)
            bool ok = B.validated == 0;       // (-1: that check failed -- report it with the path's name here)
            if (ok) {
                const fview_t v = fview((size_t)f);
                const path_range_t& r = b->blocks[(size_t)mapping[(size_t)f].block][(size_t)mapping[(size_t)f].target];
                ok = fragment_spells_its_range(g, r, B, v.st, v.cnt, v.rv);
            }
            if (!ok) {
#pragma omp critical(sxg_lace_bad)
                if (bad < 0 || f < bad) bad = f;
            }
        }
        if (bad >= 0) return fail(SXG_E_INVALID, "path " + g->pname[mapping[(size_t)bad].path] + " was corrupted in the smoothed graph");
    }
    lap("validation");
    // consensus paths (src/main.cpp:812-869): one per block that had sequences, after the input paths
    std::vector<int64_t> cons_blocks;
    if (p->add_consensus) for (int64_t k = 0; k < nb; ++k) if (cb[(size_t)k].has_paths) cons_blocks.push_back(k);
    // ---- the global unchop (src/main.cpp:1021), incrementally ----
    // nodes at which a path starts or ends block a merge on that side
    std::vector<uint64_t> rblk, lblk;   // rblk: a path ends at u+ / starts at u-; lblk: a path starts at v+ / ends at v-
    auto path_ends = [&](handle_t front, handle_t back) {
        if (rev(front)) rblk.push_back(nid(front)); else lblk.push_back(nid(front));
        if (rev(back)) lblk.push_back(nid(back)); else rblk.push_back(nid(back));
    };
    for (size_t q = 0; q < nruns; ++q) if (pany[q]) path_ends(pfront[q], pback[q]);
    for (int64_t k : cons_blocks) {
        const cblock_t& B = cb[(size_t)k];
        if (B.ncons > 0) path_ends(mk(id_trans[(size_t)k] + (uint64_t)B.cons[0], false), mk(id_trans[(size_t)k] + (uint64_t)B.cons[B.ncons - 1], false));
    }
    std::sort(rblk.begin(), rblk.end()); std::sort(lblk.begin(), lblk.end());
    // links that the blocks do not hold already
    std::vector<edge_t> lk;
    for (auto& l : links) lk.insert(lk.end(), l.begin(), l.end());
    std::sort(lk.begin(), lk.end());
    lk.erase(std::unique(lk.begin(), lk.end()), lk.end());
    {
        size_t w = 0;
        for (size_t j = 0; j < lk.size(); ++j) {
            const edge_t& e = lk[j];
            bool have = false;
            if (!rev(e.first) && !rev(e.second) && nid(e.first) < nid(e.second)) {
                const int64_t k = block_of(nid(e.first));
                if (k == block_of(nid(e.second))) have = cb[(size_t)k].has_edge(nid(e.first) - id_trans[(size_t)k], nid(e.second) - id_trans[(size_t)k]);
            }
            if (!have) lk[w++] = e;
        }
        lk.resize(w);
    }
    // what the links add to the sides of their nodes: (handle the edge leaves, handle it arrives at)
    std::vector<edge_t> adj;
    adj.reserve(2 * lk.size());
    for (auto& e : lk) { adj.emplace_back(e.first, e.second); adj.emplace_back(flip(e.second), flip(e.first)); }
    std::sort(adj.begin(), adj.end());
    auto adj_count = [&](handle_t h) {
        auto lo = std::lower_bound(adj.begin(), adj.end(), edge_t(h, 0));
        size_t c = 0;
        while (lo != adj.end() && lo->first == h) { ++c; ++lo; }
        return c;
    };
    std::vector<std::vector<std::pair<uint64_t, uint64_t>>> found((size_t)nb);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t k = 0; k < nb; ++k) {
        const cblock_t& B = cb[(size_t)k];
        if (B.n == 0) continue;
        const uint64_t o = id_trans[(size_t)k];
        size_t ai = (size_t)(std::lower_bound(adj.begin(), adj.end(), edge_t(mk(o, false), 0)) - adj.begin());
        for (int64_t u = 0; u < B.n; ++u) {
            const uint64_t U = o + (uint64_t)u;
            const handle_t uf = mk(U, false);
            while (ai < adj.size() && adj[ai].first < uf) ++ai;
            size_t cr = 0;
            handle_t other = 0;
            for (size_t j = ai; j < adj.size() && adj[j].first == uf; ++j) { ++cr; other = adj[j].second; }
            if ((size_t)B.outdeg[u] + cr != 1) continue;
            const handle_t vf = B.outdeg[u] == 1 ? mk(o + (uint64_t)B.eto[B.eoff[(size_t)u]], false) : other;
            if (rev(vf) || nid(vf) == U) continue;
            const uint64_t V = nid(vf);
            const int64_t kv = B.outdeg[u] == 1 ? k : block_of(V);
            const unsigned ind = cb[(size_t)kv].indeg[V - id_trans[(size_t)kv]];
            if (ind > 1) continue;
            if (ind + adj_count(mk(V, true)) != 1) continue;   // (the one edge on v's left side is then the one from u+)
            if (std::binary_search(rblk.begin(), rblk.end(), U) || std::binary_search(lblk.begin(), lblk.end(), V)) continue;
            found[(size_t)k].emplace_back(U, V);
        }
    }
    std::unordered_map<uint64_t, uint64_t> nxt, prv;
    for (auto& fv : found) for (auto& m : fv) { nxt[m.first] = m.second; prv[m.second] = m.first; }
    // chains hang off their heads; what no head reaches is a pure cycle, broken at its smallest member
    struct cmem_t { uint32_t local; uint8_t first, last; uint64_t head; };   // head: the chain's head (laced id), later its new id
    std::vector<std::vector<cmem_t>> cmem((size_t)nb);
    std::vector<std::vector<uint32_t>> rem((size_t)nb);      // removed nodes (chain members that are not heads), by block
    std::vector<std::vector<uint64_t>> chains;
    {
        std::vector<uint64_t> keys;
        for (auto& kv : nxt) keys.push_back(kv.first);
        for (auto& kv : prv) keys.push_back(kv.first);
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        std::unordered_map<uint64_t, char> reached;
        auto walk = [&](uint64_t h) {
            chains.emplace_back();
            for (uint64_t x = h;;) { reached[x] = 1; chains.back().push_back(x); auto it = nxt.find(x); if (it == nxt.end()) break; x = it->second; }
        };
        for (uint64_t u : keys) if (!prv.count(u)) walk(u);
        for (uint64_t u : keys) {
            if (reached.count(u)) continue;
            nxt.erase(prv[u]); prv.erase(u);   // u is the smallest member of its cycle: smaller ones would have been met first
            walk(u);
        }
        std::sort(chains.begin(), chains.end(), [](const std::vector<uint64_t>& a, const std::vector<uint64_t>& c) { return a[0] < c[0]; });
        for (auto& ch : chains)
            for (size_t x = 0; x < ch.size(); ++x) {
                const int64_t k = block_of(ch[x]);
                cmem[(size_t)k].push_back(cmem_t{(uint32_t)(ch[x] - id_trans[(size_t)k]), (uint8_t)(x == 0), (uint8_t)(x + 1 == ch.size()), ch[0]});
                if (x) rem[(size_t)k].push_back((uint32_t)(ch[x] - id_trans[(size_t)k]));
            }
        for (int64_t k = 0; k < nb; ++k) {
            std::sort(cmem[(size_t)k].begin(), cmem[(size_t)k].end(), [](const cmem_t& a, const cmem_t& c) { return a.local < c.local; });
            std::sort(rem[(size_t)k].begin(), rem[(size_t)k].end());
        }
    }
    std::vector<uint64_t> basem((size_t)nb + 1, 0);
    for (int64_t k = 0; k < nb; ++k) basem[(size_t)k + 1] = basem[(size_t)k] + rem[(size_t)k].size();
    // new (0-based) id of a node that is not removed
    auto newid_in = [&](int64_t k, uint64_t x) {
        const auto& r = rem[(size_t)k];
        return id_trans[(size_t)k] + x - basem[(size_t)k] - (uint64_t)(std::lower_bound(r.begin(), r.end(), (uint32_t)x) - r.begin());
    };
    for (int64_t k = 0; k < nb; ++k)
        for (auto& m : cmem[(size_t)k]) { const int64_t kh = block_of(m.head); m.head = newid_in(kh, m.head - id_trans[(size_t)kh]); }
    auto find_cm = [&](int64_t k, uint64_t x) -> const cmem_t* {
        const auto& c = cmem[(size_t)k];
        for (auto& m : c) if (m.local == (uint32_t)x) return &m;
        return nullptr;
    };
    // new id of any node (a chain member maps to its chain)
    auto newid_any = [&](uint64_t X) {
        const int64_t k = block_of(X);
        const uint64_t x = X - id_trans[(size_t)k];
        if (!cmem[(size_t)k].empty()) if (const cmem_t* m = find_cm(k, x)) return m->head;
        return newid_in(k, x);
    };
    auto remap = [&](handle_t h) { return mk(newid_any(nid(h)), rev(h)); };
    // edges that do not keep their place: the links (mapped) and every block edge that touches a chain member (its
    // canonical form and its place among the L lines follow the chain's new id); the edges inside a chain go away
    auto is_member = [&](int64_t k, uint64_t x) {
        const auto& c = cmem[(size_t)k];
        return !c.empty() && x >= c.front().local && x <= c.back().local && find_cm(k, x) != nullptr;
    };
    auto interior = [&](uint64_t tail, uint64_t head) { auto it = nxt.find(tail); return it != nxt.end() && it->second == head; };
    std::vector<edge_t> extra;
    for (auto& e : lk) {   // (the merged link u+ -> v+ may be stored as (u+, v+) or as (v-, u-))
        if (!rev(e.first) && !rev(e.second) && interior(nid(e.first), nid(e.second))) continue;
        if (rev(e.first) && rev(e.second) && interior(nid(e.second), nid(e.first))) continue;
        extra.push_back(ograph_t::canon(remap(e.first), remap(e.second)));
    }
    for (int64_t k = 0; k < nb; ++k) {
        if (cmem[(size_t)k].empty()) continue;
        const cblock_t& B = cb[(size_t)k];
        const uint64_t o = id_trans[(size_t)k];
        for (int64_t u = 0; u < B.n; ++u) {
            const bool mu = is_member(k, (uint64_t)u);
            for (uint32_t x = B.eoff[(size_t)u]; x < B.eoff[(size_t)u + 1]; ++x) {
                const uint64_t w = (uint64_t)B.eto[x];
                if (!mu && !is_member(k, w)) continue;
                const uint64_t tail = o + (uint64_t)u, head = o + w;
                if (interior(tail, head)) continue;
                extra.push_back(ograph_t::canon(mk(newid_any(tail), false), mk(newid_any(head), false)));
            }
        }
    }
    std::sort(extra.begin(), extra.end());
    extra.erase(std::unique(extra.begin(), extra.end()), extra.end());
    // ... handed to the block whose (new) id range holds their first node
    std::vector<uint64_t> nstart((size_t)nb + 1);
    for (int64_t k = 0; k <= nb; ++k) nstart[(size_t)k] = id_trans[(size_t)k] - basem[(size_t)k];
    std::vector<size_t> xlo((size_t)nb + 1, extra.size());
    {
        size_t j = 0;
        for (int64_t k = 0; k < nb; ++k) {
            xlo[(size_t)k] = j;
            while (j < extra.size() && nid(extra[j].first) < nstart[(size_t)k + 1]) ++j;
        }
        xlo[(size_t)nb] = j;
        // (nothing is left over: every first node lies in some block's range; ids beyond the last block cannot occur)
        if (j != extra.size()) return fail(SXG_E_INVALID, "internal: edge outside the laced graph");
    }
    lap("unchop");
    // ---- GFA text: sizes, then every piece in place ----
    auto emit_S = [&](auto& sk, int64_t k) {
        const cblock_t& B = cb[(size_t)k];
        const auto& r = rem[(size_t)k];
        size_t ri = 0;
        uint64_t id = nstart[(size_t)k] + 1;
        for (int64_t v = 0; v < B.n; ++v) {
            if (ri < r.size() && r[ri] == (uint32_t)v) { ++ri; continue; }
            sk.ch('S'); sk.ch('\t'); sk.num(id++); sk.ch('\t');
            const cmem_t* m = cmem[(size_t)k].empty() ? nullptr : find_cm(k, (uint64_t)v);
            if (!m) sk.raw(B.seq + B.soff[(size_t)v], (size_t)(B.soff[(size_t)v + 1] - B.soff[(size_t)v]));
            else {   // a chain head: the sequences of its members
                const auto it = std::lower_bound(chains.begin(), chains.end(), id_trans[(size_t)k] + (uint64_t)v,
                                                 [](const std::vector<uint64_t>& c, uint64_t x) { return c[0] < x; });
                for (uint64_t X : *it) {
                    const int64_t kx = block_of(X);
                    const cblock_t& Bx = cb[(size_t)kx];
                    const uint64_t x = X - id_trans[(size_t)kx];
                    sk.raw(Bx.seq + Bx.soff[(size_t)x], (size_t)(Bx.soff[(size_t)x + 1] - Bx.soff[(size_t)x]));
                }
            }
            sk.ch('\n');
        }
    };
    auto put_L = [&](auto& sk, const edge_t& e) {
        sk.ch('L'); sk.ch('\t'); sk.num(nid(e.first) + 1); sk.ch('\t'); sk.ch(rev(e.first) ? '-' : '+'); sk.ch('\t');
        sk.num(nid(e.second) + 1); sk.ch('\t'); sk.ch(rev(e.second) ? '-' : '+'); sk.raw("\t0M\n", 4);
    };
    auto emit_L = [&](auto& sk, int64_t k) {
        const cblock_t& B = cb[(size_t)k];
        const auto& r = rem[(size_t)k];
        const bool members = !cmem[(size_t)k].empty();
        size_t xj = xlo[(size_t)k];
        const size_t xz = xlo[(size_t)k + 1];
        size_t ri = 0;
        uint64_t id = nstart[(size_t)k];
        const uint64_t base = nstart[(size_t)k];
        for (int64_t u = 0; u < B.n; ++u) {
            if (ri < r.size() && r[ri] == (uint32_t)u) { ++ri; continue; }
            const uint64_t nu = id++;
            if (members && is_member(k, (uint64_t)u)) continue;   // (a chain head: its edges are in `extra`)
            for (uint32_t x = B.eoff[(size_t)u]; x < B.eoff[(size_t)u + 1]; ++x) {
                const uint64_t w = (uint64_t)B.eto[x];
                if (members && is_member(k, w)) continue;           // (in `extra`)
                const uint64_t nw = members ? newid_in(k, w) : base + w;
                const edge_t e(mk(nu, false), mk(nw, false));
                while (xj < xz && extra[xj] < e) put_L(sk, extra[xj++]);
                if (xj < xz && extra[xj] == e) ++xj;
                put_L(sk, e);
            }
        }
        while (xj < xz) put_L(sk, extra[xj++]);
    };
    // steps of one fragment; `first` = nothing of this path has been written yet.  Returns the steps written.
    auto emit_F = [&](auto& sk, size_t f, bool first) -> uint64_t {
        const fview_t v = fview(f);
        const int64_t k = mapping[f].block;
        const auto& cm = cmem[(size_t)k];
        const uint64_t base = nstart[(size_t)k] + 1;
        const char sign = v.rv ? '-' : '+';
        uint64_t n = 0;
        if (cm.empty()) {
            for (int64_t j = 0; j < v.cnt; ++j) {
                const uint64_t x = (uint64_t)v.st[v.rv ? v.cnt - 1 - j : j];
                if (!first) sk.ch(',');
                first = false;
                sk.num(base + x); sk.ch(sign); ++n;
            }
            return n;
        }
        const uint32_t lo = cm.front().local, hi = cm.back().local;
        const uint64_t nrem = rem[(size_t)k].size();
        for (int64_t j = 0; j < v.cnt; ++j) {
            const uint64_t x = (uint64_t)v.st[v.rv ? v.cnt - 1 - j : j];
            uint64_t id;
            if (x < lo) id = base + x;
            else if (x > hi) id = base + x - nrem;
            else if (const cmem_t* m = find_cm(k, x)) {
                if (v.rv ? !m->last : !m->first) continue;   // a chain is stepped on once: at its head forwards, at its last node backwards
                id = m->head + 1;
            } else id = newid_in(k, x) + 1;
            if (!first) sk.ch(',');
            first = false;
            sk.num(id); sk.ch(sign); ++n;
        }
        return n;
    };
    // The write pass of a fragment.  A path walks a topologically numbered block in ascending ids (descending when the
    // range was collected in reverse) and mostly in small strides, so the decimal digits of the last id are kept and
    // bumped in place instead of being derived again for every step; ids go out with one blind 16-byte store while that stays
    // inside the fragment's piece (the steps behind them overwrite what it spills), with exact ones at its end.
    struct inc_digits_t {
        char dig[32];
        int d = 0;
        uint64_t cur = 0;
        void set(uint64_t v) { d = (int)digits_u64(v); cur = v; for (int i = d - 1; i >= 0; --i) { dig[i] = (char)('0' + v % 10); v /= 10; } }
        void to(uint64_t v) {
            const int64_t dl = (int64_t)(v - cur);
            if (d == 0 || dl >= 10 || dl <= -10) { set(v); return; }
            cur = v;
            int i = d - 1;
            const int c = dig[i] + (int)dl;
            if (c > '9') {
                dig[i] = (char)(c - 10);
                for (--i; i >= 0; --i) { if (dig[i] != '9') { dig[i]++; return; } dig[i] = '0'; }
                set(v);   // (one digit more)
            } else if (c < '0') {
                dig[i] = (char)(c + 10);
                for (--i; i >= 0; --i) { if (dig[i] != '0') { dig[i]--; if (i == 0 && dig[0] == '0') set(v); return; } dig[i] = '9'; }
                set(v);
            } else dig[i] = (char)c;
        }
    };
    auto write_F = [&](char* o, const char* lim, size_t f, bool first) -> char* {   // lim: end of the fragment's piece
        const fview_t v = fview(f);
        const int64_t k = mapping[f].block;
        const auto& cm = cmem[(size_t)k];
        const uint64_t base = nstart[(size_t)k] + 1;
        const char sign = v.rv ? '-' : '+';
        const bool members = !cm.empty();
        const uint32_t lo = members ? cm.front().local : 0xffffffffu, hi = members ? cm.back().local : 0;
        const uint64_t nrem = rem[(size_t)k].size();
        inc_digits_t w;
        if (base + (uint64_t)cb[(size_t)k].n >= 100000000000000ull) {   // (ids beyond 14 digits: the plain writer)
            write_sink_t sk{o};
            emit_F(sk, f, first);
            return sk.o;
        }
        for (int64_t j = 0; j < v.cnt; ++j) {
            const uint64_t x = (uint64_t)v.st[v.rv ? v.cnt - 1 - j : j];
            uint64_t id;
            if (x < lo) id = base + x;
            else if (x > hi) id = base + x - nrem;
            else if (const cmem_t* m = find_cm(k, x)) {
                if (v.rv ? !m->last : !m->first) continue;
                id = m->head + 1;
            } else id = newid_in(k, x) + 1;
            if (!first) *o++ = ',';
            first = false;
            w.to(id);
            if (o + 16 <= lim) memcpy(o, w.dig, 16); else memcpy(o, w.dig, (size_t)w.d);
            o += w.d;
            *o++ = sign;
        }
        return o;
    };
    auto emit_C = [&](auto& sk, int64_t k) {   // the consensus path of block k (forward steps)
        const cblock_t& B = cb[(size_t)k];
        const std::string name = cons_name(*p, k);
        sk.ch('P'); sk.ch('\t'); sk.raw(name.data(), name.size()); sk.ch('\t');
        bool first = true;
        for (int64_t j = 0; j < B.ncons; ++j) {
            const uint64_t x = (uint64_t)B.cons[j];
            uint64_t id;
            if (cmem[(size_t)k].empty()) id = nstart[(size_t)k] + 1 + x;
            else if (const cmem_t* m = find_cm(k, x)) { if (!m->first) continue; id = m->head + 1; }
            else id = newid_in(k, x) + 1;
            if (!first) sk.ch(',');
            first = false;
            sk.num(id); sk.ch('+');
        }
        sk.raw("\t*\n", 3);
    };
    // pieces: header | S of every block | L of every block | every fragment | every consensus path
    static const char head[] = "H\tVN:Z:1.0\n";
    const size_t P_S = 1, P_L = P_S + (size_t)nb, P_F = P_L + (size_t)nb, P_C = P_F + nf, pieces = P_C + cons_blocks.size();
    std::vector<size_t> off(pieces + 1, 0);
    std::vector<uint64_t> fsteps(nf, 0);   // steps every fragment writes (a path's comma logic needs the fragments before it)
    off[1] = sizeof(head) - 1;
    std::vector<size_t> run_of(nf);
    for (size_t q = 0; q < nruns; ++q) for (size_t f = runs[q].first; f < runs[q].second; ++f) run_of[f] = q;
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t q = 1; q < (int64_t)pieces; ++q) {
        count_sink_t sk;
        const size_t x = (size_t)q;
        if (x < P_L) emit_S(sk, (int64_t)(x - P_S));
        else if (x < P_F) emit_L(sk, (int64_t)(x - P_L));
        else if (x < P_C) {
            // (counted as if the fragment opened its path: the commas between fragments are added below)
            const size_t f = x - P_F;
            // A block's ids mostly have one digit count, and a path walks a topologically numbered block in ascending
            // ids: then the size follows from the number of steps -- minus the chain members the walk passes without
            // writing them, found by bisection -- and the steps (1 GB on the headline batch) are not read in this pass.
            const fview_t v = fview(f);
            const int64_t k = mapping[f].block;
            const uint64_t lo_id = nstart[(size_t)k] + 1, hi_id = nstart[(size_t)k] + (uint64_t)cb[(size_t)k].n;
            if (v.cnt > 0 && digits_u64(lo_id) == digits_u64(hi_id)) {
                uint64_t n = (uint64_t)v.cnt;
                bool plain = true;
                for (auto& m : cmem[(size_t)k]) {
                    if (v.rv ? m.last : m.first) {   // written under the chain's id, which may have another digit count
                        if (digits_u64(m.head + 1) != digits_u64(lo_id) && std::binary_search(v.st, v.st + v.cnt, (int32_t)m.local)) { plain = false; break; }
                        continue;
                    }
                    if (std::binary_search(v.st, v.st + v.cnt, (int32_t)m.local)) --n;
                }
                if (plain) {
                    fsteps[f] = n;
                    sk.n = n ? n * (digits_u64(lo_id) + 1) + (n - 1) : 0;
                } else fsteps[f] = emit_F(sk, f, true);
            } else fsteps[f] = emit_F(sk, f, true);
        } else emit_C(sk, cons_blocks[x - P_C]);
        off[x + 1] = sk.n;
    }
    // path prefix / suffix and the comma in front of every fragment that is not the first to write steps
    std::vector<char> lead(nf, 0);
    for (size_t q = 0; q < nruns; ++q) {
        const size_t a = runs[q].first, z = runs[q].second;
        bool any = false;
        for (size_t f = a; f < z; ++f) {
            if (fsteps[f] > 0) { if (any) { lead[f] = 1; off[P_F + f + 1] += 1; } any = true; }
        }
        off[P_F + a + 1] += 3 + g->pname[mapping[a].path].size();   // "P\t" name "\t"
        off[P_F + (z - 1) + 1] += 3;                                // "\t*\n"
    }
    for (size_t q = 0; q < pieces; ++q) off[q + 1] += off[q];
    char* buf = (char*)big_alloc(off.back() + 1);
    if (!buf) return fail(SXG_E_NOMEM, "out of memory for the GFA text");
    memcpy(buf, head, sizeof(head) - 1);
    int64_t wrong = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : wrong)
    for (int64_t q = 1; q < (int64_t)pieces; ++q) {
        const size_t x = (size_t)q;
        write_sink_t sk{buf + off[x]};
        if (x < P_L) emit_S(sk, (int64_t)(x - P_S));
        else if (x < P_F) emit_L(sk, (int64_t)(x - P_L));
        else if (x < P_C) {
            const size_t f = x - P_F, rq = run_of[f];
            if (f == runs[rq].first) { const std::string& nm = g->pname[mapping[f].path]; sk.ch('P'); sk.ch('\t'); sk.raw(nm.data(), nm.size()); sk.ch('\t'); }
            sk.o = write_F(sk.o, buf + off[x + 1], f, !lead[f]);
            if (f + 1 == runs[rq].second) sk.raw("\t*\n", 3);
        } else emit_C(sk, cons_blocks[x - P_C]);
        if (sk.o != buf + off[x + 1]) ++wrong;
    }
    if (wrong) { free(buf); return fail(SXG_E_INVALID, "internal: GFA size and write passes disagree"); }
    buf[off.back()] = 0;
    (void)n_laced;
    *out_gfa = buf;
    lap("GFA text");
    return SXG_OK;
}

// every entry point that takes the parameters: the caller's struct is this header's, and the scores fit the engine's int8
int check_params(const sxg_smooth_params* p) {
    if (!p) return fail(SXG_E_INVALID, "NULL parameters");
    if (p->struct_size != sizeof(sxg_smooth_params))
        return fail(SXG_E_INVALID, "sxg_smooth_params was not initialised by sxg_smooth_default_params of this library's header (struct_size " +
                                       std::to_string(p->struct_size) + ", expected " + std::to_string(sizeof(sxg_smooth_params)) + ")");
    const int v[6] = {p->poa_m, p->poa_n, p->poa_g, p->poa_e, p->poa_q, p->poa_c};
    for (int x : v) if (x < 0) return fail(SXG_E_INVALID, "POA scores are given as non-negative numbers (CLI convention)");
    int worst = 0;
    if (!p->use_abpoa) for (int x : v) worst = std::max(worst, x);
    else worst = std::max(std::max(p->poa_m, p->poa_n), std::max(p->poa_g + p->poa_e, p->poa_q + p->poa_c));   // g = -(o + e)
    if (worst > 127) return fail(SXG_E_INVALID, "POA scores beyond 127 do not fit the engine's int8 parameters (src/smooth.cpp:631-632 narrows them the same way)");
    return SXG_OK;
}

char* dup_out(const std::string& s) {
    char* r = (char*)malloc(s.size() + 1);
    if (r) memcpy(r, s.c_str(), s.size() + 1);
    return r;
}

}  // namespace

// =============================================================================================
extern "C" {

int sxg_smooth_abi_version(void) { return SXG_SMOOTH_ABI_VERSION; }

void sxg_smooth_default_params(sxg_smooth_params* p) {
    if (!p) return;
    p->struct_size = (uint32_t)sizeof(sxg_smooth_params);
    p->abpoa_band_local = 1;
    p->poa_spoa_order = 1;   // (round 6: what smooth_spoa's graph.AddAlignment does, src/smooth.cpp:764 -- spoa re-sorts the graph every time)
    p->poa_m = 1; p->poa_n = 4; p->poa_g = 6; p->poa_e = 2; p->poa_q = 26; p->poa_c = 1;  // src/main.cpp:322-327
    p->local_alignment = 1;                                                              // src/main.cpp:487
    p->poa_padding_fraction = 0.001f; p->max_block_depth_for_padding_more = 1000;         // src/main.cpp:293-295
    p->add_consensus = 0; p->consensus_base_name = "Consensus_";
    p->adaptive_poa_params = 0; p->kmer_size = 17;                                        // src/main.cpp:111,304
    p->use_abpoa = 0;                                                                    // src/main.cpp:130-132
}

void sxg_adaptive_poa_scores(float thr, const int32_t set_scores[6], int32_t out_scores[6]) { adaptive_scores(thr, set_scores, out_scores); }

int sxg_block_identity_threshold(const sxg_graph* g, const sxg_blockset* b, int64_t block_id, int32_t kmer_size, float* thr, int32_t* n_used) {
    if (!g || !b || !thr || !n_used || block_id < 0 || block_id >= (int64_t)b->blocks.size() || kmer_size < 1 || kmer_size > 32)
        return fail(SXG_E_INVALID, "bad argument");
    *thr = 0;
    *n_used = identity_threshold(*g, b->blocks[block_id], kmer_size, thr);
    return SXG_OK;
}
const char* sxg_smooth_last_error(void) { return g_err.c_str(); }
void sxg_smooth_free(void* p) { free(p); }

int sxg_graph_from_gfa(const char* text, size_t len, sxg_graph** out) {
    if (!text || !out) return fail(SXG_E_INVALID, "NULL argument");
    *out = nullptr;
    std::vector<std::pair<int64_t, std::string>> nodes;
    std::vector<std::pair<std::string, std::string>> plines;
    struct lrec_t { int64_t a, b; bool ar, br; };
    std::vector<lrec_t> llines;
    size_t i = 0;
    while (i < len) {
        size_t j = i;
        while (j < len && text[j] != '\n') ++j;
        if (j > i) {
            std::vector<std::string> f;
            size_t a = i;
            for (size_t k = i; k <= j; ++k)
                if (k == j || text[k] == '\t') { f.emplace_back(text + a, k - a); a = k + 1; }
            if (!f.empty() && !f.back().empty() && f.back().back() == '\r') f.back().pop_back();
            if (f[0] == "S" && f.size() >= 3) {
                std::string s = f[2];
                for (auto& ch : s) { ch = (char)toupper(ch); if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T') ch = 'N'; }
                nodes.emplace_back(atoll(f[1].c_str()), s);
            } else if (f[0] == "P" && f.size() >= 3) plines.emplace_back(f[1], f[2]);
            else if (f[0] == "L" && f.size() >= 5) llines.push_back(lrec_t{atoll(f[1].c_str()), atoll(f[3].c_str()), f[2] == "-", f[4] == "-"});
        }
        i = j + 1;
    }
    std::sort(nodes.begin(), nodes.end(), [](auto& a, auto& b) { return a.first < b.first; });
    sxg_graph* g = new sxg_graph();
    std::unordered_map<int64_t, uint64_t> rank;
    for (auto& nd : nodes) {
        if (rank.count(nd.first)) { delete g; return fail(SXG_E_INVALID, "duplicate node id " + std::to_string(nd.first)); }
        rank[nd.first] = g->ids.size(); g->ids.push_back(nd.first); g->seq.push_back(nd.second);
    }
    for (auto& pl : plines) {
        g->pname.push_back(pl.first);
        g->steps.emplace_back(); g->pos.emplace_back();
        uint64_t bp = 0;
        const std::string& s = pl.second;
        size_t a = 0;
        for (size_t k = 0; k <= s.size(); ++k)
            if (k == s.size() || s[k] == ',') {
                if (k > a) {
                    const char o = s[k - 1];
                    const int64_t id = atoll(s.substr(a, k - a - 1).c_str());
                    auto it = rank.find(id);
                    if ((o != '+' && o != '-') || it == rank.end()) { delete g; return fail(SXG_E_INVALID, "bad step in path " + pl.first); }
                    g->steps.back().push_back(mk(it->second, o == '-'));
                    g->pos.back().push_back(bp);
                    bp += g->seq[it->second].size();
                }
                a = k + 1;
            }
        g->pos.back().push_back(bp);
    }
    const size_t nn = g->seq.size();
    g->right_of.assign(nn, {}); g->left_of.assign(nn, {}); g->on_node.assign(nn, {}); g->vec_off.assign(nn + 1, 0);
    for (size_t u = 0; u < nn; ++u) g->vec_off[u + 1] = g->vec_off[u] + g->seq[u].size();
    for (auto& l : llines) {
        auto ia = rank.find(l.a), ib = rank.find(l.b);
        if (ia == rank.end() || ib == rank.end()) { delete g; return fail(SXG_E_INVALID, "L line names an unknown segment"); }
        const handle_t A = mk(ia->second, l.ar), Bh = mk(ib->second, l.br);   // edge A -> B  ==  flip(B) -> flip(A)
        if (!rev(A)) g->right_of[nid(A)].push_back(Bh); else g->left_of[nid(A)].push_back(flip(Bh));
        if (!rev(Bh)) g->left_of[nid(Bh)].push_back(A); else g->right_of[nid(Bh)].push_back(flip(A));
    }
    for (size_t p = 0; p < g->steps.size(); ++p)
        for (size_t st = 0; st < g->steps[p].size(); ++st) g->on_node[nid(g->steps[p][st])].emplace_back((uint32_t)p, (uint32_t)st);
    *out = g;
    return SXG_OK;
}
void sxg_graph_free(sxg_graph* g) { delete g; }
int64_t sxg_graph_node_count(const sxg_graph* g) { return g ? (int64_t)g->seq.size() : 0; }
int64_t sxg_graph_path_count(const sxg_graph* g) { return g ? (int64_t)g->pname.size() : 0; }

int sxg_blockset_by_path_windows(const sxg_graph* g, uint64_t target_bp, sxg_blockset** out) {
    if (!g || !out || target_bp == 0) return fail(SXG_E_INVALID, "bad argument");
    sxg_blockset* b = new sxg_blockset();
    for (size_t p = 0; p < g->steps.size(); ++p) {
        const auto& st = g->steps[p];
        size_t s = 0, k = 0;
        while (s < st.size()) {
            size_t e = s;
            uint64_t bp = 0;
            while (e < st.size() && bp < target_bp) { bp += g->seq[nid(st[e])].size(); ++e; }
            if (b->blocks.size() <= k) b->blocks.resize(k + 1);
            b->blocks[k].push_back(path_range_t{p, s, e, bp});
            s = e; ++k;
        }
    }
    for (auto& blk : b->blocks)  // longest first, as src/blocks.cpp:206-219
        std::stable_sort(blk.begin(), blk.end(), [](const path_range_t& a, const path_range_t& c) { return a.length > c.length; });
    *out = b;
    return SXG_OK;
}
void sxg_blockset_free(sxg_blockset* b) { delete b; }
int64_t sxg_blockset_size(const sxg_blockset* b) { return b ? (int64_t)b->blocks.size() : 0; }

int sxg_block_collect_text(const sxg_graph* g, const sxg_blockset* b, int64_t block_id, const sxg_smooth_params* p, char** out_text) {
    if (!g || !b || !p || !out_text || block_id < 0 || block_id >= (int64_t)b->blocks.size()) return fail(SXG_E_INVALID, "bad argument");
    if (int prc = check_params(p)) return prc;
    const collected_t c = collect(*g, b->blocks[block_id], *p);
    std::string o = "padding\t" + std::to_string(c.poa_padding) + "\n";
    for (size_t i = 0; i < c.seqs.size(); ++i) o += "seq\t" + std::to_string(i) + "\t" + std::to_string(c.weights[i]) + "\t" + c.seqs[i] + "\n";
    for (size_t i = 0; i < c.seqs.size(); ++i)
        for (size_t j = 0; j < c.dup_seq_names[i].size(); ++j)
            o += "dup\t" + std::to_string(i) + "\t" + std::to_string(c.dup_rank_in_path_ranges[i][j]) + "\t" + (c.dup_is_revs[i][j] ? "1" : "0") + "\t" +
                 c.dup_seq_names[i][j] + "\n";
    *out_text = dup_out(o);
    return SXG_OK;
}

int sxg_block_graph_gfa(const sxg_graph* g, const sxg_blockset* b, int64_t block_id, const sxg_smooth_params* p, sxg_poa_run_fn run,
                        sxg_poa_free_fn fre, void* ctx, char** out_gfa) {
    if (!g || !b || !p || !run || !out_gfa || block_id < 0 || block_id >= (int64_t)b->blocks.size()) return fail(SXG_E_INVALID, "bad argument");
    if (int prc = check_params(p)) return prc;
    const collected_t c = collect(*g, b->blocks[block_id], *p);
    batch_t B;
    add_to_batch(B, c);
    const sxg_poa_params pp = block_poa_params(*g, b->blocks[block_id], *p);
    sxg_poa_batch_in in;
    memset(&in, 0, sizeof(in));
    in.n_blocks = 1; in.blk_off = B.blk_off.data(); in.seq_off = B.seq_off.data(); in.bases = B.bases.data();
    in.weights = B.weights.data(); in.params = &pp; in.want_consensus = p->add_consensus;
    sxg_poa_batch_out out;
    memset(&out, 0, sizeof(out));
    const int rc = run(ctx, &in, &out);
    if (rc != SXG_OK) { if (fre) fre(&out); return fail(rc, "POA provider failed"); }
    const ograph_t G = block_graph_from_out(c, B, out, 0, cons_name(*p, block_id), p->use_abpoa != 0);
    if (fre) fre(&out);
    *out_gfa = dup_out(to_gfa(G));
    return SXG_OK;
}

static int block_maf_rows(const sxg_graph* g, const sxg_blockset* b, int64_t block_id, const sxg_smooth_params* p, sxg_poa_run_fn run,
                          sxg_poa_free_fn fre, void* ctx, std::vector<maf_row_t>& rows) {
    if (!g || !b || !p || !run || block_id < 0 || block_id >= (int64_t)b->blocks.size()) return fail(SXG_E_INVALID, "bad argument");
    if (int prc = check_params(p)) return prc;
    const collected_t c = collect(*g, b->blocks[block_id], *p);
    rows.clear();
    if (c.seqs.empty()) return SXG_OK;
    batch_t B;
    add_to_batch(B, c);
    const sxg_poa_params pp = block_poa_params(*g, b->blocks[block_id], *p);
    sxg_poa_batch_in in;
    memset(&in, 0, sizeof(in));
    in.n_blocks = 1; in.blk_off = B.blk_off.data(); in.seq_off = B.seq_off.data(); in.bases = B.bases.data();
    in.weights = B.weights.data(); in.params = &pp; in.want_consensus = p->add_consensus; in.want_msa = 1;
    sxg_poa_batch_out out;
    memset(&out, 0, sizeof(out));
    const int rc = run(ctx, &in, &out);
    if (rc != SXG_OK) { if (fre) fre(&out); return fail(rc, "POA provider failed"); }
    if (!out.msa || !out.msa_off || !out.msa_cols) { if (fre) fre(&out); return fail(SXG_E_INVALID, "POA provider returned no MSA"); }
    const size_t nrow = c.seqs.size() + (p->add_consensus ? 1 : 0), cols = (size_t)out.msa_cols[0];
    std::vector<std::string> msa;
    for (size_t r = 0; r < nrow; ++r) msa.emplace_back(out.msa + out.msa_off[0] + r * cols, cols);
    const size_t cons_len = out.cons_off ? (size_t)(out.cons_off[1] - out.cons_off[0]) : 0;
    if (fre) fre(&out);
    rows = maf_rows_from_msa(*g, b->blocks[block_id], c, msa, cons_name(*p, block_id), cons_len);
    return SXG_OK;
}

int sxg_block_maf_rows(const sxg_graph* g, const sxg_blockset* b, int64_t block_id, const sxg_smooth_params* p, sxg_poa_run_fn run,
                       sxg_poa_free_fn fre, void* ctx, char** out_rows) {
    if (!out_rows) return fail(SXG_E_INVALID, "NULL argument");
    std::vector<maf_row_t> rows;
    const int rc = block_maf_rows(g, b, block_id, p, run, fre, ctx, rows);
    if (rc != SXG_OK) return rc;
    *out_rows = dup_out(maf_rows_text(rows));
    return SXG_OK;
}

int sxg_block_maf(const sxg_graph* g, const sxg_blockset* b, int64_t block_id, const sxg_smooth_params* p, sxg_poa_run_fn run,
                  sxg_poa_free_fn fre, void* ctx, char** out_maf) {
    if (!out_maf) return fail(SXG_E_INVALID, "NULL argument");
    std::vector<maf_row_t> rows;
    const int rc = block_maf_rows(g, b, block_id, p, run, fre, ctx, rows);
    if (rc != SXG_OK) return rc;
    *out_maf = dup_out(maf_block_text(rows));
    return SXG_OK;
}

// OpenMP team size: the hardware threads this process may use, capped by the container's CPU quota (cgroup
// cpu.max).  Measured on the GPU box: 256 hardware threads, quota 16 CPUs -- the default team of 256 is
// throttled by CFS and runs the host phases ~2x slower than a team of 16.  OMP_NUM_THREADS overrides.
static int host_threads() {
    static std::atomic<int> cached{0};
    if (cached.load(std::memory_order_relaxed)) return cached.load(std::memory_order_relaxed);
    int n = omp_get_num_procs();
    if (!getenv("OMP_NUM_THREADS")) {
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            long long per = 0;
            if (fscanf(f, "%63s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) {
                const long long cpus = atoll(q) / per;
                if (cpus >= 1 && cpus < n) n = (int)cpus;
            }
            fclose(f);
        }
    } else n = omp_get_max_threads();
    cached.store(n > 0 ? n : 1, std::memory_order_relaxed);
    return cached.load(std::memory_order_relaxed);
}
// The library's parallel regions run with that team; the caller's own OpenMP setting (smoothxg has its -t) is put back
// when the call returns: omp_set_num_threads changes the CALLING thread's nthreads-var, nothing process-wide.
struct OmpTeamGuard {
    int saved;
    explicit OmpTeamGuard(int n) : saved(omp_get_max_threads()) { omp_set_num_threads(n); }
    ~OmpTeamGuard() { omp_set_num_threads(saved); }
};

// one background thread that releases what an iteration leaves behind (see smooth_iteration); SXG_SMOOTH_NO_REAPER=1: inline
struct reaper_t {
    std::thread th;
    std::mutex mu;
    void run(std::function<void()> fn) {
        std::lock_guard<std::mutex> lk(mu);
        if (th.joinable()) th.join();
        if (getenv("SXG_SMOOTH_NO_REAPER")) { fn(); return; }
        try { th = std::thread(fn); } catch (...) { fn(); }   // (thread creation failed: inline)
    }
    ~reaper_t() { if (th.joinable()) th.join(); }
};
static reaper_t g_reaper;
static void reap(std::function<void()> fn) { g_reaper.run(std::move(fn)); }

static int smooth_iteration(const sxg_graph* g, const sxg_blockset* b, const sxg_smooth_params* p, const sxg_merge_params* mp,
                            sxg_poa_run_fn run, sxg_poa_free_fn fre, void* ctx, char** out_gfa, char** out_maf, int64_t* n_flipped) {
    if (!g || !b || !p || !run || !out_gfa) return fail(SXG_E_INVALID, "NULL argument");
    if (int prc = check_params(p)) return prc;
    if (out_maf) *out_maf = nullptr;
    if (n_flipped) *n_flipped = 0;
    const OmpTeamGuard team(host_threads());
    const int64_t nb = (int64_t)b->blocks.size();
    const bool timing = getenv("SXG_SMOOTH_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto T1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[sxg_smooth] %-22s %.3f s\n", what, std::chrono::duration<double>(T1 - T0).count());
        T0 = T1;
    };
    // Phases 1-3 run over CHUNKS of blocks as a three-stage pipeline -- collect(c+1) and block graphs(c-1) on the OpenMP
    // team while the POA provider works on chunk c in a thread of its own -- so that on batches of many small blocks (the
    // reference's default -l 700...1100: thousands of ~1 kbp blocks, where the host phases outweigh the kernels) the GPU and
    // the host cores are busy at the same time.  Blocks are independent (src/smooth.cpp:1931) and their results do not
    // depend on what else is in a batch, so the output is the same for every chunking.  A batch below 2 x
    // SXG_SMOOTH_CHUNK_BLOCKS blocks (default 8192; the headline's 1000 x 64 x 5 kbp) is ONE chunk: one provider call, as before.
    struct frag_t { uint64_t path, start, end; int64_t target, block; };
    // Without the MAF consumer (no flips, no merged consensus paths) the iteration works on compact block graphs -- asked of
    // the provider (the GPU engine builds them on the device), built here from raw results otherwise -- and laces them
    // without materialising the laced graph (lace_fast).  SXG_SMOOTH_LEGACY=1 keeps the ograph_t path for A/B runs and tests.
    const bool fast = !mp && !getenv("SXG_SMOOTH_LEGACY");
    std::vector<collected_t> col((size_t)nb);
    std::vector<ograph_t> graphs(fast ? 0 : (size_t)nb);
    std::vector<cblock_t> cblocks(fast ? (size_t)nb : 0);
    std::vector<omap_t> block_mafs(mp ? (size_t)nb : 0);
    std::vector<char> groom(mp ? (size_t)nb : 0, 0);
    std::unordered_map<std::string, size_t> rank_of;
    for (size_t q = 0; q < g->pname.size(); ++q) rank_of[g->pname[q]] = q;
    struct chunk_t {
        int64_t k0 = 0, k1 = 0;
        batch_t B;
        std::vector<sxg_poa_params> pps;
        std::vector<int32_t> trims;
        uvec<uint32_t> soff, eoff;   // sequence / edge offsets of the chunk's block graphs, block after block (n + 1 entries each)
        bool keep_out = false;   // block graphs of this chunk are views of `out`: released when the text is written
        sxg_poa_batch_in in;
        sxg_poa_batch_out out;
        uint8_t dummy = 0;
        int rc = SXG_OK;
    };
    int64_t chunk_blocks = 8192;   // (round 4: the host phases of a chunk now cost less than what smaller launches lose on the GPU)
    if (const char* e = getenv("SXG_SMOOTH_CHUNK_BLOCKS")) chunk_blocks = std::max<int64_t>(1, atoll(e));
    const int64_t nc = std::max<int64_t>(1, nb / chunk_blocks);
    std::vector<chunk_t> chunks((size_t)nc);
    for (int64_t c = 0; c < nc; ++c) { chunks[(size_t)c].k0 = nb * c / nc; chunks[(size_t)c].k1 = nb * (c + 1) / nc; }
    double t_collect = 0, t_wait = 0, t_graphs = 0;
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    // stage 1: A2-A4 for every block of the chunk, in parallel over blocks as the reference's OpenMP loop
    // (src/smooth.cpp:1931, schedule(dynamic,1)) up to :743; the flat batch is filled in place
    auto prepare = [&](chunk_t& C) {
        const auto t0 = std::chrono::steady_clock::now();
        if (nc == 1) sublap(nullptr);
        const int64_t k0 = C.k0, n = C.k1 - C.k0;
#pragma omp parallel for schedule(dynamic, 1)
        for (int64_t k = k0; k < C.k1; ++k) col[(size_t)k] = collect(*g, b->blocks[(size_t)k], *p);
        if (nc == 1) sublap("collect: sequences");
        batch_t& B = C.B;
        B.blk_off.assign((size_t)n + 1, 0);
        for (int64_t k = 0; k < n; ++k) B.blk_off[(size_t)k + 1] = B.blk_off[(size_t)k] + (int32_t)col[(size_t)(k0 + k)].seqs.size();
        const size_t ns = (size_t)B.blk_off[(size_t)n];
        B.seq_off.assign(ns + 1, 0);
        B.weights.assign(ns, 1);
        for (int64_t k = 0; k < n; ++k)
            for (size_t i = 0; i < col[(size_t)(k0 + k)].seqs.size(); ++i) {
                const size_t sidx = (size_t)B.blk_off[(size_t)k] + i;
                B.seq_off[sidx + 1] = B.seq_off[sidx] + (int64_t)col[(size_t)(k0 + k)].seqs[i].size();
                B.weights[sidx] = col[(size_t)(k0 + k)].weights[i];
            }
        B.bases.resize((size_t)B.seq_off[ns]);
        if (nc == 1) sublap("collect: offsets");
#pragma omp parallel for schedule(dynamic, 1)
        for (int64_t k = 0; k < n; ++k)
            for (size_t i = 0; i < col[(size_t)(k0 + k)].seqs.size(); ++i) {
                const std::string& sq = col[(size_t)(k0 + k)].seqs[i];
                uint8_t* dst = B.bases.data() + B.seq_off[(size_t)B.blk_off[(size_t)k] + i];
                // code_of without a table look-up, so that the loop vectorises (a byte gather does not): bits 1-2 of 'A' 'C' 'G'
                // 'T' are 0 1 3 2, one xor puts G and T in order; every other letter is 4.  (the table loop was 70 % of "collect")
                const uint8_t* src = (const uint8_t*)sq.data();
                const size_t len = sq.size();
#pragma omp simd
                for (size_t x = 0; x < len; ++x) {
                    const uint8_t ch = src[x];
                    uint8_t v = (uint8_t)((ch >> 1) & 3);
                    v = (uint8_t)(v ^ (v >> 1));
                    const bool acgt = (ch == 'A') | (ch == 'C') | (ch == 'G') | (ch == 'T');
                    dst[x] = acgt ? v : (uint8_t)4;
                }
            }
        if (nc == 1) sublap("collect: codes");
        // A14: with -a every block brings its own scores (the engine's per_block_params)
        if (p->adaptive_poa_params) {
            C.pps.resize((size_t)n);
#pragma omp parallel for schedule(dynamic, 1)
            for (int64_t k = 0; k < n; ++k) C.pps[(size_t)k] = block_poa_params(*g, b->blocks[(size_t)(k0 + k)], *p);
        }
        if (C.pps.empty()) C.pps.push_back(poa_params(*p));
        memset(&C.in, 0, sizeof(C.in));
        memset(&C.out, 0, sizeof(C.out));
        C.in.n_blocks = (int32_t)n; C.in.blk_off = B.blk_off.data(); C.in.seq_off = B.seq_off.data();
        C.in.bases = B.bases.empty() ? &C.dummy : B.bases.data(); C.in.weights = B.weights.data(); C.in.params = C.pps.data();
        C.in.per_block_params = p->adaptive_poa_params && n > 0 ? 1 : 0;
        C.in.want_consensus = p->add_consensus;
        C.in.want_msa = mp ? 1 : 0;   // the MAF rows (and with them the merge / flip decisions) need the blocks' MSAs
        if (fast) {   // A9 + A10 asked of the provider: trim = the block's padding, consensus filter of the abPOA path
            C.trims.resize((size_t)std::max<int64_t>(n, 1), 0);
            for (int64_t k = 0; k < n; ++k) C.trims[(size_t)k] = col[(size_t)(k0 + k)].poa_padding;
            C.in.want_block_graph = 3; C.in.bg_trim = C.trims.data(); C.in.bg_consensus_visited_only = p->use_abpoa ? 1 : 0;
        }
        t_collect += since(t0);
    };
    // stage 2: ONE batched POA call per chunk (replaces src/smooth.cpp:752-786 of every block)
    auto call = [&](chunk_t* C) { C->rc = run(ctx, &C->in, &C->out); };
    // stage 3: A9/A10 per block in parallel (the second half of the reference's loop)
    auto finish = [&](chunk_t& C) {
        const auto t0 = std::chrono::steady_clock::now();
        const sxg_poa_batch_out& out = C.out;
        const int64_t k0 = C.k0;
        const bool have_bg = fast && out.bg_node_off && out.bg_node_len && out.bg_node_outdeg && out.bg_node_indeg && out.bg_seq_off &&
                             out.bg_edge_off && out.bg_step_off && (out.bg_seq || out.bg_seq_off[C.k1 - C.k0] == 0);
        if (have_bg) {
            C.keep_out = true;
            const size_t room = (size_t)out.bg_node_off[C.k1 - C.k0] + (size_t)(C.k1 - C.k0);
            C.soff.resize(room); C.eoff.resize(room);
        }
#pragma omp parallel for schedule(dynamic, 1)
        for (int64_t k = C.k0; k < C.k1; ++k) {
            if (col[(size_t)k].seqs.empty()) continue;
            const int64_t slot = k - k0;
            if (fast) {
                const collected_t& c = col[(size_t)k];
                cblock_t& Bk = cblocks[(size_t)k];
                if (have_bg) cblock_from_out(Bk, c, b->blocks[(size_t)k].size(), C.B, out, slot, p->add_consensus != 0,
                                             C.soff.data() + out.bg_node_off[slot] + slot, C.eoff.data() + out.bg_node_off[slot] + slot);
                else {
                    std::vector<const int32_t*> sp;
                    for (int32_t sq = C.B.blk_off[(size_t)slot]; sq < C.B.blk_off[(size_t)slot + 1]; ++sq) sp.push_back(out.seq_path_nodes + C.B.seq_off[(size_t)sq]);
                    const int64_t n0 = out.node_off[slot], nn = out.node_off[slot + 1] - n0;
                    const bool hc = p->add_consensus && out.cons_nodes && out.cons_off;
                    cblock_from_raw(Bk, c, b->blocks[(size_t)k].size(), out.node_code + n0, nn, sp, hc ? out.cons_nodes + out.cons_off[slot] : nullptr,
                                    hc ? out.cons_off[slot + 1] - out.cons_off[slot] : 0, p->add_consensus != 0, p->use_abpoa != 0);
                }
                // With several chunks in flight this stage runs while the provider -- on a sharded run: every rank's GPU -- works
                // on the next chunk: the block's share of the validation (src/main.cpp:770-810) is done here, off the serial
                // tail behind the last chunk.  (One chunk: nothing to overlap with, lace_fast does it.)
                if (nc > 1) Bk.validated = block_spells_its_ranges(g, b->blocks[(size_t)k], Bk) ? 1 : -1;
                continue;   // (the padded sequences go with everything else when the iteration is over)
            }
            graphs[(size_t)k] = block_graph_from_out(col[(size_t)k], C.B, out, slot, cons_name(*p, k), p->use_abpoa != 0);
            if (mp) {   // MSA -> MAF rows of the block (src/smooth.cpp:782-905), and its grooming orientation (:1826-1842)
                const collected_t& c = col[(size_t)k];
                const size_t nrow = c.seqs.size() + (p->add_consensus ? 1 : 0), cols = (size_t)out.msa_cols[slot];
                std::vector<std::string> msa;
                for (size_t r = 0; r < nrow; ++r) msa.emplace_back(out.msa + out.msa_off[slot] + r * cols, cols);
                const size_t cons_len = out.cons_off ? (size_t)(out.cons_off[slot + 1] - out.cons_off[slot]) : 0;
                block_mafs[(size_t)k] = maf_block_map(maf_rows_from_msa(*g, b->blocks[(size_t)k], c, msa, cons_name(*p, k), cons_len));
                groom[(size_t)k] = groom_flip(rank_of, graphs[(size_t)k], cons_name(*p, k)) ? 1 : 0;
            }
            collected_t().seqs.swap(col[(size_t)k].seqs);   // the padded sequences are not needed any more
        }
        t_graphs += since(t0);
    };
    {
        bool not_root = false;
        int fail_rc = SXG_OK;
        std::string fail_msg;
        std::thread worker;
        struct joiner_t { std::thread& t; ~joiner_t() { if (t.joinable()) t.join(); } } joiner{worker};   // (also when an exception unwinds: see guarded())
        prepare(chunks[0]);
        if (nc > 1) worker = std::thread(call, &chunks[0]); else call(&chunks[0]);
        for (int64_t c = 0; c < nc; ++c) {
            chunk_t& C = chunks[(size_t)c];
            if (c + 1 < nc) prepare(chunks[(size_t)c + 1]);
            const auto tw = std::chrono::steady_clock::now();
            if (worker.joinable()) worker.join();
            t_wait += since(tw);
            // (every chunk reaches the provider even after a failure or on a rank that does not lace: a sharded provider is
            //  a collective, the other ranks are in the same call)
            if (c + 1 < nc) worker = std::thread(call, &chunks[(size_t)c + 1]);
            if (C.rc == SXG_NOT_ROOT) not_root = true;   // multi-GPU provider (sxg_poa_batch_run_sharded) on a rank that does not lace
            else if (C.rc != SXG_OK) { if (fail_rc == SXG_OK) { fail_rc = C.rc; fail_msg = "POA provider failed"; } }
            else if (mp && C.k1 > C.k0 && (!C.out.msa || !C.out.msa_off || !C.out.msa_cols)) { if (fail_rc == SXG_OK) { fail_rc = SXG_E_INVALID; fail_msg = "POA provider returned no MSA"; } }
            else if (fast && C.k1 > C.k0 && !C.out.bg_node_off && (!C.out.seq_path_nodes || !C.out.node_off || !C.out.node_code)) {
                if (fail_rc == SXG_OK) { fail_rc = SXG_E_INVALID; fail_msg = "POA provider returned neither block graphs nor per-base paths"; }
            }
            else if (fail_rc == SXG_OK && !not_root) finish(C);
            if (fre && !C.keep_out) fre(&C.out);
            batch_t().bases.swap(C.B.bases);
        }
        auto release_outs = [&]() { for (auto& C : chunks) if (C.keep_out && fre) { fre(&C.out); C.keep_out = false; } };
        if (fail_rc != SXG_OK) { release_outs(); return fail(fail_rc, fail_msg); }
        if (not_root) { release_outs(); return fail(SXG_NOT_ROOT, "not the lacing rank"); }
    }
    if (timing) {
        fprintf(stderr, "[sxg_smooth] %-22s %.3f s  (%lld chunk%s; collect %.3f s, waiting for the POA provider %.3f s, block graphs %.3f s)\n",
                "collect|POA|graphs", since(T0), (long long)nc, nc == 1 ? "" : "s, pipelined", t_collect, t_wait, t_graphs);
        T0 = std::chrono::steady_clock::now();
    }
    if (fast) {
        const int rc = lace_fast(g, b, p, cblocks, out_gfa, lap);
        // The providers' results go back to the provider HERE, on the caller's thread, before the call returns: the free
        // callback and its context need only live for the call (sxg_poa_free_fn, include/sxg_smooth.h), and a pinned
        // download buffer is back in its pool before the next iteration asks for one.  What the library itself owns -- the
        // compact block graphs (views, never dereferenced again), the collected sequences, the chunks' host arrays -- is
        // released behind the caller's back: unmapping it is on nobody's critical path.  The reaper is joined by the next
        // iteration (and when the library is unloaded); without a thread the release happens inline.
        for (auto& C : chunks) if (C.keep_out && fre) { fre(&C.out); C.keep_out = false; }
        {
            struct grave_t { std::vector<cblock_t> cb; std::vector<chunk_t> ch; std::vector<collected_t> col; };
            grave_t* graveyard = nullptr;
            try {
                graveyard = new grave_t();
                graveyard->cb.swap(cblocks);
                graveyard->ch.swap(chunks);
                graveyard->col.swap(col);
                reap([graveyard]() { delete graveyard; });
            } catch (...) {   // (no memory for the grave, no thread: release inline -- a finished iteration stays finished)
                delete graveyard;
            }
        }
        lap("teardown");
        return rc;
    }
    // the in-order MAF consumer: merges contiguous blocks, decides which block graphs get flipped (-M), writes the MAF
    merge_state_t mstate;
    if (mp) {
        std::vector<char> flips;
        const std::string maf = merge_maf_blocks(block_mafs, groom, mp->merge_blocks != 0, mp->contiguous_path_jaccard, p->add_consensus != 0,
                                                 p->consensus_base_name ? p->consensus_base_name : "Consensus_",
                                                 mp->max_merged_groups_in_memory ? (size_t)mp->max_merged_groups_in_memory : 50,
                                                 mp->preserve_unmerged_consensus != 0, mp->maf_header, flips, mstate);
        int64_t nf = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : nf)
        for (int64_t k = 0; k < nb; ++k)
            if (flips[(size_t)k] && !graphs[(size_t)k].seq.empty()) { flip_block_graph(graphs[(size_t)k], cons_name(*p, k)); ++nf; }   // A13
        if (n_flipped) *n_flipped = nf;
        if (out_maf) *out_maf = dup_out(maf);
        lap("MAF merge + flips");
    }
    std::vector<frag_t> mapping;
    for (int64_t k = 0; k < nb; ++k) {
        if (graphs[(size_t)k].seq.empty()) continue;
        int64_t path_id = 0;
        for (auto& r : b->blocks[(size_t)k]) mapping.push_back(frag_t{r.path, g->pos[r.path][r.begin], g->pos[r.path][r.end], path_id++, k});
    }
    lap("block graphs");
    // lacing (src/main.cpp:599-764): fragments by (path, start); blocks concatenated with an id offset
    std::stable_sort(mapping.begin(), mapping.end(), [](const frag_t& a, const frag_t& c) { return a.path < c.path || (a.path == c.path && a.start < c.start); });
    ograph_t S;
    std::vector<uint64_t> id_trans((size_t)nb + 1, 0), e_trans((size_t)nb + 1, 0);
    for (int64_t k = 0; k < nb; ++k) {
        id_trans[(size_t)k + 1] = id_trans[(size_t)k] + graphs[(size_t)k].seq.size();
        e_trans[(size_t)k + 1] = e_trans[(size_t)k] + graphs[(size_t)k].edges.size();
    }
    S.seq.resize((size_t)id_trans[(size_t)nb]);
    S.edges.resize((size_t)e_trans[(size_t)nb]);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t k = 0; k < nb; ++k) {
        ograph_t& Gk = graphs[(size_t)k];
        const uint64_t o = id_trans[(size_t)k];
        for (size_t x = 0; x < Gk.seq.size(); ++x) S.seq[(size_t)o + x].swap(Gk.seq[x]);
        // (block-local edges are canonical and sorted; the offset keeps both, and blocks are disjoint)
        for (size_t x = 0; x < Gk.edges.size(); ++x)
            S.edges[(size_t)e_trans[(size_t)k] + x] = edge_t(mk(nid(Gk.edges[x].first) + o, rev(Gk.edges[x].first)),
                                                              mk(nid(Gk.edges[x].second) + o, rev(Gk.edges[x].second)));
    }
    // one laced path per input path: its fragments in order (parallel over paths, src/main.cpp:706-764)
    std::vector<std::pair<size_t, size_t>> runs;   // [a, z) of every path's fragments
    for (size_t a = 0; a < mapping.size();) {
        size_t z = a;
        while (z < mapping.size() && mapping[z].path == mapping[a].path) ++z;
        runs.emplace_back(a, z);
        a = z;
    }
    S.paths.resize(runs.size());
    std::vector<std::string> errs(runs.size());
    std::vector<std::vector<edge_t>> links(runs.size());   // edges between the fragments of a path
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t q = 0; q < (int64_t)runs.size(); ++q) {
        const size_t a = runs[(size_t)q].first, z = runs[(size_t)q].second;
        size_t total = 0;
        for (size_t f = a; f < z; ++f) total += graphs[(size_t)mapping[f].block].paths[(size_t)mapping[f].target].second.size();
        steps_t steps;
        steps.reserve(total);
        uint64_t last_end = 0;
        for (size_t f = a; f < z; ++f) {
            if (mapping[f].start != last_end) { errs[(size_t)q] = "path " + g->pname[mapping[a].path] + " is not covered by the blocks"; break; }
            const auto& bp = graphs[(size_t)mapping[f].block].paths[(size_t)mapping[f].target].second;
            const uint64_t o = id_trans[(size_t)mapping[f].block];
            if (!steps.empty() && !bp.empty()) links[(size_t)q].push_back(ograph_t::canon(steps.back(), mk(nid(bp.front()) + o, rev(bp.front()))));
            for (handle_t h : bp) steps.push_back(mk(nid(h) + o, rev(h)));
            last_end = mapping[f].end;
        }
        if (errs[(size_t)q].empty() && last_end != g->pos[mapping[a].path].back())
            errs[(size_t)q] = "path " + g->pname[mapping[a].path] + " is not covered to its end";
        S.paths[(size_t)q].first = g->pname[mapping[a].path];
        S.paths[(size_t)q].second.swap(steps);
    }
    for (auto& e : errs) if (!e.empty()) return fail(SXG_E_INVALID, e);
    lap("lacing");
    // validation (src/main.cpp:770-810)
    {
        size_t nonempty = 0;
        for (auto& st : g->steps) if (!st.empty()) ++nonempty;
        if (S.paths.size() != nonempty) return fail(SXG_E_INVALID, "path count mismatch between input and smoothed graph");
        std::unordered_map<std::string, size_t> byname;
        for (size_t q = 0; q < g->pname.size(); ++q) byname[g->pname[q]] = q;
        std::vector<char> bad(S.paths.size(), 0);
#pragma omp parallel for schedule(dynamic, 1)
        for (int64_t q = 0; q < (int64_t)S.paths.size(); ++q) {
            const auto& sp = S.paths[(size_t)q];
            std::string s;
            s.reserve((size_t)g->pos[byname[sp.first]].back());
            for (handle_t h : sp.second) { if (rev(h)) s += revcomp(S.seq[nid(h)]); else s += S.seq[nid(h)]; }
            if (s != g->path_sequence(byname[sp.first])) bad[(size_t)q] = 1;
        }
        for (size_t q = 0; q < bad.size(); ++q)
            if (bad[q]) return fail(SXG_E_INVALID, "path " + S.paths[q].first + " was corrupted in the smoothed graph");
    }
    lap("validation");
    // consensus paths (src/main.cpp:812-986): the blocks' own consensus paths -- except, unless they are to be
    // preserved, those of blocks that went into a merged group -- then one path per merged group that strings the
    // member blocks' consensus paths together in the group's order
    if (p->add_consensus) {
        auto cons_steps = [&](int64_t k, steps_t& steps) {
            if (graphs[(size_t)k].paths.empty()) return;
            for (handle_t h : graphs[(size_t)k].paths.back().second) steps.push_back(mk(nid(h) + id_trans[(size_t)k], rev(h)));
        };
        const bool preserve = !mp || mp->preserve_unmerged_consensus != 0;
        for (int64_t k = 0; k < nb; ++k) {
            if (graphs[(size_t)k].paths.empty()) continue;
            if (mp && !mstate.groups.empty() && !preserve && mstate.in_merged[(size_t)k]) continue;
            steps_t steps;
            cons_steps(k, steps);
            S.paths.emplace_back(graphs[(size_t)k].paths.back().first, steps);
        }
        if (mp)
            for (auto& grp : mstate.groups) {
                std::vector<std::pair<uint64_t, uint64_t>> iv = grp.intervals;   // [start, end)
                std::sort(iv.begin(), iv.end());
                steps_t steps;
                if (!grp.inverted) { for (auto& x : iv) for (uint64_t k = x.first; k < x.second; ++k) cons_steps((int64_t)k, steps); }
                else for (size_t j = iv.size(); j-- > 0;) for (uint64_t k = iv[j].second; k-- > iv[j].first;) cons_steps((int64_t)k, steps);
                // (a merged consensus steps from one block's consensus into the next: edges the blocks do not hold)
                links.emplace_back();
                for (size_t x = 1; x < steps.size(); ++x) links.back().push_back(ograph_t::canon(steps[x - 1], steps[x]));
                S.paths.emplace_back(std::string(p->consensus_base_name ? p->consensus_base_name : "Consensus_") + grp.ranges, steps);
            }
    }
    sublap(nullptr);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t k = 0; k < nb; ++k) { ograph_t none; std::swap(none, graphs[(size_t)k]); }
    std::vector<ograph_t>().swap(graphs);
    sublap("free block graphs");
    // walk every path and make sure its edges exist (src/main.cpp:1002-1016): inside a block they do by
    // construction (A10 keeps exactly the path-supported edges), so only the links between fragments are new
    {
        std::vector<edge_t> lk;
        for (auto& l : links) lk.insert(lk.end(), l.begin(), l.end());
        std::sort(lk.begin(), lk.end());
        lk.erase(std::unique(lk.begin(), lk.end()), lk.end());
        // (the block edges are sorted and unique: canonical and sorted inside a block, blocks at increasing id offsets.
        //  The few links -- one per fragment boundary -- are merged in by position: chunks of the block edges are
        //  copied in parallel, each preceded by the new links that sort in front of its elements.)
        std::vector<size_t> pos;   // new link j goes in front of block edge pos[j]
        {
            size_t w = 0;
            for (size_t j = 0; j < lk.size(); ++j) {
                const size_t at = (size_t)(std::lower_bound(S.edges.begin(), S.edges.end(), lk[j]) - S.edges.begin());
                if (at < S.edges.size() && S.edges[at] == lk[j]) continue;   // the blocks hold it already
                lk[w++] = lk[j];
                pos.push_back(at);
            }
            lk.resize(w);
        }
        if (!lk.empty()) {
            const int64_t ne0 = (int64_t)S.edges.size(), CHK = 1 << 16, nch = std::max<int64_t>(1, (ne0 + CHK - 1) / CHK);
            uvec<edge_t> merged((size_t)ne0 + lk.size());
#pragma omp parallel for schedule(static)
            for (int64_t q = 0; q < nch; ++q) {
                const int64_t lo = q * CHK, hi = q + 1 == nch ? ne0 : (q + 1) * CHK;
                size_t j = (size_t)(std::lower_bound(pos.begin(), pos.end(), (size_t)lo) - pos.begin());
                size_t w = (size_t)lo + j;
                for (int64_t x = lo; x < hi; ++x) {
                    while (j < pos.size() && pos[j] == (size_t)x) merged[w++] = lk[j++];
                    merged[w++] = S.edges[(size_t)x];
                }
                if (q + 1 == nch) while (j < pos.size()) merged[w++] = lk[j++];   // links behind every block edge
            }
            S.edges.swap(merged);
        }
    }
    sublap("link edges");
    unchop(S);          // :1021
    sublap(nullptr);
    lap("unchop");
    *out_gfa = to_gfa_c(S, nullptr, true);   // (S is not used after its text)
    lap("GFA text");
    if (!*out_gfa) return fail(SXG_E_NOMEM, "out of memory for the GFA text");
    if (timing) {   // (the destructors, run here so that they show up as a phase)
        sublap(nullptr);
        { uvec<std::string> x; x.swap(S.seq); }
        sublap("free S.seq");
        { decltype(S.paths) x; x.swap(S.paths); }
        sublap("free S.paths");
        { decltype(S.edges) x; x.swap(S.edges); }
        sublap("free S.edges");
        { std::vector<collected_t> x; x.swap(col); }
        sublap("free collected");
        { std::vector<chunk_t> x; x.swap(chunks); }
        sublap("free batch");
        { std::vector<frag_t> x; x.swap(mapping); }
        lap("teardown");
    }
    return SXG_OK;
}

// ---------------------------------------------------------------------------------------------
// Block discovery: smoothable_blocks (src/blocks.cpp:7-327).  A greedy sweep over the nodes in rank order
// collects handles until the block would outgrow its weight / path-length / edge-jump limits, turns the
// unseen steps on those handles into path ranges (contiguous within max_path_jump, broken at steps that
// an earlier block took), and splits the result into the weakly connected components of its path
// adjacencies.  Restated line by line; two decrees: ranges are ordered with a STABLE sort (the reference's
// std::sort leaves the order of equal lengths to the library), and XG's node order is id order.
int sxg_blockset_smoothable(const sxg_graph* g, uint64_t max_block_weight, uint64_t max_block_path_length, uint64_t max_path_jump,
                            uint64_t max_edge_jump, int order_paths_from_longest, sxg_blockset** out) {
    if (!g || !out) return fail(SXG_E_INVALID, "NULL argument");
    sxg_blockset* bs = new sxg_blockset();
    const size_t np = g->steps.size(), nn = g->seq.size();
    std::vector<std::vector<char>> seen(np);
    for (size_t p = 0; p < np; ++p) seen[p].assign(g->steps[p].size(), 0);
    typedef std::pair<uint32_t, uint32_t> step_t;   // (path, step rank)
    std::vector<uint64_t> block_handles;            // node ranks
    auto toposplit = [&](const std::vector<path_range_t>& ranges) {   // :45-107
        std::unordered_map<uint64_t, uint64_t> id_to_entry;
        for (auto& r : ranges)
            for (uint64_t st = r.begin; st != r.end; ++st) {
                const uint64_t id = nid(g->steps[r.path][st]);
                if (!id_to_entry.count(id)) { const uint64_t e = id_to_entry.size(); id_to_entry[id] = e; }
            }
        std::vector<uint64_t> parent(id_to_entry.size());
        for (size_t k = 0; k < parent.size(); ++k) parent[k] = k;
        auto find = [&](uint64_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
        for (auto& r : ranges)
            for (uint64_t st = r.begin; st + 1 < r.end; ++st) {
                const uint64_t a = find(id_to_entry[nid(g->steps[r.path][st])]), b = find(id_to_entry[nid(g->steps[r.path][st + 1])]);
                if (a != b) parent[a < b ? b : a] = a < b ? a : b;
            }
        std::unordered_map<uint64_t, uint64_t> dset_ids;
        std::vector<std::vector<path_range_t>> blocks;
        for (auto& r : ranges)   // sets are numbered in the order the ranges' steps meet them (:80-91)
            for (uint64_t st = r.begin; st != r.end; ++st) {
                const uint64_t d = find(id_to_entry[nid(g->steps[r.path][st])]);
                if (!dset_ids.count(d)) { const uint64_t k = dset_ids.size(); dset_ids[d] = k; blocks.emplace_back(); }
            }
        for (auto& r : ranges) blocks[dset_ids[find(id_to_entry[nid(g->steps[r.path][r.begin])])]].push_back(r);
        return blocks;
    };
    auto finalize_block = [&]() {   // :109-231
        std::vector<step_t> traversals;
        for (uint64_t u : block_handles)
            for (auto& st : g->on_node[u]) if (!seen[st.first][st.second]) traversals.push_back(st);
        block_handles.clear();
        std::sort(traversals.begin(), traversals.end());
        struct span_t { uint32_t path; uint64_t begin, last; };   // closed span of step ranks
        std::vector<span_t> spans;
        for (auto& st : traversals) {
            if (spans.empty()) { spans.push_back(span_t{st.first, st.second, st.second}); continue; }
            span_t& sp = spans.back();
            const uint64_t last_end_bp = g->pos[sp.path][sp.last] + g->seq[nid(g->steps[sp.path][sp.last])].size();
            if (sp.path != st.first || g->pos[st.first][st.second] - last_end_bp > max_path_jump) spans.push_back(span_t{st.first, st.second, st.second});
            else sp.last = st.second;
        }
        std::vector<path_range_t> ranges;
        for (auto& sp : spans) {     // break the spans on steps an earlier block has taken (:151-174)
            const uint64_t end = sp.last + 1;
            bool open = false;
            for (uint64_t cur = sp.begin; cur != end; ++cur) {
                if (!open) { ranges.push_back(path_range_t{sp.path, cur, cur, 0}); open = true; }
                ranges.back().end = cur;
                if (seen[sp.path][cur]) open = false;
            }
            if (open) ranges.back().end = end;
        }
        ranges.erase(std::remove_if(ranges.begin(), ranges.end(), [](const path_range_t& r) { return r.begin == r.end; }), ranges.end());
        uint64_t total = 0;
        for (auto& r : ranges) {
            r.length = 0;
            for (uint64_t st = r.begin; st != r.end; ++st) { seen[r.path][st] = 1; r.length += g->seq[nid(g->steps[r.path][st])].size(); }
            total += r.length;
        }
        if (total > 0) {
            if (order_paths_from_longest) std::stable_sort(ranges.begin(), ranges.end(), [](const path_range_t& a, const path_range_t& b) { return a.length > b.length; });
            else std::stable_sort(ranges.begin(), ranges.end(), [](const path_range_t& a, const path_range_t& b) { return a.length < b.length; });
            for (auto& split : toposplit(ranges)) bs->blocks.push_back(split);
        }
    };
    uint64_t total_path_length = 0;
    std::unordered_map<uint32_t, std::pair<uint64_t, uint64_t>> path_coverage;
    for (size_t u = 0; u < nn; ++u) {   // :239-316
        const int64_t handle_length = (int64_t)g->seq[u].size();
        uint64_t sequence_to_add = 0;
        for (auto& st : g->on_node[u]) if (!seen[st.first][st.second]) sequence_to_add += (uint64_t)handle_length;
        uint64_t max_path_length = 0;
        for (auto& pc : path_coverage) {
            const double div = pc.second.second < block_handles.size() ? 1.0 : (double)pc.second.second / (double)block_handles.size();
            const uint64_t est = (uint64_t)std::round((double)pc.second.first / div);
            max_path_length = std::max<uint64_t>(est + (uint64_t)handle_length, max_path_length);
        }
        int64_t longest_edge_jump = 0;
        const int64_t off = (int64_t)g->vec_off[u];
        for (handle_t o : g->right_of[u]) {
            const int64_t other = (int64_t)g->vec_off[nid(o)] + (rev(o) ? (int64_t)g->seq[nid(o)].size() : 0);
            longest_edge_jump = std::max<int64_t>(longest_edge_jump, std::llabs(other - (off + handle_length)));
        }
        for (handle_t o : g->left_of[u]) {
            const int64_t other = (int64_t)g->vec_off[nid(o)] + (rev(o) ? 0 : (int64_t)g->seq[nid(o)].size());
            longest_edge_jump = std::max<int64_t>(longest_edge_jump, std::llabs(other - off));
        }
        if (!block_handles.empty() &&
            (total_path_length + sequence_to_add > max_block_weight || (max_edge_jump && (uint64_t)longest_edge_jump > max_edge_jump) ||
             max_path_length > max_block_path_length)) {
            finalize_block();
            total_path_length = 0;
            path_coverage.clear();
        }
        total_path_length += sequence_to_add;
        for (auto& st : g->on_node[u])
            if (!seen[st.first][st.second]) { path_coverage[st.first].first += (uint64_t)handle_length; path_coverage[st.first].second++; }
        block_handles.push_back(u);
    }
    finalize_block();   // :322-324
    *out = bs;
    return SXG_OK;
}

// sautocorr::repeat(vec, min_copy, max_copy, min_copy, min_z, stride) as src/breaks.cpp:236-245 calls it -- BY DECREE
// (ekg/sautocorr is an un-vendored dependency, absent from the reference snapshot; DESIGN.md section 9, mirrored by
// oracle/smooth_oracle.py::repeat_length): for every lag L in [min_copy, min(max_copy, n - min_copy)] the autocorrelation
// is the fraction of the sampled positions i = 0, stride, 2 stride, ... < n - L whose letter comes back L bases later; a
// lag whose z-score over all lags is at least min_z is a repeat, the repeat's length is the FIRST lag of greatest z.
// Returns 0 when there is none.  Sums run in lag order in double precision, as in the Python restatement.
static double repeat_length(const std::string& seq, uint64_t min_copy, uint64_t max_copy, double min_z, uint64_t stride) {
    const uint64_t n = seq.size();
    if (min_copy == 0 || stride == 0 || n < 2 * min_copy) return 0.0;
    const uint64_t hi = std::min<uint64_t>(max_copy, n - min_copy);
    if (hi < min_copy) return 0.0;
    std::vector<double> r;
    r.reserve((size_t)(hi - min_copy + 1));
    for (uint64_t L = min_copy; L <= hi; ++L) {
        uint64_t match = 0, cnt = 0;
        for (uint64_t i = 0; i < n - L; i += stride) { match += seq[(size_t)i] == seq[(size_t)(i + L)] ? 1 : 0; ++cnt; }
        r.push_back((double)match / (double)cnt);
    }
    double mean = 0.0;
    for (double v : r) mean += v;
    mean /= (double)r.size();
    double var = 0.0;
    for (double v : r) var += (v - mean) * (v - mean);
    const double sd = std::sqrt(var / (double)r.size());
    if (sd == 0.0) return 0.0;
    int64_t best = -1;
    double best_z = 0.0;
    for (size_t k = 0; k < r.size(); ++k) {
        const double z = (r[k] - mean) / sd;
        if (best < 0 || z > best_z) { best = (int64_t)k; best_z = z; }
    }
    return best_z >= min_z ? (double)(min_copy + (uint64_t)best) : 0.0;
}

}  // extern "C"
// The cutting half of break_blocks (src/breaks.cpp:210-330): a block with more than one range and a range longer than
// max_poa_length is cut -- at half the mean length of the repeats its ranges hold when there are any (:224-272: the
// reference always asks for this, break_repeats = true at src/main.cpp:476), every range of it; otherwise blindly, the
// ranges of at least max_poa_length bases into pieces of just over that (a piece is closed by the step that takes it past
// the limit) -- and re-ordered by length.  Splitting by identity (:335+) stays off as in the defaults
// (block_group_identity = 0, src/main.cpp:316-320).
static int blockset_break(const sxg_graph* g, const sxg_blockset* in, uint64_t max_poa_length, bool break_repeats, uint64_t min_copy_length,
                          uint64_t max_copy_length, double min_autocorr_z, uint64_t autocorr_stride, int order_paths_from_longest, sxg_blockset** out) {
    if (!g || !in || !out) return fail(SXG_E_INVALID, "NULL argument");
    if (break_repeats && (min_copy_length == 0 || autocorr_stride == 0 || max_copy_length < min_copy_length)) return fail(SXG_E_INVALID, "bad repeat parameters");
    sxg_blockset* bs = new sxg_blockset();
    const int64_t nb = (int64_t)in->blocks.size();
    bs->blocks.resize((size_t)nb);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t k = 0; k < nb; ++k) {
        const auto& blk = in->blocks[(size_t)k];
        bool to_break = false;
        for (auto& r : blk) if (r.length > max_poa_length) { to_break = true; break; }
        if (!(blk.size() > 1 && to_break)) { bs->blocks[(size_t)k] = blk; continue; }
        uint64_t cut_length = max_poa_length;
        bool found_repeat = false;
        if (break_repeats) {
            std::vector<double> lengths;
            for (auto& r : blk) {
                std::string seq;
                for (uint64_t st = r.begin; st != r.end; ++st) seq += g->sequence(g->steps[r.path][st]);
                if (seq.size() >= 2 * min_copy_length) {
                    const double rl = repeat_length(seq, min_copy_length, max_copy_length, min_autocorr_z, autocorr_stride);
                    if (rl > 0) lengths.push_back(rl);
                }
            }
            if (!lengths.empty()) {
                double total = 0.0;
                for (double v : lengths) total += v;
                found_repeat = true;
                cut_length = (uint64_t)std::round(total / (double)lengths.size() / 2.0);
            }
        }
        std::vector<path_range_t> chopped;
        for (auto& r : blk) {
            if (!found_repeat && r.length < cut_length) { chopped.push_back(r); continue; }
            uint64_t last_cut = 0, last_end = r.begin, pos = 0, st;
            for (st = r.begin; st != r.end; ++st) {
                pos += g->seq[nid(g->steps[r.path][st])].size();
                if (pos - last_cut > cut_length) {
                    chopped.push_back(path_range_t{r.path, last_end, st + 1, pos - last_cut});
                    last_end = st + 1;
                    last_cut = pos;
                }
            }
            if (st != last_end) chopped.push_back(path_range_t{r.path, last_end, st, pos - last_cut});
        }
        if (order_paths_from_longest) std::stable_sort(chopped.begin(), chopped.end(), [](const path_range_t& a, const path_range_t& b) { return a.length > b.length; });
        else std::stable_sort(chopped.begin(), chopped.end(), [](const path_range_t& a, const path_range_t& b) { return a.length < b.length; });
        bs->blocks[(size_t)k] = chopped;
    }
    *out = bs;
    return SXG_OK;
}
extern "C" {

// ... with the reference's own repeat parameters (min_copy_length 1000, max_copy_length 20000: src/main.cpp:285-286;
// min_autocorr_z 5, autocorr_stride 50: :457-458)
int sxg_blockset_break(const sxg_graph* g, const sxg_blockset* in, uint64_t max_poa_length, int order_paths_from_longest, sxg_blockset** out) {
    return blockset_break(g, in, max_poa_length, true, 1000, 20000, 5.0, 50, order_paths_from_longest, out);
}
int sxg_blockset_break_ex(const sxg_graph* g, const sxg_blockset* in, uint64_t max_poa_length, int break_repeats, uint64_t min_copy_length,
                          uint64_t max_copy_length, double min_autocorr_z, uint64_t autocorr_stride, int order_paths_from_longest, sxg_blockset** out) {
    return blockset_break(g, in, max_poa_length, break_repeats != 0, min_copy_length, max_copy_length, min_autocorr_z, autocorr_stride, order_paths_from_longest, out);
}

}  // extern "C"
// no exception crosses the C ABI: an allocation that fails anywhere in the iteration becomes SXG_E_NOMEM (the pipeline's
// worker thread is joined while the stack unwinds)
template <class F> static int guarded(F f) {
    try { return f(); }
    catch (const std::bad_alloc&) { return fail(SXG_E_NOMEM, "out of host memory in the smoothing iteration"); }
    catch (const std::exception& e) { return fail(SXG_E_INVALID, std::string("internal error: ") + e.what()); }
}
extern "C" {

int sxg_smooth_gfa(const sxg_graph* g, const sxg_blockset* b, const sxg_smooth_params* p, sxg_poa_run_fn run, sxg_poa_free_fn fre, void* ctx,
                   char** out_gfa) {
    return guarded([&] { return smooth_iteration(g, b, p, nullptr, run, fre, ctx, out_gfa, nullptr, nullptr); });
}

void sxg_merge_default_params(sxg_merge_params* mp) {
    if (!mp) return;
    mp->merge_blocks = 0; mp->contiguous_path_jaccard = 1.0; mp->preserve_unmerged_consensus = 0;   // src/main.cpp:282, -M / -J / -N
    mp->max_merged_groups_in_memory = 50; mp->maf_header = nullptr;                                  // src/main.cpp:297-298
}

int sxg_smooth_maf_gfa(const sxg_graph* g, const sxg_blockset* b, const sxg_smooth_params* p, const sxg_merge_params* mp, sxg_poa_run_fn run,
                       sxg_poa_free_fn fre, void* ctx, char** out_gfa, char** out_maf, int64_t* n_flipped) {
    if (!mp || !out_maf) return fail(SXG_E_INVALID, "NULL argument");
    return guarded([&] { return smooth_iteration(g, b, p, mp, run, fre, ctx, out_gfa, out_maf, n_flipped); });
}

// blockset_t from the caller's own blocks (src/blocks.hpp:29-43,70-120): block k owns ranges
// blk_off[k] .. blk_off[k+1]-1, each {path rank, first step, one-past-last step, length in bp}, in the order
// they are to be aligned (smoothxg: longest first, src/blocks.cpp:206-219 -- kept as given).
int sxg_blockset_from_ranges(const sxg_graph* g, int64_t n_blocks, const int64_t* blk_off, const sxg_path_range* ranges, sxg_blockset** out) {
    if (!g || !out || n_blocks < 0 || (n_blocks > 0 && (!blk_off || !ranges))) return fail(SXG_E_INVALID, "bad argument");
    *out = nullptr;
    if (n_blocks > 0 && blk_off[0] != 0) return fail(SXG_E_INVALID, "blk_off[0] must be 0");
    sxg_blockset* b = new sxg_blockset();
    b->blocks.resize((size_t)n_blocks);
    for (int64_t k = 0; k < n_blocks; ++k) {
        if (blk_off[k + 1] < blk_off[k]) { delete b; return fail(SXG_E_INVALID, "blk_off not monotone"); }
        for (int64_t r = blk_off[k]; r < blk_off[k + 1]; ++r) {
            const sxg_path_range& pr = ranges[r];
            if (pr.path < 0 || pr.path >= (int64_t)g->steps.size() || pr.step_begin < 0 || pr.step_end < pr.step_begin ||
                pr.step_end > (int64_t)g->steps[(size_t)pr.path].size()) {
                delete b;
                return fail(SXG_E_INVALID, "range " + std::to_string(r) + " is outside its path");
            }
            const uint64_t bp = g->pos[(size_t)pr.path][(size_t)pr.step_end] - g->pos[(size_t)pr.path][(size_t)pr.step_begin];
            if (pr.length != 0 && (uint64_t)pr.length != bp) { delete b; return fail(SXG_E_INVALID, "range " + std::to_string(r) + ": length does not match its steps"); }
            b->blocks[(size_t)k].push_back(path_range_t{(uint64_t)pr.path, (uint64_t)pr.step_begin, (uint64_t)pr.step_end, bp});
        }
    }
    *out = b;
    return SXG_OK;
}

int64_t sxg_blockset_block_size(const sxg_blockset* b, int64_t block_id) {
    if (!b || block_id < 0 || block_id >= (int64_t)b->blocks.size()) return -1;
    return (int64_t)b->blocks[(size_t)block_id].size();
}
int sxg_blockset_block_ranges(const sxg_blockset* b, int64_t block_id, sxg_path_range* out) {
    if (!b || !out || block_id < 0 || block_id >= (int64_t)b->blocks.size()) return fail(SXG_E_INVALID, "bad argument");
    size_t k = 0;
    for (auto& r : b->blocks[(size_t)block_id]) out[k++] = sxg_path_range{(int64_t)r.path, (int64_t)r.begin, (int64_t)r.end, (int64_t)r.length};
    return SXG_OK;
}

}  // extern "C"
