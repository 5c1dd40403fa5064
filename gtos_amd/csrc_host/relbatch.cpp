// Host-side graph -> relation tensors, C++17.  Three phases: (a) per graph, in parallel: BFS
// order, all-pairs shortest label paths, and the graph's distinct paths in first-seen order; (b) serial: merge the
// per-graph lists, in graph order, into the batch-wide type table (this is what fixes the reference's type numbering);
// (c) in parallel over the rows of relation[a][c][b][k]: translate and write contiguous memory.
// See include/gtos_host.h for the contract and the reference lines this replaces.  Pure integer work.
#include "../../include/gtos_host.h"
#include "../csrc/relbatch_kernels.h"     // bfs_core / key_core: the search and the path choice shared with the GPU builder

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

namespace {

struct alignas(128) Graph {                          // (neighbours in the per-batch vector belong to different worker threads:
    int n = 0, root = 0;
    std::vector<int> adj_off, adj_dst, adj_lab;     // CSR in networkx adjacency (insertion) order
    std::vector<int> order, pos, depth;             // BFS order from the root, position of each node, BFS depth
};

struct alignas(128) PairPaths {                      //  no two of them on one cache line)  per graph: for every (i,j) in BFS-position space, its packed paths
    std::vector<uint32_t> off;                       // [n*n+1]
    std::vector<uint64_t> keys;
    std::vector<uint32_t> lid;                       // per key: index into uniq
    uint64_t* kview = nullptr;                       // one-path-per-pair modes: this graph's n*n keys / local ids inside the batch-wide
    uint32_t* lview = nullptr;                       // arrays (allocated once, first touched by the workers), no offset table
    std::vector<uint64_t> uniq;                      // this graph's distinct keys in first-seen (i, j, alternative) order
    std::vector<int32_t> gid;                        // per uniq entry: batch-wide type id (phase b)
};

// open-addressing map uint64 -> int (linear probing, 2x capacity); value 0 marks an empty slot
struct FlatMap {
    std::vector<uint64_t> key;
    std::vector<int32_t> val;
    size_t mask = 0;
    void init(size_t n) {                          // (re)usable: a map that is already large enough is only cleared -- no fresh pages
        size_t cap = 16;
        while (cap < 2 * n + 2) cap <<= 1;
        if (cap <= val.size()) { cap = val.size(); std::fill(val.begin(), val.end(), 0); }
        else { key.assign(cap, 0); val.assign(cap, 0); }
        mask = cap - 1;
    }
    static inline size_t hash(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return (size_t)k; }
    inline int find(uint64_t k) const {
        for (size_t p = hash(k) & mask;; p = (p + 1) & mask) { if (!val[p]) return -1; if (key[p] == k) return val[p] - 1; }
    }
    inline void insert(uint64_t k, int id) {                          // k must be absent
        size_t p = hash(k) & mask;
        while (val[p]) p = (p + 1) & mask;
        key[p] = k; val[p] = id + 1;
    }
};

inline uint64_t splitmix(uint64_t& s) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// labels first..last packed one per byte, first label in the low byte (labels are >= 1, so the length is implicit)
inline uint64_t pack(const int* labs, int k) {
    uint64_t key = 0;
    for (int i = 0; i < k; ++i) key |= (uint64_t)(labs[i] & 0xff) << (8 * i);
    return key;
}

bool build_graph(Graph& g, int n, int root, int64_t e0, int64_t e1, const int* src, const int* dst, const int* lab) {
    g.n = n; g.root = root;
    if (n <= 0 || root < 0 || root >= n) return false;
    // ordered adjacency with overwrite-in-place for repeated (u,v) (networkx DiGraph.add_edge)
    std::vector<std::vector<std::pair<int, int>>> adj(n);
    for (int64_t e = e0; e < e1; ++e) {
        const int u = src[e], v = dst[e], l = lab[e];
        if (u < 0 || u >= n || v < 0 || v >= n || l < 1 || l > 255) return false;
        bool found = false;
        for (auto& p : adj[u]) if (p.first == v) { p.second = l; found = true; break; }
        if (!found) adj[u].emplace_back(v, l);
    }
    g.adj_off.assign(n + 1, 0);
    for (int u = 0; u < n; ++u) g.adj_off[u + 1] = g.adj_off[u] + (int)adj[u].size();
    g.adj_dst.resize(g.adj_off[n]); g.adj_lab.resize(g.adj_off[n]);
    for (int u = 0; u < n; ++u)
        for (size_t k = 0; k < adj[u].size(); ++k) { g.adj_dst[g.adj_off[u] + k] = adj[u][k].first; g.adj_lab[g.adj_off[u] + k] = adj[u][k].second; }
    // BFS order / depth from the root (AMRGraph.py:82-98, dependencyGraph.py:36-52)
    g.order.clear(); g.depth.clear(); g.pos.assign(n, -1);
    g.order.push_back(root); g.depth.push_back(0); g.pos[root] = 0;
    for (size_t step = 0; step < g.order.size(); ++step) {
        const int u = g.order[step];
        for (int k = g.adj_off[u]; k < g.adj_off[u + 1]; ++k) {
            const int v = g.adj_dst[k];
            if (g.pos[v] < 0) { g.pos[v] = (int)g.order.size(); g.order.push_back(v); g.depth.push_back(g.depth[step] + 1); }
        }
    }
    return (int)g.order.size() == n;                 // "not connected" is an error in the reference too
}

// All-pairs label paths of one graph.
// Per-thread scratch of the one-path-per-pair modes: the flat search state of gtos_relbatch_dev::Slot, reused for every source.
struct SlotBuf {
    std::vector<int16_t> i16;
    std::vector<double> count;
    std::vector<uint8_t> dlab;
    gtos_relbatch_dev::Slot slot(int n, int e) {
        i16.resize((size_t)4 * n + 2 * (size_t)e);
        count.resize(n);
        dlab.resize(e);
        gtos_relbatch_dev::Slot sl;
        sl.level = i16.data(); sl.head = sl.level + n; sl.tail = sl.head + n; sl.queue = sl.tail + n; sl.dpred = sl.queue + n; sl.dnext = sl.dpred + e;
        sl.count = count.data(); sl.dlab = dlab.data();
        return sl;
    }
};

// One path per pair (GTOS_PATH_FIRST / GTOS_PATH_UNIFORM): no per-pair lists -- a flat search state per source (a few KB, reused) and one
// key per pair written in place.  The same bfs_core / key_core as the GPU builder runs (csrc/relbatch_kernels.h).
void graph_paths_single(const Graph& g, int mode, uint64_t seed, int gid, int max_len, uint64_t self_key, uint64_t tl_key, PairPaths& out, SlotBuf& buf) {
    using namespace gtos_relbatch_dev;
    const int n = g.n;
    const int e = std::max(1, g.adj_off[n]);
    const Slot sl = buf.slot(n, e);
    const int m = mode == GTOS_PATH_UNIFORM ? MODE_UNIFORM : MODE_FIRST;
    for (int i = 0; i < n; ++i) {
        bfs_core(n, g.adj_off.data(), g.adj_dst.data(), g.adj_lab.data(), g.order[i], sl);
        uint64_t* row = out.kview + (size_t)i * n;
        int d;
        for (int j = 0; j < n; ++j) row[j] = key_core(gid, i, j, g.order[j], sl, m, max_len, seed, self_key, tl_key, &d);
    }
}

void graph_paths(const Graph& g, int mode, uint64_t seed, int gid, int max_len, uint64_t self_key, uint64_t tl_key, PairPaths& out) {
    const int n = g.n;
    out.off.assign((size_t)n * n + 1, 0);
    out.keys.clear();
    out.keys.reserve((size_t)n * n + 64);
    std::vector<int> level(n), parent(n), plabel(n), frontier, next;
    std::vector<double> count(n);
    std::vector<std::vector<std::pair<int, int>>> pred(n);            // (predecessor, label of pred->node), discovery order
    std::vector<std::vector<uint64_t>> row((size_t)n);                 // paths per target position for the current source
    int labs[64];
    for (int i = 0; i < n; ++i) {
        const int s = g.order[i];
        std::fill(level.begin(), level.end(), -1);
        for (auto& p : pred) p.clear();
        level[s] = 0; parent[s] = -1; count[s] = 1.0;
        frontier.assign(1, s);
        int lev = 0;
        while (!frontier.empty()) {                                   // nx.predecessor / _single_shortest_path level loop
            ++lev; next.clear();
            for (int v : frontier)
                for (int k = g.adj_off[v]; k < g.adj_off[v + 1]; ++k) {
                    const int w = g.adj_dst[k];
                    if (level[w] < 0) { level[w] = lev; parent[w] = v; plabel[w] = g.adj_lab[k]; count[w] = count[v];
                                        pred[w].emplace_back(v, g.adj_lab[k]); next.push_back(w); }
                    else if (level[w] == lev) { pred[w].emplace_back(v, g.adj_lab[k]); count[w] += count[v]; }
                }
            frontier.swap(next);
        }
        for (auto& r : row) r.clear();
        for (int j = 0; j < n; ++j) {
            const int t = g.order[j];
            const int d = level[t];
            std::vector<uint64_t>& dstv = row[j];
            if (d == 0) { dstv.push_back(self_key); continue; }
            if (d > max_len) { dstv.push_back(tl_key); continue; }   // every alternative collapses to <TL> (data.py:201-202)
            if (mode == GTOS_PATH_FIRST) {
                int v = t;
                for (int k = d - 1; k >= 0; --k) { labs[k] = plabel[v]; v = parent[v]; }
                dstv.push_back(pack(labs, d));
            } else if (mode == GTOS_PATH_UNIFORM) {
                uint64_t st = seed ^ (0x100000001B3ull * (uint64_t)(gid + 1)) ^ ((uint64_t)i << 40) ^ ((uint64_t)j << 20);
                int v = t;
                for (int k = d - 1; k >= 0; --k) {
                    const double u01 = (double)(splitmix(st) >> 11) * (1.0 / 9007199254740992.0);
                    double acc = 0.0, target = u01 * count[v];
                    const auto& pv = pred[v];
                    size_t pick = pv.size() - 1;
                    for (size_t q = 0; q < pv.size(); ++q) { acc += count[pv[q].first]; if (target < acc) { pick = q; break; } }
                    labs[k] = pv[pick].second; v = pv[pick].first;
                }
                dstv.push_back(pack(labs, d));
            } else {                                                  // nx.all_shortest_paths enumeration order
                // explicit stack of (node, next predecessor index), emitting when the source is reached
                int st_node[64], st_idx[64], st_lab[64];
                int top = 0;
                st_node[0] = t; st_idx[0] = 0;
                while (top >= 0) {
                    const int node = st_node[top];
                    if (node == s) {                                  // labels along the stack from the source side
                        for (int k = 0; k < top; ++k) labs[k] = st_lab[top - k];
                        dstv.push_back(pack(labs, top));
                    }
                    if ((int)pred[node].size() > st_idx[top]) {
                        const auto pr = pred[node][st_idx[top]];
                        st_idx[top]++;
                        ++top;
                        st_node[top] = pr.first; st_idx[top] = 0; st_lab[top] = pr.second;
                    } else {
                        --top;
                    }
                }
            }
        }
        for (int j = 0; j < n; ++j) {
            out.off[(size_t)i * n + j + 1] = (uint32_t)(out.keys.size() + row[j].size());
            out.keys.insert(out.keys.end(), row[j].begin(), row[j].end());
        }
    }
    // prefix form: off[k+1] currently holds the end of pair k (monotone by construction)
}

}  // namespace

struct gtos_relbatch {
    int B = 0, n = 0, R = 0, L = 0, K = 1;
    std::vector<int64_t> relation, length;
    std::vector<uint64_t> type_key;               // packed label path of every type: the bank is written from it at export time
    std::vector<int32_t> order, depth;
};

extern "C" gtos_relbatch* gtos_relbatch_build(int B, const int* n_nodes, const int* roots, const int64_t* edge_off,
                                              const int* e_src, const int* e_dst, const int* e_label,
                                              int path_mode, uint64_t seed, const int* ids, int max_len, int n_threads) {
    if (B <= 0 || !n_nodes || !roots || !edge_off || !ids || max_len < 1 || max_len > 8) return nullptr;
    if (path_mode < 0 || path_mode > 2) return nullptr;
    const int pad_id = ids[0], cls_id = ids[1], rcls_id = ids[2], self_id = ids[3], tl_id = ids[4];
    const uint64_t self_key = (uint64_t)self_id, tl_key = (uint64_t)tl_id;
    static const bool timing = getenv("GTOS_RELBATCH_TIMING") != nullptr;      // phase times on stderr
    auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "relbatch %-18s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_start).count());
        t_start = now;
    };
    std::vector<Graph> graphs(B);
    std::vector<PairPaths> pp(B);
    std::atomic<int> next(0), bad(0);
    auto work = [&]() {
        SlotBuf slotbuf;
        FlatMap local;                               // this thread's per-graph dedup map, reused from graph to graph
        std::vector<uint64_t> uniq;
        int done = 0;
        double t_graph = 0, t_paths = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const int g = next.fetch_add(1);
            if (g >= B) {
                if (timing) fprintf(stderr, "relbatch   worker: %d graphs in %.2f ms (graph build %.2f, paths %.2f)\n", done,
                                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), t_graph, t_paths);
                return;
            }
            ++done;
            auto ta = std::chrono::steady_clock::now();
            if (!build_graph(graphs[g], n_nodes[g], roots[g], edge_off[g], edge_off[g + 1], e_src, e_dst, e_label)) { bad = 1; continue; }
            // the flat search of the one-path-per-pair modes keeps node ids and shortest-path-DAG edge ids in int16 scratch (shared with the
            // GPU builder): refuse what does not fit instead of wrapping (include/gtos_host.h states the limit)
            if (path_mode != GTOS_PATH_ALL && (graphs[g].n > 32767 || graphs[g].adj_off[graphs[g].n] > 32767)) { bad = 1; continue; }
            auto tb = std::chrono::steady_clock::now();
            if (path_mode == GTOS_PATH_ALL) graph_paths(graphs[g], path_mode, seed, g, max_len, self_key, tl_key, pp[g]);
            else graph_paths_single(graphs[g], path_mode, seed, g, max_len, self_key, tl_key, pp[g], slotbuf);
            auto tc = std::chrono::steady_clock::now();
            t_graph += std::chrono::duration<double, std::milli>(tb - ta).count();
            t_paths += std::chrono::duration<double, std::milli>(tc - tb).count();
            // phase (a), second half: this graph's distinct keys, in the order the reference would first meet them
            PairPaths& P = pp[g];
            const bool single = path_mode != GTOS_PATH_ALL;
            const size_t nk = single ? (size_t)graphs[g].n * graphs[g].n : P.keys.size();
            const uint64_t* keys = single ? P.kview : P.keys.data();
            if (!single) P.lid.resize(nk);
            uint32_t* lid = single ? P.lview : P.lid.data();
            local.init(nk);
            uniq.clear();                            // (a thread-local list: push_back on a member of pp[g] would write the vector's
            for (size_t q = 0; q < nk; ++q) {        //  end pointer, next to another thread's pp[g + 1], for every new key)
                int id = local.find(keys[q]);
                if (id < 0) { id = (int)uniq.size(); local.insert(keys[q], id); uniq.push_back(keys[q]); }
                lid[q] = (uint32_t)id;
            }
            P.uniq.assign(uniq.begin(), uniq.end());
        }
    };
    if (n_threads < 1) n_threads = (int)std::thread::hardware_concurrency();
    n_threads = std::max(1, std::min(n_threads, B));
    auto run_parallel = [&](auto&& fn) {
        std::vector<std::thread> pool;
        for (int t = 1; t < n_threads; ++t) pool.emplace_back(fn);
        fn();
        for (auto& t : pool) t.join();
    };
    std::unique_ptr<uint64_t[]> all_keys;
    std::unique_ptr<uint32_t[]> all_lid;
    if (path_mode != GTOS_PATH_ALL) {
        size_t P = 0;
        for (int g = 0; g < B; ++g) { if (n_nodes[g] <= 0) return nullptr; P += (size_t)n_nodes[g] * n_nodes[g]; }
        all_keys.reset(new uint64_t[P]);             // uninitialised on purpose: the workers touch their own slices first
        all_lid.reset(new uint32_t[P]);
        size_t at = 0;
        for (int g = 0; g < B; ++g) { pp[g].kview = all_keys.get() + at; pp[g].lview = all_lid.get() + at; at += (size_t)n_nodes[g] * n_nodes[g]; }
    }
    run_parallel(work);
    if (bad) return nullptr;
    lap("paths (parallel)");

    auto* h = new gtos_relbatch();
    h->B = B;
    int nmax = 0;
    for (int g = 0; g < B; ++g) nmax = std::max(nmax, graphs[g].n);
    const int n = nmax + 1;
    h->n = n;
    const bool all = path_mode == GTOS_PATH_ALL;
    int K = 1;
    if (all)
        for (int g = 0; g < B; ++g)
            for (size_t k = 0; k + 1 < pp[g].off.size(); ++k) {
                const uint32_t b0 = k ? pp[g].off[k] : 0u;
                K = std::max(K, (int)(pp[g].off[k + 1] - b0));
            }
    h->K = K;
    // phase (b): type ids in the reference's first-seen order -- graphs in order, source i, target j, alternative
    // (data.py:140-162 / :186-213); the per-graph lists are already in that order, so merging them in graph order is enough
    size_t tot_uniq = 8;
    for (int g = 0; g < B; ++g) tot_uniq += pp[g].uniq.size();
    FlatMap types;
    types.init(tot_uniq);
    std::vector<uint64_t> type_key;
    type_key.reserve(tot_uniq);
    auto intern = [&](uint64_t key) { int id = types.find(key); if (id >= 0) return id;
                                      id = (int)type_key.size(); types.insert(key, id); type_key.push_back(key); return id; };
    int t_cls, t_rcls, t_self;
    if (all) { intern((uint64_t)pad_id); t_cls = intern((uint64_t)cls_id); t_rcls = intern((uint64_t)rcls_id); t_self = intern(self_key); }
    else { t_cls = intern((uint64_t)cls_id); t_rcls = intern((uint64_t)rcls_id); t_self = intern(self_key); }
    for (int g = 0; g < B; ++g) {
        PairPaths& P = pp[g];
        P.gid.resize(P.uniq.size());
        for (size_t u = 0; u < P.uniq.size(); ++u) P.gid[u] = intern(P.uniq[u]);
    }
    lap("type table (serial)");
    h->relation.assign((size_t)n * n * B * K, 0);
    h->order.assign((size_t)B * (n - 1), -1);
    h->depth.assign((size_t)B * (n - 1), 0);
    auto rel_at = [&](int a, int c, int b, int k) -> int64_t& { return h->relation[(((size_t)a * n + c) * B + b) * K + k]; };
    for (int b = 0; b < B; ++b) {
        const Graph& g = graphs[b];
        const int ng = g.n;
        for (int p = 0; p < ng; ++p) { h->order[(size_t)b * (n - 1) + p] = g.order[p]; h->depth[(size_t)b * (n - 1) + p] = g.depth[p]; }
        rel_at(0, 0, b, 0) = t_self;                                   // brs[0] = [<SELF>, <CLS>...], brs[c][0] = <rCLS>
        for (int a = 1; a <= ng; ++a) { rel_at(a, 0, b, 0) = t_cls; rel_at(0, a, b, 0) = t_rcls; }
    }
    // phase (c): relation[a = j+1][c = i+1][b][k] = type of the k-th path from BFS position i to j of graph b.  A thread owns
    // whole rows a, i.e. contiguous memory (no false sharing); eval keeps only the first alternative when it is
    // <SELF>/<TL>, and those pairs hold exactly one key already.
    std::atomic<int> next_row(1);
    auto fill = [&]() {
        for (;;) {
            const int a = next_row.fetch_add(1);
            if (a >= n) return;
            const int j = a - 1;
            for (int c = 1; c < n; ++c) {
                const int i = c - 1;
                int64_t* dst = &h->relation[(((size_t)a * n + c) * B) * K];
                for (int b = 0; b < B; ++b) {
                    const int ng = graphs[b].n;
                    if (i >= ng || j >= ng) continue;
                    const PairPaths& P = pp[b];
                    const size_t k = (size_t)i * ng + j;
                    if (!all) { dst[b] = P.gid[P.lview[k]]; continue; }                 // one key per pair (K == 1)
                    const uint32_t b0 = k ? P.off[k] : 0u, b1 = P.off[k + 1];
                    for (uint32_t q = b0; q < b1; ++q) dst[(size_t)b * K + (q - b0)] = P.gid[P.lid[q]];
                }
            }
        }
    };
    run_parallel(fill);
    lap("relation fill");
    const int R = (int)type_key.size();
    h->R = R;
    int L = 1;
    std::vector<int> len(R);
    for (int t = 0; t < R; ++t) { int l = 0; uint64_t k = type_key[t]; while (k) { ++l; k >>= 8; } len[t] = std::max(l, 1); L = std::max(L, len[t]); }
    h->L = L;
    h->length.resize(R);
    for (int t = 0; t < R; ++t) h->length[t] = len[t];
    h->type_key.swap(type_key);                       // (the bank itself is written straight into the caller's buffer by _export)
    lap("bank");
    return h;
}

extern "C" int gtos_relbatch_dims(const gtos_relbatch* h, int* n, int* R, int* L, int* K) {
    if (!h) return -1;
    if (n) *n = h->n;
    if (R) *R = h->R;
    if (L) *L = h->L;
    if (K) *K = h->K;
    return 0;
}

extern "C" int gtos_relbatch_export(const gtos_relbatch* h, int64_t* relation, int64_t* bank, int64_t* length,
                                    int32_t* order, int32_t* depth) {
    if (!h) return -1;
    if (relation) std::memcpy(relation, h->relation.data(), h->relation.size() * sizeof(int64_t));
    if (bank)
        for (int i = 0; i < h->L; ++i) {                                    // row by row: sequential writes (0 past a path's end)
            int64_t* row = bank + (size_t)i * h->R;
            const uint64_t* tk = h->type_key.data();
            for (int t = 0; t < h->R; ++t) row[t] = (int64_t)((tk[t] >> (8 * i)) & 0xff);
        }
    if (length) std::memcpy(length, h->length.data(), h->length.size() * sizeof(int64_t));
    if (order) std::memcpy(order, h->order.data(), h->order.size() * sizeof(int32_t));
    if (depth) std::memcpy(depth, h->depth.data(), h->depth.size() * sizeof(int32_t));
    return 0;
}

extern "C" void gtos_relbatch_free(gtos_relbatch* h) { delete h; }

// The batch's graphs as the GPU relation-batch builder reads them (gtos_amd/csrc/relbatch_kernels.h struct Graphs): the same ordered
// adjacency and BFS order as gtos_relbatch_build uses, flattened.  The all-pairs work stays with the caller.
extern "C" int64_t gtos_relbatch_csr(int B, const int* n_nodes, const int* roots, const int64_t* edge_off, const int* e_src, const int* e_dst,
                                     const int* e_label, int32_t* node_off, int32_t* adj_base, int32_t* adj_off, int32_t* adj_dst,
                                     int32_t* adj_lab, int32_t* order, int32_t* depth) {
    if (B <= 0 || !n_nodes || !roots || !edge_off || !node_off || !adj_base || !adj_off || !adj_dst || !adj_lab || !order || !depth) return -1;
    int64_t nodes = 0, adj = 0;
    for (int b = 0; b < B; ++b)                      // sizes first: the caller sized the outputs by the sums of these
        if (n_nodes[b] <= 0 || edge_off[b] < 0 || edge_off[b + 1] < edge_off[b]) return -1;
    Graph g;
    for (int b = 0; b < B; ++b) {
        if (!build_graph(g, n_nodes[b], roots[b], edge_off[b], edge_off[b + 1], e_src, e_dst, e_label)) return -1;
        if (nodes + g.n > 0x7fffffffLL || g.n > 32767 || g.adj_off[g.n] > 32767) return -1;   // the device scratch holds int16 ids
        node_off[b] = (int32_t)nodes;
        adj_base[b] = (int32_t)adj;
        std::memcpy(adj_off + nodes + b, g.adj_off.data(), (size_t)(g.n + 1) * sizeof(int32_t));
        if (!g.adj_dst.empty()) {                     // (a single node has no adjacency: no copy from a null vector)
            std::memcpy(adj_dst + adj, g.adj_dst.data(), g.adj_dst.size() * sizeof(int32_t));
            std::memcpy(adj_lab + adj, g.adj_lab.data(), g.adj_lab.size() * sizeof(int32_t));
        }
        std::memcpy(order + nodes, g.order.data(), (size_t)g.n * sizeof(int32_t));
        std::memcpy(depth + nodes, g.depth.data(), (size_t)g.n * sizeof(int32_t));
        nodes += g.n;
        adj += (int64_t)g.adj_dst.size();
    }
    node_off[B] = (int32_t)nodes;
    adj_base[B] = (int32_t)adj;
    return adj;
}
