// Per-thread bodies of the GPU relation-batch builder (csrc/relbatch_dev.hip), written so that the SAME code compiles for the host:
// oracle/relbatch_emul.cpp runs every stage as a serial loop (std::stable_sort / a running sum where the GPU uses rocPRIM) and
// tests/test_relbatch_dev.py compares the result with csrc_host/relbatch.cpp (itself bit-exact with the reference's batchify,
// tests/test_host_relbatch.py) array for array.  What the emulation cannot cover is the launch glue and the rocPRIM calls.
//
// The work is the relation section of the reference's batchify (generator/data.py:134-176, translator/data.py:132-176) and the
// all-pairs shortest label paths under it (generator/AMRGraph.py:100-115, translator/dependencyGraph.py:54-74):
//   one thread per (graph, source): BFS over the graph's adjacency, the shortest-path DAG (predecessors in discovery order) and the
//     number of shortest paths to every node;
//   one thread per (graph, source, target): one shortest label path packed into a 64-bit key (first discovery, or uniform among
//     the alternatives by the same splitmix64 stream as the host builder), <SELF> / <TL> for distance 0 / beyond max_len;
//   key sort -> distinct keys -> numbered in FIRST-SEEN order (graph, source, target: the order the reference's dict meets them,
//     which fixes the type ids) by a second sort of the first positions -> relation[n,n,B], relation_bank[L,R], relation_length[R].
// Covered: GTOS_PATH_FIRST and GTOS_PATH_UNIFORM (one path per pair: the train-mode batches) and GTOS_PATH_ALL (every shortest path of a
// pair, the eval-mode batches: own stages further down).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define GTOS_RB_HD __host__ __device__ inline
#else
#define GTOS_RB_HD inline
#endif

namespace gtos_relbatch_dev {

enum { MODE_FIRST = 0, MODE_UNIFORM = 1, MODE_ALL = 2 };   // (ALL: every shortest path of a pair -- the eval-mode batches, own entry points)
// sizes[] written on the device, read by the host once per batch
enum { RZ_R = 0, RZ_L = 1, RZ_N = 2, RZ_T = 3, RZ_K = 4, RZ_TOTAL = 8 };     // distinct paths, longest path, sum of their lengths; ALL: paths in total, most per pair
enum { N_SPECIAL = 3, N_SPECIAL_ALL = 4 };               // <CLS>, <rCLS>, <SELF> are interned before any pair (relbatch.cpp phase b); ALL: <PAD> first

struct Geom {
    int32_t B, n, nmax, emax, max_len, mode;              // n = 1 + most nodes of a graph (index 0 = <CLS>); nmax / emax: scratch strides
    int32_t S;                                            // (graph, source) slots = sum of the node counts
    int64_t P;                                            // pairs = sum of the squared node counts
    uint64_t seed, cls_key, rcls_key, self_key, tl_key;
    int64_t T;                                            // MODE_ALL: paths of all pairs together (known after the counting phase)
    int32_t K;                                            // MODE_ALL: most paths of one pair
    uint64_t pad_key;
};

struct Graphs {                                           // device (or host) arrays describing the batch's graphs
    const int32_t* ng;                                    // [B] nodes of graph g
    const int32_t* node_off;                              // [B + 1] prefix sums of ng: slot of (g, i) = node_off[g] + i
    const int64_t* pair_off;                              // [B + 1] prefix sums of ng^2
    const int32_t* adj_base;                              // [B + 1] first adjacency entry of graph g
    const int32_t* adj_off;                               // [S + B] per graph ng + 1 LOCAL offsets, graph g's block at node_off[g] + g
    const int32_t* adj_dst;                               // adjacency in networkx (insertion) order
    const int32_t* adj_lab;
    const int32_t* order;                                 // [S] node at BFS position i of graph g
};

struct Scratch {                                          // per (graph, source) slot, strides nmax / emax
    int16_t* level;                                       // [S, nmax] BFS level (-1: not reached)
    double* count;                                        // [S, nmax] number of shortest paths from the source
    int16_t* head;                                        // [S, nmax] first DAG edge into the node (-1: none)
    int16_t* tail;                                        // [S, nmax] last one
    int16_t* queue;                                       // [S, nmax] BFS queue
    int16_t* dpred;                                       // [S, emax] DAG edge: predecessor node
    int16_t* dnext;                                       // [S, emax] next DAG edge into the same node (-1: last)
    uint8_t* dlab;                                        // [S, emax] label of predecessor -> node
};

GTOS_RB_HD uint64_t splitmix(uint64_t& s) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// largest g with off[g] <= x (off ascending, off[0] = 0, off[B] > x)
template <typename T>
GTOS_RB_HD int32_t graph_of(const T* off, int32_t B, T x) {
    int32_t lo = 0, hi = B;
    while (hi - lo > 1) {
        const int32_t mid = (lo + hi) >> 1;
        if (off[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

struct Slot {                                             // the scratch of ONE (graph, source) search
    int16_t *level, *head, *tail, *queue, *dpred, *dnext;
    double* count;
    uint8_t* dlab;
};

// ---- stage 1 core: BFS from node ``src`` of a graph of n nodes (LOCAL adjacency offsets off[n + 1], dst / lab): levels, path counts,
// shortest-path DAG in discovery order.  The level loop of nx.predecessor (generator/AMRGraph.py:100-115 via networkx): a queue visits
// the nodes in the order of the frontier lists, so every DAG list and every floating-point sum is built in one fixed order.
// Shared by the GPU kernels, their emulation and the host builder (csrc_host/relbatch.cpp, one-path-per-pair modes).
GTOS_RB_HD void bfs_core(int32_t n, const int32_t* off, const int32_t* dst, const int32_t* lab, int32_t src, const Slot& sl) {
    for (int32_t v = 0; v < n; ++v) { sl.level[v] = -1; sl.head[v] = -1; sl.tail[v] = -1; sl.count[v] = 0.0; }
    sl.level[src] = 0; sl.count[src] = 1.0;
    int32_t qh = 0, qt = 0, ne = 0;
    sl.queue[qt++] = (int16_t)src;
    while (qh < qt) {
        const int32_t v = sl.queue[qh++];
        const int32_t lev = sl.level[v] + 1;
        for (int32_t k = off[v]; k < off[v + 1]; ++k) {
            const int32_t w = dst[k];
            bool edge = false;
            if (sl.level[w] < 0) { sl.level[w] = (int16_t)lev; sl.count[w] = sl.count[v]; sl.queue[qt++] = (int16_t)w; edge = true; }
            else if (sl.level[w] == lev) { sl.count[w] += sl.count[v]; edge = true; }
            if (edge) {
                sl.dpred[ne] = (int16_t)v; sl.dlab[ne] = (uint8_t)lab[k]; sl.dnext[ne] = -1;
                if (sl.tail[w] < 0) sl.head[w] = (int16_t)ne; else sl.dnext[sl.tail[w]] = (int16_t)ne;
                sl.tail[w] = (int16_t)ne;
                ++ne;
            }
        }
    }
}

// ---- stage 2 core: the key of the path from the slot's source (BFS position i of graph g) to node t (BFS position j): first
// discovery, or uniform among the alternatives (generator/data.py:149-150) by a splitmix64 stream keyed by (seed, g, i, j);
// <SELF> for distance 0, <TL> beyond max_len (data.py:151-154).  *d_out = the distance.
GTOS_RB_HD uint64_t key_core(int32_t g, int32_t i, int32_t j, int32_t t, const Slot& sl, int32_t mode, int32_t max_len, uint64_t seed,
                             uint64_t self_key, uint64_t tl_key, int32_t* d_out) {
    const int32_t d = sl.level[t];
    *d_out = d;
    if (d == 0) return self_key;
    if (d > max_len) return tl_key;
    uint64_t k64 = 0;
    uint64_t st = seed ^ (0x100000001B3ull * (uint64_t)(g + 1)) ^ ((uint64_t)i << 40) ^ ((uint64_t)j << 20);
    int32_t v = t;
    for (int32_t k = d - 1; k >= 0; --k) {
        int32_t e = sl.head[v];
        if (mode == MODE_UNIFORM) {
            const double u01 = (double)(splitmix(st) >> 11) * (1.0 / 9007199254740992.0);
            const double target = u01 * sl.count[v];
            double acc = 0.0;
            for (int32_t guard = 0;; ++guard) {                // (guard: a DAG list has at most 32,767 entries -- a corrupted one must not spin)
                acc += sl.count[sl.dpred[e]];
                if (target < acc || sl.dnext[e] < 0 || guard >= 32767) break;
                e = sl.dnext[e];
            }
        }
        k64 |= (uint64_t)sl.dlab[e] << (8 * k);               // label k of the path in byte k (first label in the low byte)
        v = sl.dpred[e];
    }
    return k64;
}

GTOS_RB_HD Slot slot_of(const Scratch& sc, int64_t s, int32_t nmax, int32_t emax) {
    Slot sl;
    sl.level = sc.level + s * nmax; sl.count = sc.count + s * nmax; sl.head = sc.head + s * nmax; sl.tail = sc.tail + s * nmax;
    sl.queue = sc.queue + s * nmax; sl.dpred = sc.dpred + s * emax; sl.dnext = sc.dnext + s * emax; sl.dlab = sc.dlab + s * emax;
    return sl;
}

// ---- stage 1: BFS from the node at position i of graph g (slot s of the batch-wide scratch)
GTOS_RB_HD void bfs_source(int32_t s, const Geom& G, const Graphs& gr, const Scratch& sc) {
    const int32_t g = graph_of<int32_t>(gr.node_off, G.B, s);
    const int32_t i = s - gr.node_off[g];
    bfs_core(gr.ng[g], gr.adj_off + gr.node_off[g] + g, gr.adj_dst + gr.adj_base[g], gr.adj_lab + gr.adj_base[g], gr.order[gr.node_off[g] + i],
             slot_of(sc, s, G.nmax, G.emax));
}

// ---- stage 2: the key of pair p (graph g, source position i, target position j) and its first-seen position
GTOS_RB_HD void pair_key(int64_t p, const Geom& G, const Graphs& gr, const Scratch& sc, uint64_t* key, int32_t* posn, int32_t* len_seen) {
    const int32_t g = graph_of<int64_t>(gr.pair_off, G.B, p);
    const int32_t n = gr.ng[g];
    const int64_t q = p - gr.pair_off[g];
    const int32_t i = (int32_t)(q / n), j = (int32_t)(q % n);
    int32_t d;
    key[N_SPECIAL + p] = key_core(g, i, j, gr.order[gr.node_off[g] + j], slot_of(sc, gr.node_off[g] + i, G.nmax, G.emax), G.mode, G.max_len, G.seed,
                                  G.self_key, G.tl_key, &d);
    posn[N_SPECIAL + p] = (int32_t)(N_SPECIAL + p);
    len_seen[d >= 1 && d <= G.max_len ? d - 1 : 0] = 1;       // benign race: every writer stores 1 (<SELF> / <TL>: one label)
}

// one thread: the three types interned before any pair
GTOS_RB_HD void special_keys(const Geom& G, uint64_t* key, int32_t* posn, int32_t* len_seen) {
    key[0] = G.cls_key; key[1] = G.rcls_key; key[2] = G.self_key;
    posn[0] = 0; posn[1] = 1; posn[2] = 2;
    len_seen[0] = 1;
}

GTOS_RB_HD int32_t key_labels(uint64_t k) {                // labels of a packed path (>= 1: key 0 is a one-label path of id 0)
    int32_t l = 0;
    while (k) { ++l; k >>= 8; }
    return l < 1 ? 1 : l;
}

// ---- stage 3 (after the stable key sort): element e opens a distinct key.  The scan input carries two counters in one 64-bit word:
// low half 1 per distinct key, high half the key's length -- the inclusive plus-scan then gives the segment of every element (low)
// and, at the end, the row count of the bank (high: sum of the path lengths, < 2^32).
GTOS_RB_HD void head_flag(int64_t e, const uint64_t* key, uint64_t* flag) {
    flag[e] = (e == 0 || key[e] != key[e - 1]) ? (((uint64_t)key_labels(key[e]) << 32) | 1ull) : 0ull;
}

// after the inclusive scan of the flags: segment of e = low(cum[e]) - 1; a head records its segment's key and first position
GTOS_RB_HD void segment_first(int64_t e, const uint64_t* key, const int32_t* posn, const uint64_t* cum, uint32_t* first_pos, int32_t* seg_id,
                              uint64_t* seg_key) {
    const uint32_t c = (uint32_t)cum[e];
    if (e == 0 || c != (uint32_t)cum[e - 1]) { first_pos[c - 1] = (uint32_t)posn[e]; seg_id[c - 1] = (int32_t)(c - 1); seg_key[c - 1] = key[e]; }
}

// one thread: R = distinct keys, N = sum of their lengths, L = the longest path among them (len_seen[l - 1]: a key of l labels exists)
GTOS_RB_HD void sizes_after_scan(const uint64_t* cum, int64_t total, const int32_t* len_seen, int32_t* sizes) {
    sizes[RZ_R] = (int32_t)(uint32_t)cum[total - 1];
    sizes[RZ_N] = (int32_t)(cum[total - 1] >> 32);
    int32_t L = 1;
    for (int32_t l = 0; l < 8; ++l) if (len_seen[l]) L = l + 1;
    sizes[RZ_L] = L;
}

// ---- stage 4 (after the sort of the segments by first position): rank r = the type id; its bank column and length.
// bank: int64 [8, R] ZERO-FILLED by the caller (rows past L stay zero and are cut off by the caller).
GTOS_RB_HD void type_of_segment(int64_t r, const int32_t* sorted_seg, const uint64_t* seg_key, int32_t* type_of_seg, int64_t R, int64_t* bank,
                                int64_t* length) {
    const int32_t sg = sorted_seg[r];
    type_of_seg[sg] = (int32_t)r;
    uint64_t k = seg_key[sg];
    int32_t l = 0;
    while (k) { bank[(int64_t)l * R + r] = (int64_t)(k & 0xff); ++l; k >>= 8; }
    length[r] = l < 1 ? 1 : l;                              // key 0 (a special with id 0): one label, id 0
}

// ---- stage 5: relation[a = j + 1][c = i + 1][g] = type of the pair's key (element e of the sorted list)
GTOS_RB_HD void scatter_relation(int64_t e, const Geom& G, const Graphs& gr, const int32_t* posn, const uint64_t* cum, const int32_t* type_of_seg,
                                 int64_t* relation) {
    const int32_t pos = posn[e];
    if (pos < N_SPECIAL) return;
    const int64_t p = pos - N_SPECIAL;
    const int32_t g = graph_of<int64_t>(gr.pair_off, G.B, p);
    const int32_t n = gr.ng[g];
    const int64_t q = p - gr.pair_off[g];
    const int32_t i = (int32_t)(q / n), j = (int32_t)(q % n);
    relation[((int64_t)(j + 1) * G.n + (i + 1)) * G.B + g] = type_of_seg[(uint32_t)cum[e] - 1];
}

// the <CLS> row / column of graph g at node position a - 1: brs[0] = [<SELF>, <CLS> ...], brs[c][0] = <rCLS> (relbatch.cpp phase b).
// <CLS>, <rCLS>, <SELF> sit at first-seen positions 0, 1, 2 and their ids differ (checked by the caller), so their types are 0, 1, 2.
GTOS_RB_HD void cls_cells(int32_t s, const Geom& G, const Graphs& gr, int64_t* relation) {
    const int32_t g = graph_of<int32_t>(gr.node_off, G.B, s);
    const int32_t a = s - gr.node_off[g] + 1;
    relation[((int64_t)a * G.n + 0) * G.B + g] = 0;
    relation[((int64_t)0 * G.n + a) * G.B + g] = 1;
    if (a == 1) relation[g] = 2;
}

// =====================================================================================================================================
// MODE_ALL: every shortest path of every pair, in networkx's enumeration order (generator/data.py:178-232: the eval-mode batches,
// relation[n,n,B,K] with type 0 = <PAD> behind a pair's last alternative).  Counting phase -> host read (T, K) -> key phase -> host
// read (R, L, N) -> fill phase.

// ---- the number of shortest paths of pair p = the BFS path count of its target (1 for <SELF> / <TL>)
GTOS_RB_HD void pair_alt_count(int64_t p, const Geom& G, const Graphs& gr, const Scratch& sc, uint32_t* nalt, uint64_t* nalt64) {
    const int32_t g = graph_of<int64_t>(gr.pair_off, G.B, p);
    const int32_t n = gr.ng[g];
    const int64_t q = p - gr.pair_off[g];
    const int32_t i = (int32_t)(q / n), j = (int32_t)(q % n);
    const Slot sl = slot_of(sc, gr.node_off[g] + i, G.nmax, G.emax);
    const int32_t t = gr.order[gr.node_off[g] + j];
    const int32_t d = sl.level[t];
    const double c = sl.count[t];
    nalt[p] = (d == 0 || d > G.max_len) ? 1u : (c >= 4294967295.0 ? 0xffffffffu : (uint32_t)c);
    nalt64[p] = nalt[p];                                      // (the sum scan runs in 64 bits)
}

// one thread: totals of the counting phase (cum = inclusive sum scan, cmax = inclusive max scan of nalt)
GTOS_RB_HD void sizes_count(const uint64_t* cum, const uint32_t* cmax, int64_t P, int32_t* sizes) {
    const uint64_t T = cum[P - 1];
    sizes[RZ_T] = T > 0x7fffffffull ? -1 : (int32_t)T;
    sizes[RZ_K] = (int32_t)(cmax[P - 1] > 0x7fffffffu ? 0x7fffffffu : cmax[P - 1]);
}

// ---- the keys of pair p, all of them, in the order of nx.all_shortest_paths: a depth-first walk over the predecessor lists from the
// target back to the source (an explicit stack of (node, next list entry)), emitting a path whenever the source is reached --
// csrc_host/relbatch.cpp graph_paths, GTOS_PATH_ALL.  cum = inclusive scan of the counts: the pair's keys start at cum[p] - nalt[p].
GTOS_RB_HD void pair_alt_keys(int64_t p, const Geom& G, const Graphs& gr, const Scratch& sc, const uint64_t* cum, const uint32_t* nalt,
                              uint64_t* key, int32_t* posn, int32_t* len_seen) {
    const int32_t g = graph_of<int64_t>(gr.pair_off, G.B, p);
    const int32_t n = gr.ng[g];
    const int64_t q = p - gr.pair_off[g];
    const int32_t i = (int32_t)(q / n), j = (int32_t)(q % n);
    const Slot sl = slot_of(sc, gr.node_off[g] + i, G.nmax, G.emax);
    const int32_t s = gr.order[gr.node_off[g] + i], t = gr.order[gr.node_off[g] + j];
    const int32_t d = sl.level[t];
    int64_t at = N_SPECIAL_ALL + (int64_t)(cum[p] - nalt[p]);
    const int64_t end = at + nalt[p];
    if (d == 0 || d > G.max_len) {
        key[at] = d == 0 ? G.self_key : G.tl_key;
        posn[at] = (int32_t)at;
        len_seen[0] = 1;
        return;
    }
    int32_t st_node[10], st_edge[10], st_lab[10];            // depth <= max_len <= 8
    int32_t top = 0;
    st_node[0] = t; st_edge[0] = sl.head[t]; st_lab[0] = 0;
    while (top >= 0) {
        const int32_t node = st_node[top];
        if (node == s && at < end) {                            // labels along the stack from the source side
            uint64_t k64 = 0;
            for (int32_t k = 0; k < top; ++k) k64 |= (uint64_t)(st_lab[top - k] & 0xff) << (8 * k);
            key[at] = k64;
            posn[at] = (int32_t)at;
            ++at;
        }
        const int32_t e = node == s ? -1 : st_edge[top];       // (the source has no predecessor in the DAG)
        if (e >= 0) {
            st_edge[top] = sl.dnext[e];
            ++top;
            st_node[top] = sl.dpred[e]; st_edge[top] = sl.head[sl.dpred[e]]; st_lab[top] = sl.dlab[e];
        } else {
            --top;
        }
    }
    len_seen[d - 1] = 1;
}

GTOS_RB_HD void special_keys_all(const Geom& G, uint64_t* key, int32_t* posn, int32_t* len_seen) {
    key[0] = G.pad_key; key[1] = G.cls_key; key[2] = G.rcls_key; key[3] = G.self_key;
    posn[0] = 0; posn[1] = 1; posn[2] = 2; posn[3] = 3;
    len_seen[0] = 1;
}

// ---- relation[a = j + 1][c = i + 1][g][k] = type of the k-th path of the pair (element e of the sorted list)
GTOS_RB_HD void scatter_relation_all(int64_t e, const Geom& G, const Graphs& gr, const int32_t* posn, const uint64_t* cum_flag, const int32_t* type_of_seg,
                                     const uint64_t* cum_alt, int64_t* relation) {
    const int32_t pos = posn[e];
    if (pos < N_SPECIAL_ALL) return;
    const uint64_t x = (uint64_t)(pos - N_SPECIAL_ALL);        // index among all paths: pair = first p with cum_alt[p] > x
    int64_t lo = 0, hi = G.P - 1;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cum_alt[mid] > x) hi = mid; else lo = mid + 1;
    }
    const int64_t p = lo;
    const int64_t k = (int64_t)x - (int64_t)(p ? cum_alt[p - 1] : 0);
    const int32_t g = graph_of<int64_t>(gr.pair_off, G.B, p);
    const int32_t n = gr.ng[g];
    const int64_t q = p - gr.pair_off[g];
    const int32_t i = (int32_t)(q / n), j = (int32_t)(q % n);
    relation[(((int64_t)(j + 1) * G.n + (i + 1)) * G.B + g) * G.K + k] = type_of_seg[(uint32_t)cum_flag[e] - 1];
}

// the <CLS> row / column: <PAD>, <CLS>, <rCLS>, <SELF> sit at first-seen positions 0..3 (different ids, checked by the caller): types 0..3
GTOS_RB_HD void cls_cells_all(int32_t s, const Geom& G, const Graphs& gr, int64_t* relation) {
    const int32_t g = graph_of<int32_t>(gr.node_off, G.B, s);
    const int32_t a = s - gr.node_off[g] + 1;
    relation[(((int64_t)a * G.n + 0) * G.B + g) * G.K] = 1;
    relation[(((int64_t)0 * G.n + a) * G.B + g) * G.K] = 2;
    if (a == 1) relation[(int64_t)g * G.K] = 3;
}

}  // namespace gtos_relbatch_dev

// ---- the two-phase C ABI (gtos_relbatch_dev_phase_a / _b and their emulation)
// geom[]: host integers, int64 each
namespace gtos_relbatch_dev {
enum { GE_B = 0, GE_N, GE_NMAX, GE_EMAX, GE_MAX_LEN, GE_MODE, GE_S, GE_P, GE_SEED, GE_CLS, GE_RCLS, GE_SELF, GE_TL, GE_T, GE_K, GE_PAD, GE_COUNT };
// tab[]: device (or host) pointers
enum { T_NG = 0,          // int32 [B]
       T_NODE_OFF,        // int32 [B + 1]
       T_PAIR_OFF,        // int64 [B + 1]
       T_ADJ_BASE,        // int32 [B + 1]
       T_ADJ_OFF,         // int32 [S + B]
       T_ADJ_DST,         // int32 [adjacency entries]
       T_ADJ_LAB,         // int32 [adjacency entries]
       T_ORDER,           // int32 [S]
       T_LEVEL,           // int16 [S, nmax]
       T_COUNT_,          // double [S, nmax]
       T_HEAD,            // int16 [S, nmax]
       T_TAIL,            // int16 [S, nmax]
       T_QUEUE,           // int16 [S, nmax]
       T_DPRED,           // int16 [S, emax]
       T_DNEXT,           // int16 [S, emax]
       T_DLAB,            // uint8 [S, emax]
       T_KEY,             // uint64 [P + 3]   keys in first-seen order
       T_POSN,            // int32 [P + 3]
       T_SKEY,            // uint64 [P + 3]   sorted by key (stable)
       T_SPOS,            // int32 [P + 3]
       T_FLAG,            // uint64 [P + 3]   (length << 32 | 1) on the first element of a distinct key
       T_CUM,             // uint64 [P + 3]   its inclusive scan
       T_FIRST_POS,       // uint32 [P + 3]  (R used)
       T_SEG_ID,          // int32 [P + 3]
       T_SEG_KEY,         // uint64 [P + 3]
       T_FIRST_ALT,       // uint32 [R]       phase B
       T_SORTED_SEG,      // int32 [R]
       T_TYPE_OF_SEG,     // int32 [R]
       T_LEN_SEEN,        // int32 [8]
       T_SIZES,           // int32 [RZ_TOTAL]
       T_RELATION,        // int64 [n, n, B]  ZERO-FILLED by the caller
       T_BANK,            // int64 [8, R]     ZERO-FILLED by the caller (phase B)
       T_LENGTH,          // int64 [R]
       T_NALT,            // uint32 [P]   MODE_ALL: shortest paths of every pair
       T_CUM_ALT,         // uint64 [P]   their inclusive sum
       T_CMAX_ALT,        // uint32 [P]   their inclusive maximum
       T_NALT64,          // uint64 [P]   the counts again, as the 64-bit input of the sum scan
       T_TABLE_COUNT };

inline Geom geom_of(const int64_t* g) {
    Geom G;
    G.B = (int32_t)g[GE_B]; G.n = (int32_t)g[GE_N]; G.nmax = (int32_t)g[GE_NMAX]; G.emax = (int32_t)g[GE_EMAX];
    G.max_len = (int32_t)g[GE_MAX_LEN]; G.mode = (int32_t)g[GE_MODE]; G.S = (int32_t)g[GE_S]; G.P = g[GE_P];
    G.seed = (uint64_t)g[GE_SEED]; G.cls_key = (uint64_t)g[GE_CLS]; G.rcls_key = (uint64_t)g[GE_RCLS];
    G.self_key = (uint64_t)g[GE_SELF]; G.tl_key = (uint64_t)g[GE_TL];
    G.T = g[GE_T]; G.K = (int32_t)g[GE_K]; G.pad_key = (uint64_t)g[GE_PAD];
    return G;
}
inline Graphs graphs_of(void** t) {
    Graphs g;
    g.ng = (const int32_t*)t[T_NG]; g.node_off = (const int32_t*)t[T_NODE_OFF]; g.pair_off = (const int64_t*)t[T_PAIR_OFF];
    g.adj_base = (const int32_t*)t[T_ADJ_BASE]; g.adj_off = (const int32_t*)t[T_ADJ_OFF]; g.adj_dst = (const int32_t*)t[T_ADJ_DST];
    g.adj_lab = (const int32_t*)t[T_ADJ_LAB]; g.order = (const int32_t*)t[T_ORDER];
    return g;
}
inline Scratch scratch_of(void** t) {
    Scratch s;
    s.level = (int16_t*)t[T_LEVEL]; s.count = (double*)t[T_COUNT_]; s.head = (int16_t*)t[T_HEAD]; s.tail = (int16_t*)t[T_TAIL];
    s.queue = (int16_t*)t[T_QUEUE]; s.dpred = (int16_t*)t[T_DPRED]; s.dnext = (int16_t*)t[T_DNEXT]; s.dlab = (uint8_t*)t[T_DLAB];
    return s;
}
inline bool geom_ok(const Geom& G) {
    return G.B > 0 && G.n > 1 && G.nmax > 0 && G.nmax <= 32767 && G.emax > 0 && G.emax <= 32767 && G.max_len >= 1 && G.max_len <= 8 &&
           (G.mode == MODE_FIRST || G.mode == MODE_UNIFORM || G.mode == MODE_ALL) && G.S > 0 && G.P > 0 && G.P + N_SPECIAL_ALL <= 0x7fffffffLL;
}
}  // namespace gtos_relbatch_dev
