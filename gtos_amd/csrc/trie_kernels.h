// Per-thread bodies of the GPU path-trie builder (csrc/pathtrie_dev.hip), written so that the SAME code compiles for the host:
// oracle/trie_emul.cpp runs every stage as a serial loop (std::stable_sort / a running sum where the GPU uses rocPRIM) and
// tests/test_pathtrie.py compares the result with csrc_host/pathtrie.cpp array for array.  What the emulation cannot cover is
// the launch glue and the rocPRIM calls of pathtrie_dev.hip.
//
// The algorithm is the fast path of csrc_host/pathtrie.cpp (a path of <= 8 labels below 255 is one 64-bit key, byte 7-t = label
// t + 1) with its two sequential walks turned into scans:
//   sort keys -> per sorted path: length, lcp with the predecessor, "opens a node at level k" bits -> inclusive scan of the bits
//   per level = the node the path sits on at every level -> nodes (label, parent, children range), the node list of every path ->
//   packed order (scan of the one-hot lengths) -> node of every packed row -> rows sorted by node -> chunks, heavy nodes,
//   children-sum indices, wave ranges.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define GTOS_HD __host__ __device__ inline
#else
#define GTOS_HD inline
#endif

namespace gtos_trie {

struct V8 { uint32_t v[8]; };
struct Add8 {
    GTOS_HD V8 operator()(const V8& a, const V8& b) const {
        V8 r;
        for (int i = 0; i < 8; ++i) r.v[i] = a.v[i] + b.v[i];
        return r;
    }
};

GTOS_HD int key_len(uint64_t k) { return 8 - (__builtin_ctzll(k) >> 3); }           // k != 0

// sizes[] written on the device, read by the host once per batch (everything the launches downstream need as host integers)
enum { SZ_ERR = 0, SZ_PF = 1, SZ_SF = 33, SZ_TOTAL = 65 };
// per side, from its base: +0 n_nodes, +1 n_chunks, +2 n_heavy, +3 n_multi, +4..+12 level_off[9], +13..+21 multi_level_off[9]
enum { S_NODES = 0, S_CHUNKS = 1, S_HEAVY = 2, S_MULTI = 3, S_LEVEL = 4, S_MLEVEL = 13 };

struct Side {                 // device (or host) pointers of one trie; arrays sized by the upper bounds of pathtrie_dev.hip
    uint64_t* key;            // [R] packed paths, sorted in place by the caller
    int32_t* order;           // [R] sorted position -> sequence id
    uint8_t* newmask;         // [R] bit k: sorted path i opens a node at level k
    V8* cum;                  // [R] inclusive scan of the bits, per level
    int32_t* node_tab;        // [R, 8] node of sorted path i at level k (-1 past its end)
    int32_t* lvl;             // [9] level offsets
    int64_t* tok;             // [n] label of the node
    int32_t* par;             // [n] parent (n for level 0)
    int64_t* par_long;        // [n] the same as int64 (index_select operand)
    int32_t* child_off;       // [2n], zero-filled by the caller
    int32_t* row_node;        // [N] node of every packed row
    uint32_t* row_key;        // [N] row_node sorted (scratch of the row sort)
    int32_t* rows;            // [N] packed rows sorted by node
    int32_t* off;             // [n + 1] first sorted row of every node
    V8* aux;                  // [n] per node {chunks, heavy flag, several-children flag, 0...}
    V8* aux_cum;              // [n] its inclusive scan
    int32_t *chunk_node, *chunk_start, *chunk_cnt, *chunk_slot, *heavy_node;
    int32_t *sum_idx, *multi_ranges, *wave_off;
};

// ---- stage 1: keys of sequence s (forward and reversed); err != 0 outside the covered case
GTOS_HD void make_keys(int64_t s, int L, int64_t R, const int64_t* bank, const int64_t* length, uint64_t* key_f, uint64_t* key_b,
                       int32_t* id_f, int32_t* id_b, uint8_t* len8, int32_t* err) {
    const int64_t l = length[s];
    uint64_t f = 0, b = 0;
    if (l < 1 || l > L || l > 8) { *err = 1; len8[s] = 1; key_f[s] = key_b[s] = 1ull << 56; id_f[s] = id_b[s] = (int32_t)s; return; }
    for (int t = 0; t < (int)l; ++t) {
        const int64_t v = bank[(int64_t)t * R + s];
        if (v < 0 || v >= 255) { *err = 1; continue; }
        f |= (uint64_t)(v + 1) << (8 * (7 - t));
        b |= (uint64_t)(v + 1) << (8 * (7 - ((int)l - 1 - t)));
    }
    if (!f) f = b = 1ull << 56;                       // (only after an error: keep key_len defined)
    len8[s] = (uint8_t)l;
    key_f[s] = f; key_b[s] = b;
    id_f[s] = id_b[s] = (int32_t)s;
}

// ---- stage 2 (after the key sort): which levels does sorted path i open
GTOS_HD void open_flags(int64_t i, const uint64_t* key, uint8_t* newmask, V8* bits) {
    const uint64_t k = key[i];
    const int len = key_len(k);
    int lcp = 0;
    if (i) {
        const uint64_t x = k ^ key[i - 1];
        lcp = x ? (__builtin_clzll(x) >> 3) : 8;
        if (lcp > len) lcp = len;
    }
    uint8_t m = 0;
    V8 o;
    for (int q = 0; q < 8; ++q) {
        const bool nw = q >= lcp && q < len;
        o.v[q] = nw ? 1u : 0u;
        if (nw) m |= (uint8_t)(1u << q);
    }
    newmask[i] = m;
    bits[i] = o;
}

// ---- stage 3 (one thread): level offsets from the totals of the scan
GTOS_HD void level_offsets(const V8* cum, int64_t R, int32_t* lvl, int32_t* sizes_side) {
    int32_t run = 0;
    for (int q = 0; q < 8; ++q) { lvl[q] = run; sizes_side[S_LEVEL + q] = run; run += (int32_t)cum[R - 1].v[q]; }
    lvl[8] = run;
    sizes_side[S_LEVEL + 8] = run;
    sizes_side[S_NODES] = run;
}

// ---- stage 4: the nodes sorted path i opens, and its node list
GTOS_HD void write_nodes(int64_t i, const Side& t) {
    const uint64_t k = t.key[i];
    const int len = key_len(k);
    const uint8_t m = t.newmask[i];
    const int32_t n = t.lvl[8];
    int32_t below = n;                                  // node at level q - 1 (n: the all-zero state row behind the last node)
    for (int q = 0; q < 8; ++q) {
        int32_t v = -1;
        if (q < len) {
            v = t.lvl[q] + (int32_t)t.cum[i].v[q] - 1;
            if (m & (1u << q)) {
                t.tok[v] = (int64_t)((k >> (8 * (7 - q))) & 0xff) - 1;
                t.par[v] = below;
                t.par_long[v] = below;
            }
            below = v;
        }
        t.node_tab[i * 8 + q] = v;
    }
}

// ---- stage 5: children range of the parent of node v (nodes of a level are sorted by parent)
GTOS_HD void children(int64_t v, const Side& t) {
    const int32_t n = t.lvl[8];
    const int32_t p = t.par[v];
    if (p >= n) return;
    if (v == 0 || t.par[v - 1] != p) t.child_off[2 * p] = (int32_t)v;
    if (v == n - 1 || t.par[v + 1] != p) t.child_off[2 * p + 1] = (int32_t)(v + 1);
}

// ---- packed order: one-hot of the length class of the i-th path in forward lexicographic order (bucket 0 = longest)
GTOS_HD void length_onehot(int64_t i, const int32_t* order_f, const uint8_t* len8, V8* hot) {
    V8 o;
    for (int q = 0; q < 8; ++q) o.v[q] = 0;
    o.v[8 - len8[order_f[i]]] = 1;                       // bucket b holds the paths of 8 - b labels
    hot[i] = o;
}

// one thread: bucket starts, batch sizes (sequences longer than t) and the row offset of every step
GTOS_HD void packed_geometry(const V8* cumlen, int64_t R, int32_t* start, int32_t* batch_sizes, int64_t* offs) {
    int32_t run = 0;
    for (int b = 0; b < 8; ++b) { start[b] = run; run += (int32_t)cumlen[R - 1].v[b]; }
    // sequences longer than t = those of at least t + 1 labels = buckets 0 .. 7 - t
    int64_t o = 0;
    for (int t = 0; t < 8; ++t) {
        int32_t longer = 0;
        for (int b = 0; b <= 7 - t; ++b) longer += (int32_t)cumlen[R - 1].v[b];
        batch_sizes[t] = longer;
        offs[t] = o;
        o += longer;
    }
    offs[8] = o;
}

GTOS_HD void packed_position(int64_t i, const int32_t* order_f, const uint8_t* len8, const V8* cumlen, const int32_t* start,
                             int32_t* seq_order, int32_t* seq_pos, int64_t* seq_order64, int64_t* seq_pos64, int32_t* lexf_of_m) {
    const int32_t s = order_f[i];
    const int b = 8 - len8[s];
    const int32_t m = start[b] + (int32_t)cumlen[i].v[b] - 1;
    seq_order[m] = s; seq_order64[m] = s;
    seq_pos[s] = m; seq_pos64[s] = m;
    lexf_of_m[m] = (int32_t)i;
}

GTOS_HD void lex_position(int64_t i, const int32_t* order_b, int32_t* lexb) { lexb[order_b[i]] = (int32_t)i; }

// ---- node of every packed row of the sequence at packed position m
GTOS_HD void fill_rows(int64_t m, const int32_t* seq_order, const uint8_t* len8, const int32_t* lexf_of_m, const int32_t* lexb,
                       const int64_t* offs, const int32_t* tab_f, const int32_t* tab_b, int32_t* row_pf, int32_t* row_sf) {
    const int32_t s = seq_order[m];
    const int len = len8[s];
    const int32_t* nf = tab_f + (int64_t)lexf_of_m[m] * 8;
    const int32_t* nb = tab_b + (int64_t)lexb[s] * 8;
    for (int q = 0; q < len; ++q) {
        row_pf[offs[q] + m] = nf[q];
        row_sf[offs[len - 1 - q] + m] = nb[q];
    }
}

// ---- rows sorted by node: first sorted row of every node (every node has at least one row)
GTOS_HD void node_offsets(int64_t e, int64_t N, const uint32_t* row_key, int32_t* off, int32_t n) {
    if (e == 0 || row_key[e] != row_key[e - 1]) off[row_key[e]] = (int32_t)e;
    if (e == N - 1) off[n] = (int32_t)N;
}

// per node: chunks, heavy flag, several-children flag (the scan's input)
GTOS_HD void node_counts(int64_t u, int chunk, const Side& t) {
    const int32_t cnt = t.off[u + 1] - t.off[u];
    const int32_t nch = cnt > 0 ? (cnt + chunk - 1) / chunk : 1;
    V8 o;
    for (int q = 0; q < 8; ++q) o.v[q] = 0;
    o.v[0] = (uint32_t)nch;
    o.v[1] = nch > 1;
    o.v[2] = t.child_off[2 * u + 1] - t.child_off[2 * u] >= 2;
    t.aux[u] = o;
}

// after the inclusive scan of aux into aux_cum: chunk records, heavy nodes, children-sum indices of node u
GTOS_HD void node_records(int64_t u, int chunk, const Side& t) {
    const int32_t n = t.lvl[8];
    const V8 inc = t.aux_cum[u], own = t.aux[u];
    const int32_t base = (int32_t)(inc.v[0] - own.v[0]), nch = (int32_t)own.v[0];
    const int32_t lo = t.off[u], hi = t.off[u + 1];
    int32_t slot = -1;
    if (own.v[1]) {
        slot = (int32_t)(inc.v[1] - 1);
        t.heavy_node[slot] = (int32_t)u;
    }
    for (int32_t c = 0; c < nch; ++c) {
        const int32_t s = lo + c * chunk;
        int32_t k = hi - s;
        k = k < 0 ? 0 : (k > chunk ? chunk : k);
        t.chunk_node[base + c] = (int32_t)u;
        t.chunk_start[base + c] = s;
        t.chunk_cnt[base + c] = k;
        t.chunk_slot[base + c] = slot;
    }
    const int32_t c0 = t.child_off[2 * u], c1 = t.child_off[2 * u + 1];
    const int32_t nc = c1 - c0;
    if (own.v[2]) {
        const int32_t j = (int32_t)(inc.v[2] - 1);
        t.multi_ranges[2 * j] = c0;
        t.multi_ranges[2 * j + 1] = c1;
        t.sum_idx[u] = n + 1 + j;
    } else {
        t.sum_idx[u] = nc == 1 ? c0 : n;
    }
}

// one thread: the counts of this trie and how many several-children nodes precede every level
GTOS_HD void side_sizes(const Side& t, int32_t* sizes_side) {
    const int32_t n = t.lvl[8];
    const V8 last = t.aux_cum[n - 1];
    sizes_side[S_CHUNKS] = (int32_t)last.v[0];
    sizes_side[S_HEAVY] = (int32_t)last.v[1];
    sizes_side[S_MULTI] = (int32_t)last.v[2];
    for (int q = 0; q <= 8; ++q) {
        const int32_t at = t.lvl[q];
        sizes_side[S_MLEVEL + q] = at > 0 ? (int32_t)t.aux_cum[at - 1].v[2] : 0;
    }
}

// ---- wave ranges of the streaming segmented sum: first chunk whose start is >= w * rows_per_wave
GTOS_HD void wave_range(int64_t w, int64_t n_waves, int rows_per_wave, const Side& t, const int32_t* sizes_side) {
    const int32_t nc = sizes_side[S_CHUNKS];
    if (w == n_waves) { t.wave_off[w] = nc; return; }
    if (w == 0) { t.wave_off[0] = 0; return; }
    const int64_t target = w * (int64_t)rows_per_wave;
    int32_t lo = 0, hi = nc;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (t.chunk_start[mid] < target) lo = mid + 1; else hi = mid;
    }
    t.wave_off[w] = lo;
}

// ---- the pointer tables of the two-phase C ABI (gtos_pathtrie_dev_phase_a / _b and their emulation)
// common[]: buffers shared by both tries
enum { C_LEN8 = 0,        // uint8 [R]
       C_SCRATCH,         // V8    [R]   input of the scans
       C_CUMLEN,          // V8    [R]   inclusive scan of the length one-hots over the forward lexicographic order
       C_START,           // int32 [8]
       C_BATCH,           // int32 [8]   sequences longer than t
       C_OFFS,            // int64 [9]   first packed row of step t
       C_SEQ_ORDER,       // int32 [R]
       C_SEQ_POS,         // int32 [R]
       C_SEQ_ORDER64,     // int64 [R]
       C_SEQ_POS64,       // int64 [R]
       C_LEXF,            // int32 [R]   forward lexicographic index of the path at packed position m
       C_LEXB,            // int32 [R]   backward lexicographic index of sequence s
       C_ROW_PF,          // int32 [N]
       C_ROW_SF,          // int32 [N]
       C_KEY_ALT,         // uint64 [R]  second buffer of the key sort
       C_ID_ALT,          // int32 [R]
       C_IOTA,            // int32 [N]   0 .. N-1 (values of the row sort)
       C_COUNT };
// side[]: one trie (phase A uses the R-sized entries up to T_LVL; phase B needs the rest, sized from sizes[])
enum { T_KEY = 0,         // uint64 [R]
       T_ORDER,           // int32 [R]
       T_NEWMASK,         // uint8 [R]
       T_CUM,             // V8    [R]
       T_NODE_TAB,        // int32 [8 R]
       T_LVL,             // int32 [9]
       T_TOK,             // int64 [n]
       T_PAR,             // int32 [n]
       T_PAR_LONG,        // int64 [n]
       T_CHILD_OFF,       // int32 [2 n]   ZERO-FILLED by the caller
       T_ROW_KEY,         // uint32 [N]
       T_ROWS,            // int32 [N]
       T_OFF,             // int32 [n + 1]
       T_AUX,             // V8    [n]
       T_AUX_CUM,         // V8    [n]
       T_CHUNK_NODE, T_CHUNK_START, T_CHUNK_CNT, T_CHUNK_SLOT,   // int32 [n + N / chunk + 1]
       T_HEAVY_NODE,      // int32 [N / chunk + 1]
       T_SUM_IDX,         // int32 [n]
       T_MULTI_RANGES,    // int32 [2 (n / 2 + 1)]
       T_WAVE_OFF,        // int32 [n_waves + 1]
       T_COUNT };

inline Side side_of(void** t) {
    Side s;
    s.key = (uint64_t*)t[T_KEY]; s.order = (int32_t*)t[T_ORDER]; s.newmask = (uint8_t*)t[T_NEWMASK]; s.cum = (V8*)t[T_CUM];
    s.node_tab = (int32_t*)t[T_NODE_TAB]; s.lvl = (int32_t*)t[T_LVL]; s.tok = (int64_t*)t[T_TOK]; s.par = (int32_t*)t[T_PAR];
    s.par_long = (int64_t*)t[T_PAR_LONG]; s.child_off = (int32_t*)t[T_CHILD_OFF]; s.row_node = nullptr;
    s.row_key = (uint32_t*)t[T_ROW_KEY]; s.rows = (int32_t*)t[T_ROWS]; s.off = (int32_t*)t[T_OFF]; s.aux = (V8*)t[T_AUX];
    s.aux_cum = (V8*)t[T_AUX_CUM]; s.chunk_node = (int32_t*)t[T_CHUNK_NODE]; s.chunk_start = (int32_t*)t[T_CHUNK_START];
    s.chunk_cnt = (int32_t*)t[T_CHUNK_CNT]; s.chunk_slot = (int32_t*)t[T_CHUNK_SLOT]; s.heavy_node = (int32_t*)t[T_HEAVY_NODE];
    s.sum_idx = (int32_t*)t[T_SUM_IDX]; s.multi_ranges = (int32_t*)t[T_MULTI_RANGES]; s.wave_off = (int32_t*)t[T_WAVE_OFF];
    return s;
}

}  // namespace gtos_trie
