// libgtos_host.so, second half: the prefix / suffix tries of a batch's relation bank (include/gtos_host.h).
//
// The relation bank of a batch (generator/data.py:166-176: relation_bank[L,R], relation_length[R]) is a set of label
// paths out of BFS trees, so it is (almost) prefix- and suffix-closed: R paths with sum(len) = 5.7 R tokens have only ~R
// distinct prefixes and ~1.05 R distinct suffixes.  The first GRU layer of RelationEncoder (generator/encoder.py:93-111)
// in the forward direction depends only on the prefix read so far, in the reverse direction only on the suffix, so it
// is evaluated once per TRIE NODE; and the second layer's input-gate product splits into a prefix-node term plus a
// suffix-node term (gtos_amd/gru.py).  This file builds, in O(sum(len)) after one sort per trie:
//   * the packed sequence order of the second layer (length descending, then lexicographic),
//   * both tries, level-major (level k = prefixes / suffixes of k+1 tokens), nodes of a level in lexicographic order, so
//     the children of a node are a contiguous range of the next level,
//   * the node of every packed row (sequence s, position t) in both tries,
//   * the rows of every node (CSR), cut into chunks of <= `chunk` rows for the segmented gradient reduction.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>

#include "../../include/gtos_host.h"

namespace {

struct Trie {
    int64_t n_nodes = 0;
    std::vector<int64_t> level_off;          // [L+1]
    std::vector<int32_t> tok, par;           // per node; par = n_nodes for level-0 nodes (the all-zero state row)
    std::vector<int32_t> child_off;          // [2*n_nodes]: children of node u are nodes child_off[2u] .. child_off[2u+1]-1
    // rows of every node, chunked
    std::vector<int32_t> rows;               // [N] packed rows sorted by node
    std::vector<int32_t> chunk_node, chunk_start, chunk_cnt, chunk_slot, heavy_node;
};

struct Seqs {
    int L = 0;
    int64_t R = 0;
    std::vector<int32_t> tok;                // [R*L] row-major per sequence, 0-padded
    std::vector<int32_t> len;                // [R]
};

// lexicographic order of the sequences (a proper prefix sorts first); stable in the sequence id
std::vector<int32_t> lex_order(const Seqs& q) {
    std::vector<int32_t> idx(q.R);
    std::iota(idx.begin(), idx.end(), 0);
    const int L = q.L;
    bool small = L <= 8;
    if (small)
        for (int64_t i = 0; i < q.R * L && small; ++i) small = q.tok[i] >= 0 && q.tok[i] < 255;
    if (small) {
        // pack (token + 1) bytes, most significant first: integer order == lexicographic order with "shorter first";
        // LSD radix sort of (key, id) pairs, one pass per byte position that is not constant (stable: ties keep id order)
        std::vector<uint64_t> key(q.R), key2(q.R);
        std::vector<int32_t> idx2(q.R);
        uint64_t all_or = 0, all_and = ~0ull;
        for (int64_t s = 0; s < q.R; ++s) {
            uint64_t k = 0;
            for (int t = 0; t < 8; ++t) k = (k << 8) | (uint64_t)((t < L && t < q.len[s]) ? q.tok[s * L + t] + 1 : 0);
            key[s] = k;
            all_or |= k;
            all_and &= k;
        }
        for (int byte = 0; byte < 8; ++byte) {
            const int sh = byte * 8;
            if ((((all_or ^ all_and) >> sh) & 0xff) == 0) continue;       // the same digit everywhere
            int64_t cnt[257] = {0};
            for (int64_t i = 0; i < q.R; ++i) cnt[((key[i] >> sh) & 0xff) + 1]++;
            for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
            for (int64_t i = 0; i < q.R; ++i) {
                const int64_t o = cnt[(key[i] >> sh) & 0xff]++;
                key2[o] = key[i];
                idx2[o] = idx[i];
            }
            key.swap(key2);
            idx.swap(idx2);
        }
    } else {
        std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) {
            const int la = q.len[a], lb = q.len[b], m = la < lb ? la : lb;
            for (int t = 0; t < m; ++t) {
                const int32_t x = q.tok[(int64_t)a * L + t], y = q.tok[(int64_t)b * L + t];
                if (x != y) return x < y;
            }
            return la < lb;
        });
    }
    return idx;
}

// One pass over the lexicographically sorted sequences: sequence i shares its first lcp(i-1, i) nodes with its
// predecessor and opens a new node at every later position.  A first pass over the sorted order takes the lcp's and
// counts the nodes per level, so the second pass numbers the nodes with their final ids (level-major, lexicographic inside a
// level) and writes the node of every packed row straight into `row_node`: row of (sequence s, level k) =
// offs[pos(k, len)] + seq_pos[s] with pos = k for the prefix trie and len - 1 - k for the suffix trie (`reversed`).
void build_trie(const Seqs& q, const std::vector<int32_t>& order, const std::vector<int32_t>& seq_pos, const std::vector<int64_t>& offs,
                bool reversed, Trie& tr, std::vector<int32_t>& row_node) {
    const int L = q.L;
    const int64_t R = q.R;
    std::vector<uint8_t> lcps(R);
    std::vector<int64_t> per_level(L + 1, 0);
    {
        const int32_t* prev = nullptr;
        int prev_len = 0;
        for (int64_t i = 0; i < R; ++i) {
            const int32_t s = order[i];
            const int32_t* t = &q.tok[(int64_t)s * L];
            const int len = q.len[s];
            int lcp = 0;
            if (prev) {
                const int m = len < prev_len ? len : prev_len;
                while (lcp < m && t[lcp] == prev[lcp]) ++lcp;
            }
            lcps[i] = (uint8_t)lcp;
            for (int k = lcp; k < len; ++k) per_level[k]++;
            prev = t;
            prev_len = len;
        }
    }
    tr.level_off.assign(L + 1, 0);
    for (int k = 0; k < L; ++k) tr.level_off[k + 1] = tr.level_off[k] + per_level[k];
    const int64_t next = tr.level_off[L];
    tr.n_nodes = next;
    tr.tok.resize(next);
    tr.par.resize(next);
    std::vector<int64_t> fill(tr.level_off.begin(), tr.level_off.end() - 1);   // next free node id per level
    std::vector<int32_t> cur(L, -1);                                          // node of the previous sequence at each level
    for (int64_t i = 0; i < R; ++i) {
        const int32_t s = order[i];
        const int32_t* t = &q.tok[(int64_t)s * L];
        const int len = q.len[s], lcp = lcps[i];
        for (int k = lcp; k < len; ++k) {
            const int32_t v = (int32_t)fill[k]++;
            cur[k] = v;
            tr.tok[v] = t[k];
            tr.par[v] = k ? cur[k - 1] : (int32_t)next;       // level 0: the all-zero row behind the last node
        }
        const int64_t m = seq_pos[s];
        if (!reversed) for (int k = 0; k < len; ++k) row_node[offs[k] + m] = cur[k];
        else           for (int k = 0; k < len; ++k) row_node[offs[len - 1 - k] + m] = cur[k];
    }
    // children of a node: a contiguous range of the next level (nodes of a level are sorted by parent, then token);
    // stored as [start, end) pairs, start == end for leaves
    tr.child_off.assign(2 * next, 0);
    for (int64_t v = 0; v < next; ++v) {
        const int64_t p = tr.par[v];
        if (p >= next) continue;
        if (tr.child_off[2 * p + 1] == 0) tr.child_off[2 * p] = (int32_t)v;
        tr.child_off[2 * p + 1] = (int32_t)(v + 1);
    }
}

// rows of every node (counting sort by node), cut into chunks
void build_rows(const std::vector<int32_t>& row_node, int64_t n_nodes, int chunk, Trie& tr) {
    const int64_t N = (int64_t)row_node.size();
    std::vector<int64_t> off(n_nodes + 1, 0);
    for (int64_t p = 0; p < N; ++p) off[row_node[p] + 1]++;
    for (int64_t u = 0; u < n_nodes; ++u) off[u + 1] += off[u];
    tr.rows.resize(N);
    {
        std::vector<int64_t> cur(off.begin(), off.end() - 1);
        for (int64_t p = 0; p < N; ++p) tr.rows[cur[row_node[p]]++] = (int32_t)p;
    }
    tr.chunk_node.clear(); tr.chunk_start.clear(); tr.chunk_cnt.clear(); tr.chunk_slot.clear(); tr.heavy_node.clear();
    {
        const size_t cap = (size_t)(n_nodes + N / chunk + 1);
        tr.chunk_node.reserve(cap); tr.chunk_start.reserve(cap); tr.chunk_cnt.reserve(cap); tr.chunk_slot.reserve(cap);
    }
    for (int64_t u = 0; u < n_nodes; ++u) {
        const int64_t lo = off[u], hi = off[u + 1];
        const int64_t nch = hi > lo ? (hi - lo + chunk - 1) / chunk : 1;      // a node without rows still gets a (zero) result
        int32_t slot = -1;
        if (nch > 1) {
            slot = (int32_t)tr.heavy_node.size();
            tr.heavy_node.push_back((int32_t)u);
        }
        for (int64_t c = 0; c < nch; ++c) {
            const int64_t s = lo + c * chunk;
            tr.chunk_node.push_back((int32_t)u);
            tr.chunk_start.push_back((int32_t)s);
            tr.chunk_cnt.push_back((int32_t)std::max<int64_t>(0, std::min<int64_t>(chunk, hi - s)));
            tr.chunk_slot.push_back(slot);
        }
    }
}

// ------------------------------------------------------------------------------------------------ fast path (L <= 8, labels < 255)
// A path is ONE 64-bit key: byte 7-t holds label t + 1, 0 behind the end -- integer order == lexicographic order with "a proper
// prefix first".  Everything the generic path reads from the [R, L] token rows through the sorted order (random 32-byte reads)
// comes out of the sorted keys themselves, front to back: length = 8 - trailing zero bytes, lcp with the predecessor = leading
// zero bytes of the XOR, label k = byte 7-k.
struct KeyId { uint64_t key; int64_t id; };

// LSD radix sort of (key, id), stable in id; the histograms of all eight bytes come from ONE pass, bytes that are the same
// everywhere are skipped (two of eight at C2 have a handful of values, none is constant)
void sort_keys(std::vector<KeyId>& a) {
    const size_t n = a.size();
    std::vector<uint32_t> hist(8 * 256, 0);
    for (size_t i = 0; i < n; ++i) {
        const uint64_t k = a[i].key;
        for (int b = 0; b < 8; ++b) hist[b * 256 + ((k >> (8 * b)) & 0xff)]++;
    }
    std::vector<KeyId> tmp(n);
    KeyId *src = a.data(), *dst = tmp.data();
    for (int b = 0; b < 8; ++b) {
        uint32_t* h = &hist[b * 256];
        bool constant = false;
        for (int d = 0; d < 256; ++d) if (h[d] == n) constant = true;
        if (constant) continue;
        uint32_t pos[256];
        uint32_t run = 0;
        for (int d = 0; d < 256; ++d) { pos[d] = run; run += h[d]; }
        const int sh = 8 * b;
        for (size_t i = 0; i < n; ++i) dst[pos[(src[i].key >> sh) & 0xff]++] = src[i];
        std::swap(src, dst);
    }
    if (src != a.data()) std::memcpy(a.data(), src, n * sizeof(KeyId));
}

inline int key_len(uint64_t k) { return 8 - (__builtin_ctzll(k) >> 3); }          // k != 0: every path has a label

// The trie of the sorted keys: nodes numbered level-major, lexicographic inside a level (as build_trie), `node_tab[i*8 + k]` = the
// node of the i-th sorted path at level k, `cnt[u]` = paths through node u (= its rows in the packed layout).
void trie_of_sorted_keys(const std::vector<KeyId>& a, int L, Trie& tr, std::vector<int32_t>& node_tab, std::vector<int64_t>& cnt) {
    const int64_t R = (int64_t)a.size();
    std::vector<uint8_t> lcps(R);
    int64_t per_level[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t prev = 0;
    for (int64_t i = 0; i < R; ++i) {
        const uint64_t k = a[i].key;
        const int len = key_len(k);
        int lcp = 0;
        if (i) {
            const uint64_t x = k ^ prev;
            lcp = x ? (__builtin_clzll(x) >> 3) : 8;
            if (lcp > len) lcp = len;
        }
        lcps[i] = (uint8_t)lcp;
        for (int q = lcp; q < len; ++q) per_level[q]++;
        prev = k;
    }
    tr.level_off.assign(L + 1, 0);
    for (int q = 0; q < L; ++q) tr.level_off[q + 1] = tr.level_off[q] + (q < 8 ? per_level[q] : 0);
    const int64_t next = tr.level_off[L];
    tr.n_nodes = next;
    tr.tok.resize(next);
    tr.par.resize(next);
    cnt.assign(next + 1, 0);
    node_tab.resize(R * 8);
    int64_t fill[8];
    for (int q = 0; q < 8; ++q) fill[q] = q < L ? tr.level_off[q] : next;
    int32_t cur[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    for (int64_t i = 0; i < R; ++i) {
        const uint64_t k = a[i].key;
        const int len = key_len(k), lcp = lcps[i];
        for (int q = lcp; q < len; ++q) {
            const int32_t v = (int32_t)fill[q]++;
            cur[q] = v;
            tr.tok[v] = (int32_t)((k >> (8 * (7 - q))) & 0xff) - 1;
            tr.par[v] = q ? cur[q - 1] : (int32_t)next;       // level 0: the all-zero row behind the last node
        }
        int32_t* nt = &node_tab[i * 8];
        for (int q = 0; q < 8; ++q) nt[q] = q < len ? cur[q] : -1;
        for (int q = 0; q < len; ++q) cnt[cur[q]]++;
    }
    tr.child_off.assign(2 * next, 0);
    for (int64_t v = 0; v < next; ++v) {
        const int64_t p = tr.par[v];
        if (p >= next) continue;
        if (tr.child_off[2 * p + 1] == 0) tr.child_off[2 * p] = (int32_t)v;
        tr.child_off[2 * p + 1] = (int32_t)(v + 1);
    }
}

// rows of every node from the per-node counts (no histogram pass over the rows), cut into chunks
void build_rows_counted(const std::vector<int32_t>& row_node, const std::vector<int64_t>& cnt, int64_t n_nodes, int chunk, Trie& tr,
                        bool by_level) {
    const int64_t N = (int64_t)row_node.size();
    std::vector<int64_t> off(n_nodes + 1, 0);
    for (int64_t u = 0; u < n_nodes; ++u) off[u + 1] = off[u] + cnt[u];
    tr.rows.resize(N);
    if (!by_level) {
        // prefix trie: the rows of level k are one contiguous range of p, so the scatter already stays inside one level's rows
        std::vector<int64_t> cur(off.begin(), off.end() - 1);
        for (int64_t p = 0; p < N; ++p) tr.rows[cur[row_node[p]]++] = (int32_t)p;
    } else {
        // suffix trie: a step's rows mix all levels, and a one-pass scatter jumps over the whole 10 MB row list.  First the rows
        // of every level (ascending p, sequential writes), then the scatter level by level -- inside a cache-sized window
        const int nl = (int)tr.level_off.size() - 1;
        std::vector<uint8_t> lvl(n_nodes);
        std::vector<int64_t> lo(nl + 1, 0);
        for (int q = 0; q < nl; ++q) {
            for (int64_t u = tr.level_off[q]; u < tr.level_off[q + 1]; ++u) lvl[u] = (uint8_t)q;
            lo[q + 1] = off[tr.level_off[q + 1]];                           // rows of the levels below q + 1
        }
        std::vector<int32_t> by(N);
        {
            std::vector<int64_t> w(lo.begin(), lo.end() - 1);
            for (int64_t p = 0; p < N; ++p) by[w[lvl[row_node[p]]]++] = (int32_t)p;
        }
        std::vector<int64_t> cur(off.begin(), off.end() - 1);
        for (int64_t e = 0; e < N; ++e) { const int32_t p = by[e]; tr.rows[cur[row_node[p]]++] = p; }
    }
    tr.chunk_node.clear(); tr.chunk_start.clear(); tr.chunk_cnt.clear(); tr.chunk_slot.clear(); tr.heavy_node.clear();
    {
        const size_t cap = (size_t)(n_nodes + N / chunk + 1);
        tr.chunk_node.reserve(cap); tr.chunk_start.reserve(cap); tr.chunk_cnt.reserve(cap); tr.chunk_slot.reserve(cap);
    }
    for (int64_t u = 0; u < n_nodes; ++u) {
        const int64_t lo = off[u], hi = off[u + 1];
        const int64_t nch = hi > lo ? (hi - lo + chunk - 1) / chunk : 1;
        int32_t slot = -1;
        if (nch > 1) {
            slot = (int32_t)tr.heavy_node.size();
            tr.heavy_node.push_back((int32_t)u);
        }
        for (int64_t c = 0; c < nch; ++c) {
            const int64_t s = lo + c * chunk;
            tr.chunk_node.push_back((int32_t)u);
            tr.chunk_start.push_back((int32_t)s);
            tr.chunk_cnt.push_back((int32_t)std::max<int64_t>(0, std::min<int64_t>(chunk, hi - s)));
            tr.chunk_slot.push_back(slot);
        }
    }
}

}  // namespace

struct gtos_pathtrie {
    int L = 0;
    int64_t R = 0, N = 0;
    std::vector<int32_t> batch_sizes;        // [L]
    std::vector<int32_t> seq_order, seq_pos; // sorted position -> sequence id, sequence id -> sorted position
    std::vector<int32_t> row_pf, row_sf;     // [N]
    Trie pf, sf;
};

extern "C" gtos_pathtrie* gtos_pathtrie_build(int L, int64_t R, const int64_t* bank, const int64_t* length, int chunk) {
    if (L <= 0 || L > 64 || R <= 0 || !bank || !length || chunk <= 0 || chunk > 64 || R > 0x7fffffffLL / L) return nullptr;
    // GTOS_TRIE_TIMING=1: phase times of this call on stderr (unpack, sort, packed order, tries, row lists)
    static const bool timing = getenv("GTOS_TRIE_TIMING") && getenv("GTOS_TRIE_TIMING")[0] == '1';
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "pathtrie %-12s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    if (L <= 8) {
        // ---- fast path: every path as one 64-bit key (see sort_keys / trie_of_sorted_keys above)
        std::vector<KeyId> kf(R), kb(R);
        std::vector<uint8_t> len8(R);
        int64_t N = 0;
        int maxlen = 0;
        bool fast = true;
        for (int64_t s = 0; s < R && fast; ++s) {
            const int64_t l = length[s];
            if (l < 1 || l > L) return nullptr;
            len8[s] = (uint8_t)l;
            N += l;
            maxlen = std::max<int>(maxlen, (int)l);
            uint64_t f = 0, b = 0;
            for (int t = 0; t < (int)l; ++t) {                               // bank is [L, R]: L strided streams, one pass over the keys
                const int64_t v = bank[(int64_t)t * R + s];
                if (v < 0 || v > 0x7fffffff) return nullptr;
                if (v >= 255) { fast = false; break; }                      // a label the one-byte digits cannot hold: generic path
                f |= (uint64_t)(v + 1) << (8 * (7 - t));
                b |= (uint64_t)(v + 1) << (8 * (7 - ((int)l - 1 - t)));
            }
            kf[s] = {f, s};
            kb[s] = {b, s};
        }
        if (fast && N > 0x7fffffffLL) return nullptr;
        if (fast) {
            auto* h = new gtos_pathtrie();
            h->L = maxlen;
            h->R = R;
            h->N = N;
            lap("unpack");
            {
                std::thread tb([&] { sort_keys(kb); });
                sort_keys(kf);
                tb.join();
            }
            lap("sort");
            // packed order: length descending, then lexicographic = counting sort by length over the lexicographic order; the
            // lexicographic index of every packed position comes with it
            h->seq_order.resize(R);
            h->seq_pos.resize(R);
            std::vector<int32_t> lexf_of_m(R), lexb(R);
            {
                std::vector<int64_t> start(L + 2, 0);
                for (int64_t s = 0; s < R; ++s) start[L - len8[s] + 1]++;
                for (int b = 0; b <= L; ++b) start[b + 1] += start[b];
                for (int64_t i = 0; i < R; ++i) {
                    const int64_t s = kf[i].id;
                    const int64_t m = start[L - len8[s]]++;
                    h->seq_order[m] = (int32_t)s;
                    lexf_of_m[m] = (int32_t)i;
                    h->seq_pos[s] = (int32_t)m;
                }
                for (int64_t i = 0; i < R; ++i) lexb[kb[i].id] = (int32_t)i;
            }
            h->batch_sizes.assign(maxlen, 0);
            for (int64_t s = 0; s < R; ++s) h->batch_sizes[len8[s] - 1]++;
            for (int t = maxlen - 2; t >= 0; --t) h->batch_sizes[t] += h->batch_sizes[t + 1];
            std::vector<int64_t> offs(L + 1, 0);
            for (int t = 0; t < maxlen; ++t) offs[t + 1] = offs[t] + h->batch_sizes[t];
            lap("packed order");
            std::vector<int32_t> tab_f, tab_b;
            std::vector<int64_t> cnt_f, cnt_b;
            h->row_pf.resize(N);
            h->row_sf.resize(N);
            {
                // the node of every packed row, written in PACKED order: one random 32-byte read of the path's node list, up to
                // eight sequential write streams (the generic path scatters 4-byte writes from the lexicographic order)
                std::thread tb([&] {
                    trie_of_sorted_keys(kb, L, h->sf, tab_b, cnt_b);
                    for (int64_t m = 0; m < R; ++m) {
                        const int64_t s = h->seq_order[m];
                        const int len = len8[s];
                        const int32_t* nt = &tab_b[(int64_t)lexb[s] * 8];
                        for (int q = 0; q < len; ++q) h->row_sf[offs[len - 1 - q] + m] = nt[q];
                    }
                });
                trie_of_sorted_keys(kf, L, h->pf, tab_f, cnt_f);
                for (int64_t m = 0; m < R; ++m) {
                    const int len = len8[h->seq_order[m]];
                    const int32_t* nt = &tab_f[(int64_t)lexf_of_m[m] * 8];
                    for (int q = 0; q < len; ++q) h->row_pf[offs[q] + m] = nt[q];
                }
                tb.join();
            }
            lap("tries");
            {
                std::thread tr([&] { build_rows_counted(h->row_sf, cnt_b, h->sf.n_nodes, chunk, h->sf, true); });
                build_rows_counted(h->row_pf, cnt_f, h->pf.n_nodes, chunk, h->pf, false);
                tr.join();
            }
            lap("row lists");
            h->pf.level_off.resize(maxlen + 1);
            h->sf.level_off.resize(maxlen + 1);
            h->pf.level_off[maxlen] = h->pf.n_nodes;
            h->sf.level_off[maxlen] = h->sf.n_nodes;
            return h;
        }
    }
    Seqs fw, bw;
    fw.L = bw.L = L;
    fw.R = bw.R = R;
    fw.tok.assign(R * L, 0);
    bw.tok.assign(R * L, 0);
    fw.len.resize(R);
    bw.len.resize(R);
    int64_t N = 0;
    int maxlen = 0;
    for (int64_t s = 0; s < R; ++s) {
        const int64_t l = length[s];
        if (l < 1 || l > L) return nullptr;
        fw.len[s] = bw.len[s] = (int32_t)l;
        N += l;
        maxlen = std::max<int>(maxlen, (int)l);
        for (int t = 0; t < l; ++t) {
            const int64_t v = bank[(int64_t)t * R + s];
            if (v < 0 || v > 0x7fffffff) return nullptr;
            fw.tok[s * L + t] = (int32_t)v;
            bw.tok[s * L + (l - 1 - t)] = (int32_t)v;
        }
    }
    if (N > 0x7fffffffLL) return nullptr;
    auto* h = new gtos_pathtrie();
    h->L = maxlen;
    h->R = R;
    h->N = N;
    lap("unpack");
    std::vector<int32_t> ord_f, ord_b;
    {
        std::thread tb([&] { ord_b = lex_order(bw); });
        ord_f = lex_order(fw);
        tb.join();
    }
    lap("sort");
    // packed order: length descending, then lexicographic = counting sort by length over the sequences in lexicographic order
    h->seq_order.resize(R);
    {
        std::vector<int64_t> start(L + 2, 0);
        for (int64_t s = 0; s < R; ++s) start[L - fw.len[s] + 1]++;               // bucket L - len: longest first
        for (int b = 0; b <= L; ++b) start[b + 1] += start[b];
        for (int64_t i = 0; i < R; ++i) {
            const int32_t s = ord_f[i];
            h->seq_order[start[L - fw.len[s]]++] = s;
        }
    }
    h->seq_pos.resize(R);
    for (int64_t i = 0; i < R; ++i) h->seq_pos[h->seq_order[i]] = (int32_t)i;
    h->batch_sizes.assign(maxlen, 0);
    for (int64_t s = 0; s < R; ++s) h->batch_sizes[fw.len[s] - 1]++;                 // sequences of exactly this length ...
    for (int t = maxlen - 2; t >= 0; --t) h->batch_sizes[t] += h->batch_sizes[t + 1]; // ... -> sequences longer than t
    std::vector<int64_t> offs(L + 1, 0);                  // rows of step t are the first batch_sizes[t] sequences of the order
    for (int t = 0; t < maxlen; ++t) offs[t + 1] = offs[t] + h->batch_sizes[t];
    h->row_pf.resize(N);
    h->row_sf.resize(N);
    lap("packed order");
    {
        std::thread tb([&] { build_trie(bw, ord_b, h->seq_pos, offs, true, h->sf, h->row_sf); });
        build_trie(fw, ord_f, h->seq_pos, offs, false, h->pf, h->row_pf);
        tb.join();
    }
    lap("tries");
    std::thread tr([&] { build_rows(h->row_sf, h->sf.n_nodes, chunk, h->sf); });
    build_rows(h->row_pf, h->pf.n_nodes, chunk, h->pf);
    tr.join();
    lap("row lists");
    // level offsets beyond the longest sequence collapse
    h->pf.level_off.resize(maxlen + 1);
    h->sf.level_off.resize(maxlen + 1);
    h->pf.level_off[maxlen] = h->pf.n_nodes;
    h->sf.level_off[maxlen] = h->sf.n_nodes;
    return h;
}

// sizes[0..]: L, R, N, nPF, nSF, pf chunks, pf heavy, sf chunks, sf heavy
extern "C" int gtos_pathtrie_sizes(const gtos_pathtrie* h, int64_t* sizes) {
    if (!h || !sizes) return -1;
    sizes[0] = h->L; sizes[1] = h->R; sizes[2] = h->N; sizes[3] = h->pf.n_nodes; sizes[4] = h->sf.n_nodes;
    sizes[5] = (int64_t)h->pf.chunk_node.size(); sizes[6] = (int64_t)h->pf.heavy_node.size();
    sizes[7] = (int64_t)h->sf.chunk_node.size(); sizes[8] = (int64_t)h->sf.heavy_node.size();
    return 0;
}

namespace {
template <typename V>
void put(int32_t* dst, const V& v) { if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(int32_t)); }
}

extern "C" int gtos_pathtrie_export(const gtos_pathtrie* h, int32_t** out) {
    if (!h || !out) return -1;
    int k = 0;
    put(out[k++], h->batch_sizes);
    put(out[k++], h->seq_order);
    put(out[k++], h->seq_pos);
    put(out[k++], h->row_pf);
    put(out[k++], h->row_sf);
    for (const Trie* t : {&h->pf, &h->sf}) {
        std::vector<int32_t> lo(t->level_off.begin(), t->level_off.end());
        put(out[k++], lo);
        put(out[k++], t->tok);
        put(out[k++], t->par);
        put(out[k++], t->child_off);
        put(out[k++], t->rows);
        put(out[k++], t->chunk_node);
        put(out[k++], t->chunk_start);
        put(out[k++], t->chunk_cnt);
        put(out[k++], t->chunk_slot);
        put(out[k++], t->heavy_node);
    }
    return k;
}

namespace {
int64_t count_multi(const Trie& t) {
    int64_t m = 0;
    for (int64_t u = 0; u < t.n_nodes; ++u) m += t.child_off[2 * u + 1] - t.child_off[2 * u] >= 2;
    return m;
}
int64_t waves_of(int64_t N, int rows_per_wave) { return std::max<int64_t>(1, (N + rows_per_wave - 1) / rows_per_wave); }
}  // namespace

extern "C" int gtos_pathtrie_derived_sizes(const gtos_pathtrie* h, int rows_per_wave, int64_t* sizes) {
    if (!h || !sizes || rows_per_wave <= 0) return -1;
    sizes[0] = count_multi(h->pf); sizes[1] = waves_of(h->N, rows_per_wave);
    sizes[2] = count_multi(h->sf); sizes[3] = waves_of(h->N, rows_per_wave);
    return 0;
}

extern "C" int gtos_pathtrie_export_derived(const gtos_pathtrie* h, int rows_per_wave, int32_t** out) {
    if (!h || !out || rows_per_wave <= 0) return -1;
    int k = 0;
    for (const Trie* t : {&h->pf, &h->sf}) {
        const int64_t n = t->n_nodes;
        int32_t *sum_idx = out[k], *ranges = out[k + 1], *mlo = out[k + 2], *wave = out[k + 3];
        k += 4;
        int64_t multi = 0;
        size_t lvl = 0;
        for (int64_t u = 0; u <= n; ++u) {
            while (mlo && lvl < t->level_off.size() && t->level_off[lvl] == u) mlo[lvl++] = (int32_t)multi;
            if (u == n) break;
            const int32_t lo = t->child_off[2 * u], hi = t->child_off[2 * u + 1];
            const int32_t nc = hi - lo;
            if (sum_idx) sum_idx[u] = nc == 1 ? lo : (nc >= 2 ? (int32_t)(n + 1 + multi) : (int32_t)n);
            if (nc >= 2) {
                if (ranges) { ranges[2 * multi] = lo; ranges[2 * multi + 1] = hi; }
                ++multi;
            }
        }
        if (wave) {
            // first chunk whose start is >= w * rows_per_wave (the chunk starts ascend); chunks without rows in front of the first
            // row belong to the first wave
            const int64_t nw = waves_of(h->N, rows_per_wave), nc = (int64_t)t->chunk_start.size();
            int64_t c = 0;
            for (int64_t w = 0; w < nw; ++w) {
                const int64_t target = w * (int64_t)rows_per_wave;
                while (c < nc && t->chunk_start[c] < target) ++c;
                wave[w] = (int32_t)(w == 0 ? 0 : c);
            }
            wave[nw] = (int32_t)nc;
        }
    }
    return k;
}

extern "C" void gtos_pathtrie_free(gtos_pathtrie* h) { delete h; }
