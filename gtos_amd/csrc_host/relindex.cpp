// libgtos_host.so, third part: index preparation of the FACTORED relation operand (include/gtos_host.h).
//
// generator/generator.py:79 expands relation[n,n,B] (type ids) into a dense [n,n,B,d] tensor; the MI355X attention kernels
// read the type ids instead (gtos_amd/ops.py FactoredRelation) and need, once per batch:
//   * the ids in query-major and key-major order as int32 (the kernels stream them beside q / k),
//   * for the bank gradient (the index_add of autograd's index_select backward): the pairs grouped by type, cut into chunks
//     of <= `chunk` pairs; types with several chunks ("heavy": <CLS>, <rCLS>, <SELF>, <TL>) get an fp32 accumulation slot;
//     types with ONE pair (86 % of the types of a synthetic AMR batch) get no chunk at all: their ids carry bit 31 and
//     the query-major attention backward writes their gradient row directly,
//   * the chunks ordered by (XCD that owns the graph of the chunk's first pair, graph, key row) so that consecutive
//     workgroups of the gradient kernel gather q / k rows of one graph from that XCD's L2.
// All of it is integer work on the batch's relation tensor, so it belongs to batch assembly on the host, next to the bank.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../include/gtos_host.h"

struct gtos_relindex {
    int n = 0, B = 0;
    int64_t R = 0, P = 0;
    std::vector<int32_t> idx_q, idx_k, pair_sorted, chunk_type, chunk_start, chunk_count, chunk_slot, xcd_off, heavy_types;
};

extern "C" gtos_relindex* gtos_relindex_build(int n, int B, int64_t R, const int64_t* relation, int chunk) {
    if (n <= 0 || B <= 0 || R <= 0 || !relation || chunk <= 0) return nullptr;
    const int64_t P = (int64_t)n * n * B;
    if (P > 0x7fffffffLL || R > 0x7fffffffLL) return nullptr;
    auto* h = new gtos_relindex();
    h->n = n; h->B = B; h->R = R; h->P = P;
    h->idx_q.resize(P);
    h->idx_k.resize(P);
    std::vector<int64_t> count(R + 1, 0);
    // relation[j][i][b] at flat index (j*n + i)*B + b
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = 0; i < n; ++i)
            for (int64_t b = 0; b < B; ++b) {
                const int64_t t = relation[(j * n + i) * B + b];
                if (t < 0 || t >= R) { delete h; return nullptr; }
                h->idx_q[(i * B + b) * n + j] = (int32_t)t;       // [i,b,j]
                h->idx_k[(j * B + b) * n + i] = (int32_t)t;       // [j,b,i]
                count[t + 1]++;
            }
    // a type that occurs exactly once: its bank-gradient row is its single pair's term, which the query-major attention
    // backward writes itself -- flagged with bit 31 of the id, and left out of the type-major chunk list below
    for (int64_t p = 0; p < P; ++p) {
        const int32_t tq = h->idx_q[p], tk = h->idx_k[p];
        if (count[tq + 1] == 1) h->idx_q[p] = tq | (int32_t)0x80000000;
        if (count[tk + 1] == 1) h->idx_k[p] = tk | (int32_t)0x80000000;
    }
    for (int64_t t = 0; t < R; ++t) count[t + 1] += count[t];
    // pairs grouped by type; inside a type graph-major (b, j, i): the pairs of one graph are neighbours
    h->pair_sorted.resize(P);
    {
        std::vector<int64_t> cur(count.begin(), count.end() - 1);
        for (int64_t b = 0; b < B; ++b)
            for (int64_t j = 0; j < n; ++j)
                for (int64_t i = 0; i < n; ++i) {
                    const int64_t p = (j * n + i) * B + b;
                    h->pair_sorted[cur[relation[p]]++] = (int32_t)p;
                }
    }
    struct Chunk { int32_t type, start, cnt, slot; int64_t key; int32_t x_first, x_last; };
    std::vector<Chunk> chunks;
    chunks.reserve(R + P / chunk);
    auto xcd_of = [&](int64_t pair) {                                        // the attention kernels' graph -> XCD map
        const int64_t gb = pair % B;
        return (int32_t)((B % 8 == 0) ? gb / (B / 8) : gb % 8);
    };
    for (int64_t t = 0; t < R; ++t) {
        const int64_t lo = count[t], hi = count[t + 1];
        if (hi - lo == 1) continue;                                          // singleton: written by the attention backward
        // light types: one chunk of <= `chunk` pairs; heavy types (more than `chunk` pairs): chunks of 4 * chunk pairs -- fewer
        // fp32-atomic flushes into the shared slot (65 k pairs of <TL>: 512, not 2048) without turning one wave's serial walk
        // over its chunk into the kernel's critical path (16 * chunk measured 390 us per launch for 0.44 GB of traffic)
        constexpr int64_t mult = 4;                                          // (round 3: 4 -> 800, 2 -> 890, 1 -> 1,120 us for the attention backward)
        const int64_t csz = hi - lo > chunk ? mult * (int64_t)chunk : chunk;
        const int64_t nch = hi > lo ? (hi - lo + csz - 1) / csz : 1;          // a type without pairs still writes its zero row
        int32_t slot = -1;
        if (hi - lo > chunk) {
            slot = (int32_t)h->heavy_types.size();
            h->heavy_types.push_back((int32_t)t);
        }
        for (int64_t c = 0; c < nch; ++c) {
            const int64_t s = lo + c * csz;
            const int64_t cnt = std::max<int64_t>(0, std::min<int64_t>(csz, hi - s));
            const int64_t first = h->pair_sorted[std::min<int64_t>(s, P - 1)];
            const int64_t lastp = h->pair_sorted[std::min<int64_t>(s + std::max<int64_t>(cnt, 1) - 1, P - 1)];
            const int64_t gb = first % B, j = first / ((int64_t)n * B);
            chunks.push_back({(int32_t)t, (int32_t)s, (int32_t)cnt, slot, (gb << 20) | j, xcd_of(first), xcd_of(lastp)});
        }
    }
    // Which XCD walks a chunk, and when.  The bank-gradient kernel gives every wave a chain of chunks and a chunk costs about
    // ceil(cnt / 4) + 1 dependent load rounds, so the launch lasts as long as the longest chain on the busiest XCD.  Round 2
    // sent every chunk to the XCD of its FIRST pair's graph; inside a type the pairs are graph-major, so a type shared by several
    // graphs always went to the lowest one's XCD: XCD 0 got 2.6x the chunks of XCD 7 (C2: 11,638 vs 4,517) and 3x the rounds,
    // and a wave could meet two 128-pair chunks (chains of 122 rounds where the mean is 31).  Now: a chunk whose pairs all live
    // on one XCD stays there (its q/k rows are in that XCD's L2); a chunk that spans XCDs has no home and goes to the XCD with the
    // fewest rounds so far, the longest first; inside an XCD the long chunks (> 8 pairs) come first, longest first, so each
    // starts a different wave's chain, and the short ones follow in (graph, key row) order as before (268-288 -> 202-206 us per launch).
    auto cost = [](const Chunk& c) { return (int64_t)(c.cnt + 3) / 4 + 1; };
    std::vector<int32_t> home(chunks.size());
    {
        int64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        std::vector<int32_t> roam;
        for (size_t c = 0; c < chunks.size(); ++c) {
            if (chunks[c].x_first == chunks[c].x_last) { home[c] = chunks[c].x_first; load[home[c]] += cost(chunks[c]); }
            else roam.push_back((int32_t)c);
        }
        std::stable_sort(roam.begin(), roam.end(), [&](int32_t a, int32_t b) { return chunks[a].cnt > chunks[b].cnt; });
        for (int32_t c : roam) {
            int best = 0;
            for (int x = 1; x < 8; ++x) if (load[x] < load[best]) best = x;
            home[c] = best; load[best] += cost(chunks[c]);
        }
        for (size_t c = 0; c < chunks.size(); ++c) {
            const int64_t lng = chunks[c].cnt > 8 ? 0 : 1;                     // long chunks first, longest first
            const int64_t rank = lng ? chunks[c].key : (int64_t)(0xfffff - std::min<int32_t>(chunks[c].cnt, 0xfffff)) << 20;
            chunks[c].key = ((int64_t)home[c] << 42) | (lng << 41) | rank;
        }
    }
    std::stable_sort(chunks.begin(), chunks.end(), [](const Chunk& a, const Chunk& b) { return a.key < b.key; });
    const size_t nc = chunks.size();
    h->chunk_type.resize(nc); h->chunk_start.resize(nc); h->chunk_count.resize(nc); h->chunk_slot.resize(nc);
    h->xcd_off.assign(9, 0);
    for (size_t c = 0; c < nc; ++c) {
        h->chunk_type[c] = chunks[c].type; h->chunk_start[c] = chunks[c].start;
        h->chunk_count[c] = chunks[c].cnt; h->chunk_slot[c] = chunks[c].slot;
        h->xcd_off[(chunks[c].key >> 42) + 1]++;
    }
    for (int x = 0; x < 8; ++x) h->xcd_off[x + 1] += h->xcd_off[x];
    return h;
}

extern "C" int gtos_relindex_sizes(const gtos_relindex* h, int64_t* sizes) {
    if (!h || !sizes) return -1;
    sizes[0] = h->P; sizes[1] = (int64_t)h->chunk_type.size(); sizes[2] = (int64_t)h->heavy_types.size();
    return 0;
}

extern "C" int gtos_relindex_export(const gtos_relindex* h, int32_t** out) {
    if (!h || !out) return -1;
    const std::vector<int32_t>* v[] = {&h->idx_q, &h->idx_k, &h->pair_sorted, &h->chunk_type, &h->chunk_start, &h->chunk_count,
                                       &h->chunk_slot, &h->xcd_off, &h->heavy_types};
    int k = 0;
    for (auto* a : v) {
        if (out[k] && !a->empty()) std::memcpy(out[k], a->data(), a->size() * sizeof(int32_t));
        ++k;
    }
    return k;
}

extern "C" void gtos_relindex_free(gtos_relindex* h) { delete h; }
