// TEST INFRASTRUCTURE (oracle/): the GPU path-trie builder's per-thread stages (gtos_amd/csrc/trie_kernels.h) run as serial loops
// on the host, std::stable_sort and a running sum standing in for rocPRIM.  tests/test_pathtrie.py drives it through the same
// Python glue as the HIP library (gtos_amd/pathtrie_hip.py) and compares the tries with csrc_host/pathtrie.cpp array for array.
// Not part of the product: nothing under gtos_amd/ loads this file.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#include "../gtos_amd/csrc/trie_kernels.h"
#include "emul_order.h"

using namespace gtos_trie;
using gtos_emul::for_each;
GTOS_EMUL_ORDER_ENTRY(gtos_trie_emul)

namespace {
void scan_v8(const V8* in, V8* out, int64_t n) {
    V8 run;
    for (int q = 0; q < 8; ++q) run.v[q] = 0;
    for (int64_t i = 0; i < n; ++i) { run = Add8()(run, in[i]); out[i] = run; }
}
void sort_pairs_u64(uint64_t* key, int32_t* val, int64_t n) {
    std::vector<int64_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return key[a] < key[b]; });
    std::vector<uint64_t> k2(n);
    std::vector<int32_t> v2(n);
    for (int64_t i = 0; i < n; ++i) { k2[i] = key[idx[i]]; v2[i] = val[idx[i]]; }
    std::memcpy(key, k2.data(), n * 8);
    std::memcpy(val, v2.data(), n * 4);
}
}  // namespace

extern "C" int gtos_trie_emul_phase_a(int L, int64_t R, const int64_t* bank, const int64_t* length, void** common, void** pf, void** sf,
                                      int32_t* sizes) {
    uint8_t* len8 = (uint8_t*)common[C_LEN8];
    V8* scratch = (V8*)common[C_SCRATCH];
    Side F = side_of(pf), B = side_of(sf);
    for (int q = 0; q < SZ_TOTAL; ++q) sizes[q] = 0;
    for_each(R, [&](int64_t s) { make_keys(s, L, R, bank, length, F.key, B.key, F.order, B.order, len8, &sizes[SZ_ERR]); });
    if (sizes[SZ_ERR]) return 0;
    sort_pairs_u64(F.key, F.order, R);
    sort_pairs_u64(B.key, B.order, R);
    int k = 0;
    for (Side* t : {&F, &B}) {
        for_each(R, [&](int64_t i) { open_flags(i, t->key, t->newmask, scratch); });
        scan_v8(scratch, t->cum, R);
        level_offsets(t->cum, R, t->lvl, sizes + (k++ ? SZ_SF : SZ_PF));
    }
    V8* cumlen = (V8*)common[C_CUMLEN];
    for_each(R, [&](int64_t i) { length_onehot(i, F.order, len8, scratch); });
    scan_v8(scratch, cumlen, R);
    packed_geometry(cumlen, R, (int32_t*)common[C_START], (int32_t*)common[C_BATCH], (int64_t*)common[C_OFFS]);
    for_each(R, [&](int64_t i) {
        packed_position(i, F.order, len8, cumlen, (const int32_t*)common[C_START], (int32_t*)common[C_SEQ_ORDER],
                        (int32_t*)common[C_SEQ_POS], (int64_t*)common[C_SEQ_ORDER64], (int64_t*)common[C_SEQ_POS64], (int32_t*)common[C_LEXF]);
    });
    for_each(R, [&](int64_t i) { lex_position(i, B.order, (int32_t*)common[C_LEXB]); });
    return 0;
}

extern "C" int gtos_trie_emul_phase_b(int64_t R, int64_t N, int n_pf, int n_sf, int chunk, int rows_per_wave, void** common, void** pf,
                                      void** sf, int32_t* sizes) {
    const uint8_t* len8 = (const uint8_t*)common[C_LEN8];
    Side F = side_of(pf), B = side_of(sf);
    if (F.lvl[8] != n_pf || B.lvl[8] != n_sf) return -2;              // the counts the host read after phase A
    F.row_node = (int32_t*)common[C_ROW_PF];
    B.row_node = (int32_t*)common[C_ROW_SF];
    for (Side* t : {&F, &B}) {
        for_each(R, [&](int64_t i) { write_nodes(i, *t); });
        const int64_t n = t->lvl[8];
        for_each(n, [&](int64_t v) { children(v, *t); });
    }
    for_each(R, [&](int64_t m) {
        fill_rows(m, (const int32_t*)common[C_SEQ_ORDER], len8, (const int32_t*)common[C_LEXF], (const int32_t*)common[C_LEXB],
                  (const int64_t*)common[C_OFFS], F.node_tab, B.node_tab, F.row_node, B.row_node);
    });
    const int64_t n_waves = std::max<int64_t>(1, (N + rows_per_wave - 1) / rows_per_wave);
    int k = 0;
    for (Side* t : {&F, &B}) {
        int32_t* sz = sizes + (k++ ? SZ_SF : SZ_PF);
        const int64_t n = t->lvl[8];
        // rows sorted by node, stable in the row id
        std::vector<int64_t> idx(N);
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return t->row_node[a] < t->row_node[b]; });
        for (int64_t e = 0; e < N; ++e) { t->row_key[e] = (uint32_t)t->row_node[idx[e]]; t->rows[e] = (int32_t)idx[e]; }
        for_each(N, [&](int64_t e) { node_offsets(e, N, t->row_key, t->off, (int32_t)n); });
        for_each(n, [&](int64_t u) { node_counts(u, chunk, *t); });
        scan_v8(t->aux, t->aux_cum, n);
        for_each(n, [&](int64_t u) { node_records(u, chunk, *t); });
        side_sizes(*t, sz);
        for_each((n_waves) + 1, [&](int64_t w) { wave_range(w, n_waves, rows_per_wave, *t, sz); });
    }
    return 0;
}
