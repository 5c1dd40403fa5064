// TEST INFRASTRUCTURE (oracle/): the GPU relation-batch builder's per-thread stages (gtos_amd/csrc/relbatch_kernels.h) run as serial
// loops on the host, std::stable_sort and a running sum standing in for rocPRIM.  tests/test_relbatch_dev.py drives it through the
// same Python glue as the HIP library (gtos_amd/relbatch_hip.py) and compares relation / bank / length with csrc_host/relbatch.cpp.
// Not part of the product: nothing under gtos_amd/ loads this file.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#include "../gtos_amd/csrc/relbatch_kernels.h"
#include "emul_order.h"

using namespace gtos_relbatch_dev;
using gtos_emul::for_each;
GTOS_EMUL_ORDER_ENTRY(gtos_relbatch_emul)

namespace {
template <typename K, typename V>
void sort_pairs(const K* key, const V* val, K* key_out, V* val_out, int64_t n) {
    std::vector<int64_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return key[a] < key[b]; });
    for (int64_t i = 0; i < n; ++i) { key_out[i] = key[idx[i]]; val_out[i] = val[idx[i]]; }
}
}  // namespace

extern "C" int gtos_relbatch_emul_phase_a(const int64_t* geom, void** tab) {
    const Geom G = geom_of(geom);
    if (!geom_ok(G)) return -1;
    const Graphs gr = graphs_of(tab);
    const Scratch sc = scratch_of(tab);
    uint64_t *key = (uint64_t*)tab[T_KEY], *skey = (uint64_t*)tab[T_SKEY], *seg_key = (uint64_t*)tab[T_SEG_KEY];
    int32_t *posn = (int32_t*)tab[T_POSN], *spos = (int32_t*)tab[T_SPOS], *seg_id = (int32_t*)tab[T_SEG_ID];
    uint64_t *flag = (uint64_t*)tab[T_FLAG], *cum = (uint64_t*)tab[T_CUM];
    uint32_t* first_pos = (uint32_t*)tab[T_FIRST_POS];
    int32_t *len_seen = (int32_t*)tab[T_LEN_SEEN], *sizes = (int32_t*)tab[T_SIZES];
    const int64_t total = G.P + N_SPECIAL;
    std::memset(len_seen, 0, 8 * sizeof(int32_t));
    std::memset(sizes, 0, RZ_TOTAL * sizeof(int32_t));
    special_keys(G, key, posn, len_seen);
    for_each(G.S, [&](int64_t s) { bfs_source((int32_t)s, G, gr, sc); });
    for_each(G.P, [&](int64_t p) { pair_key(p, G, gr, sc, key, posn, len_seen); });
    sort_pairs(key, posn, skey, spos, total);
    for_each(total, [&](int64_t e) { head_flag(e, skey, flag); });
    uint64_t run = 0;
    for (int64_t e = 0; e < total; ++e) { run += flag[e]; cum[e] = run; }
    for_each(total, [&](int64_t e) { segment_first(e, skey, spos, cum, first_pos, seg_id, seg_key); });
    sizes_after_scan(cum, total, len_seen, sizes);
    return 0;
}

extern "C" int gtos_relbatch_emul_phase_b(const int64_t* geom, int64_t R, void** tab) {
    const Geom G = geom_of(geom);
    if (!geom_ok(G) || R < N_SPECIAL) return -1;
    const Graphs gr = graphs_of(tab);
    const int64_t total = G.P + N_SPECIAL;
    if (((const int32_t*)tab[T_SIZES])[RZ_R] != R) return -2;             // the count the host read after phase A
    sort_pairs((const uint32_t*)tab[T_FIRST_POS], (const int32_t*)tab[T_SEG_ID], (uint32_t*)tab[T_FIRST_ALT], (int32_t*)tab[T_SORTED_SEG], R);
    for_each(R, [&](int64_t r) {
        type_of_segment(r, (const int32_t*)tab[T_SORTED_SEG], (const uint64_t*)tab[T_SEG_KEY], (int32_t*)tab[T_TYPE_OF_SEG], R, (int64_t*)tab[T_BANK],
                        (int64_t*)tab[T_LENGTH]);
    });
    for_each(total, [&](int64_t e) {
        scatter_relation(e, G, gr, (const int32_t*)tab[T_SPOS], (const uint64_t*)tab[T_CUM], (const int32_t*)tab[T_TYPE_OF_SEG], (int64_t*)tab[T_RELATION]);
    });
    for_each(G.S, [&](int64_t s) { cls_cells((int32_t)s, G, gr, (int64_t*)tab[T_RELATION]); });
    return 0;
}

// ---- MODE_ALL (every shortest path of every pair): counting phase, key phase, fill phase
extern "C" int gtos_relbatch_emul_all_count(const int64_t* geom, void** tab) {
    const Geom G = geom_of(geom);
    if (!geom_ok(G) || G.mode != MODE_ALL) return -1;
    const Graphs gr = graphs_of(tab);
    const Scratch sc = scratch_of(tab);
    uint32_t *nalt = (uint32_t*)tab[T_NALT], *cmax = (uint32_t*)tab[T_CMAX_ALT];
    uint64_t* cum = (uint64_t*)tab[T_CUM_ALT];
    for_each(G.S, [&](int64_t s) { bfs_source((int32_t)s, G, gr, sc); });
    for_each(G.P, [&](int64_t p) { pair_alt_count(p, G, gr, sc, nalt, (uint64_t*)tab[T_NALT64]); });
    uint64_t run = 0;
    uint32_t mx = 0;
    for (int64_t p = 0; p < G.P; ++p) { run += nalt[p]; cum[p] = run; mx = std::max(mx, nalt[p]); cmax[p] = mx; }
    sizes_count(cum, cmax, G.P, (int32_t*)tab[T_SIZES]);
    return 0;
}

extern "C" int gtos_relbatch_emul_all_keys(const int64_t* geom, void** tab) {
    const Geom G = geom_of(geom);
    if (!geom_ok(G) || G.mode != MODE_ALL || G.T < G.P || G.K < 1) return -1;
    const Graphs gr = graphs_of(tab);
    const Scratch sc = scratch_of(tab);
    uint64_t *key = (uint64_t*)tab[T_KEY], *skey = (uint64_t*)tab[T_SKEY], *seg_key = (uint64_t*)tab[T_SEG_KEY];
    int32_t *posn = (int32_t*)tab[T_POSN], *spos = (int32_t*)tab[T_SPOS], *seg_id = (int32_t*)tab[T_SEG_ID];
    uint64_t *flag = (uint64_t*)tab[T_FLAG], *cum = (uint64_t*)tab[T_CUM];
    uint32_t* first_pos = (uint32_t*)tab[T_FIRST_POS];
    int32_t *len_seen = (int32_t*)tab[T_LEN_SEEN], *sizes = (int32_t*)tab[T_SIZES];
    const int64_t total = G.T + N_SPECIAL_ALL;
    if (sizes[RZ_T] != G.T) return -2;                                     // the total the host read after the counting phase
    std::memset(len_seen, 0, 8 * sizeof(int32_t));
    special_keys_all(G, key, posn, len_seen);
    for_each(G.P, [&](int64_t p) { pair_alt_keys(p, G, gr, sc, (const uint64_t*)tab[T_CUM_ALT], (const uint32_t*)tab[T_NALT], key, posn, len_seen); });
    sort_pairs(key, posn, skey, spos, total);
    for_each(total, [&](int64_t e) { head_flag(e, skey, flag); });
    uint64_t run = 0;
    for (int64_t e = 0; e < total; ++e) { run += flag[e]; cum[e] = run; }
    for_each(total, [&](int64_t e) { segment_first(e, skey, spos, cum, first_pos, seg_id, seg_key); });
    sizes_after_scan(cum, total, len_seen, sizes);
    return 0;
}

extern "C" int gtos_relbatch_emul_all_fill(const int64_t* geom, int64_t R, void** tab) {
    const Geom G = geom_of(geom);
    if (!geom_ok(G) || G.mode != MODE_ALL || R < N_SPECIAL_ALL) return -1;
    const Graphs gr = graphs_of(tab);
    const int64_t total = G.T + N_SPECIAL_ALL;
    if (((const int32_t*)tab[T_SIZES])[RZ_R] != R) return -2;
    sort_pairs((const uint32_t*)tab[T_FIRST_POS], (const int32_t*)tab[T_SEG_ID], (uint32_t*)tab[T_FIRST_ALT], (int32_t*)tab[T_SORTED_SEG], R);
    for_each(R, [&](int64_t r) {
        type_of_segment(r, (const int32_t*)tab[T_SORTED_SEG], (const uint64_t*)tab[T_SEG_KEY], (int32_t*)tab[T_TYPE_OF_SEG], R, (int64_t*)tab[T_BANK],
                        (int64_t*)tab[T_LENGTH]);
    });
    for_each(total, [&](int64_t e) {
        scatter_relation_all(e, G, gr, (const int32_t*)tab[T_SPOS], (const uint64_t*)tab[T_CUM], (const int32_t*)tab[T_TYPE_OF_SEG],
                             (const uint64_t*)tab[T_CUM_ALT], (int64_t*)tab[T_RELATION]);
    });
    for_each(G.S, [&](int64_t s) { cls_cells_all((int32_t)s, G, gr, (int64_t*)tab[T_RELATION]); });
    return 0;
}
