// TEST INFRASTRUCTURE (oracle/): the GPU relation-index builder's per-thread stages (gtos_amd/csrc/relindex_kernels.h) run as serial
// loops on the host, std::stable_sort and running sums standing in for rocPRIM.  tests/test_relindex_dev.py drives it through the
// same Python glue as the HIP library (gtos_amd/relindex_hip.py) and compares every array with csrc_host/relindex.cpp.
// Not part of the product: nothing under gtos_amd/ loads this file.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#include "../gtos_amd/csrc/relindex_kernels.h"
#include "emul_order.h"

using namespace gtos_relindex_dev;
using gtos_emul::for_each;
GTOS_EMUL_ORDER_ENTRY(gtos_relindex_emul)

namespace {
template <typename K, typename V>
void sort_pairs(const K* key, const V* val, K* key_out, V* val_out, int64_t n) {
    std::vector<int64_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return key[a] < key[b]; });
    for (int64_t i = 0; i < n; ++i) { key_out[i] = key[idx[i]]; val_out[i] = val[idx[i]]; }
}
void scan_v8(const V8* in, V8* out, int64_t n) {
    V8 run;
    for (int q = 0; q < 8; ++q) run.v[q] = 0;
    for (int64_t i = 0; i < n; ++i) { run = Add8()(run, in[i]); out[i] = run; }
}
void scan_u32(const uint32_t* in, uint32_t* out, int64_t n) {
    uint32_t run = 0;
    for (int64_t i = 0; i < n; ++i) { run += in[i]; out[i] = run; }
}
}  // namespace

extern "C" int gtos_relindex_emul_phase_a(const int64_t* geom, void** tab) {
    const Geom G = geom_of(geom);
    if (!geom_ok(G)) return -1;
    const int64_t* relation = (const int64_t*)tab[T_RELATION];
    int32_t *cnt = (int32_t*)tab[T_CNT], *sizes = (int32_t*)tab[T_SIZES];
    for_each(G.P, [&](int64_t e) { cell_key(e, G, relation, (uint32_t*)tab[T_KEY], (int32_t*)tab[T_VAL], sizes); });
    if (sizes[IZ_ERR]) return 0;
    sort_pairs((const uint32_t*)tab[T_KEY], (const int32_t*)tab[T_VAL], (uint32_t*)tab[T_SKEY], (int32_t*)tab[T_PAIR_SORTED], G.P);
    for_each(G.R, [&](int64_t t) { type_bounds(t, G, (const uint32_t*)tab[T_SKEY], cnt, (uint32_t*)tab[T_CUM_CNT]); });
    for_each(G.P, [&](int64_t p) { idx_cell(p, G, relation, cnt, (int32_t*)tab[T_IDX_Q], (int32_t*)tab[T_IDX_K]); });
    for_each(G.R, [&](int64_t t) { type_counts(t, G, cnt, (uint32_t*)tab[T_NCH], (uint32_t*)tab[T_HEAVY]); });
    scan_u32((const uint32_t*)tab[T_NCH], (uint32_t*)tab[T_CUM_NCH], G.R);
    scan_u32((const uint32_t*)tab[T_HEAVY], (uint32_t*)tab[T_CUM_HEAVY], G.R);
    const Chunks ch = chunks_of(tab);
    for_each(G.R, [&](int64_t t) {
        type_chunks(t, G, cnt, (const uint32_t*)tab[T_CUM_CNT], (const uint32_t*)tab[T_NCH], (const uint32_t*)tab[T_CUM_NCH], (const uint32_t*)tab[T_HEAVY],
                    (const uint32_t*)tab[T_CUM_HEAVY], (const int32_t*)tab[T_PAIR_SORTED], ch, (int32_t*)tab[T_HEAVY_TYPES]);
    });
    sizes_a(G, (const uint32_t*)tab[T_CUM_NCH], (const uint32_t*)tab[T_CUM_HEAVY], sizes);
    return 0;
}

extern "C" int gtos_relindex_emul_phase_b(const int64_t* geom, int64_t nchunks, void** tab) {
    const Geom G = geom_of(geom);
    if (!geom_ok(G) || nchunks <= 0) return -1;
    int32_t* sizes = (int32_t*)tab[T_SIZES];
    if (sizes[IZ_NCHUNKS] != nchunks) return -2;
    const Chunks ch = chunks_of(tab);
    V8 *v8 = (V8*)tab[T_V8], *v8_cum = (V8*)tab[T_V8_CUM];
    for_each(nchunks, [&](int64_t c) { roam_key(c, ch, (uint32_t*)tab[T_RKEY], (int32_t*)tab[T_RVAL], v8); });
    sort_pairs((const uint32_t*)tab[T_RKEY], (const int32_t*)tab[T_RVAL], (uint32_t*)tab[T_RKEY_S], (int32_t*)tab[T_ROAM_SORTED], nchunks);
    scan_v8(v8, v8_cum, nchunks);
    roam_setup(nchunks, (const uint32_t*)tab[T_RKEY_S], v8_cum, (unsigned long long*)tab[T_LOAD], sizes);
    for_each(nchunks, [&](int64_t q) { roam_cost(q, (const int32_t*)tab[T_ROAM_SORTED], ch, (int32_t*)tab[T_RCOST], sizes); });
    greedy_homes((const int32_t*)tab[T_RCOST], (int32_t*)tab[T_HOME_Q], (unsigned long long*)tab[T_LOAD], sizes);
    for_each(nchunks, [&](int64_t q) { scatter_homes(q, (const int32_t*)tab[T_ROAM_SORTED], (const int32_t*)tab[T_HOME_Q], ch, sizes); });
    for_each(nchunks, [&](int64_t c) { final_key(c, ch, (uint64_t*)tab[T_FKEY], (int32_t*)tab[T_FVAL], v8); });
    sort_pairs((const uint64_t*)tab[T_FKEY], (const int32_t*)tab[T_FVAL], (uint64_t*)tab[T_FKEY_S], (int32_t*)tab[T_PERM], nchunks);
    scan_v8(v8, v8_cum, nchunks);
    for_each(nchunks, [&](int64_t q) {
        gather_chunk(q, (const int32_t*)tab[T_PERM], ch, (int32_t*)tab[T_CHUNK_TYPE], (int32_t*)tab[T_CHUNK_START], (int32_t*)tab[T_CHUNK_COUNT],
                     (int32_t*)tab[T_CHUNK_SLOT]);
    });
    xcd_offsets(nchunks, v8_cum, (int32_t*)tab[T_XCD_OFF]);
    return 0;
}
