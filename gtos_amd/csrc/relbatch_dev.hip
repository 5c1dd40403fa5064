// The relation tensors of a batch built on the GPU: the launch glue around the per-thread stages of relbatch_kernels.h (whose logic
// tests/test_relbatch_dev.py proves equal to csrc_host/relbatch.cpp on the host, running the same code as serial loops) plus rocPRIM's
// radix sort and scan.  Two phases with one host read between them (gtos_amd/relbatch_hip.py): phase A is the all-pairs work (a BFS
// per (graph, source), a key per pair), the key sort and the scan that counts the distinct keys; the host then allocates the bank
// and phase B numbers the types in first-seen order and writes relation / bank / length.
//
// Compiled WITHOUT -ffast-math (gtos_amd/build.py): the uniform choice among alternative shortest paths compares running sums of
// doubles and must take the same branch as the host builder.
//
// STATUS (end of round 3): compiles for gfx950; written after the round's GPU minutes were spent, so it has not run yet.  Opt-in.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include "relbatch_kernels.h"

using namespace gtos_relbatch_dev;

namespace {

#define GTOS_RB_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return 100 + (int)e_; } while (0)
#define GTOS_RB_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return 100 + (int)e_; } while (0)

inline dim3 grid_for(int64_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

__global__ void k_special(Geom G, uint64_t* key, int32_t* posn, int32_t* len_seen) {
    if (blockIdx.x == 0 && threadIdx.x == 0) special_keys(G, key, posn, len_seen);
}
// one thread per (graph, source): 64-thread blocks so that a batch's few thousand searches spread over the CUs
__global__ void k_bfs(Geom G, Graphs gr, Scratch sc) {
    const int32_t s = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (s < G.S) bfs_source(s, G, gr, sc);
}
__global__ void k_pair_key(Geom G, Graphs gr, Scratch sc, uint64_t* key, int32_t* posn, int32_t* len_seen) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < G.P) pair_key(p, G, gr, sc, key, posn, len_seen);
}
__global__ void k_head_flag(int64_t total, const uint64_t* key, uint64_t* flag) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < total) head_flag(e, key, flag);
}
__global__ void k_segment_first(int64_t total, const uint64_t* key, const int32_t* posn, const uint64_t* cum, uint32_t* first_pos, int32_t* seg_id,
                                uint64_t* seg_key) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < total) segment_first(e, key, posn, cum, first_pos, seg_id, seg_key);
}
__global__ void k_sizes(const uint64_t* cum, int64_t total, const int32_t* len_seen, int32_t* sizes) {
    if (blockIdx.x == 0 && threadIdx.x == 0) sizes_after_scan(cum, total, len_seen, sizes);
}
__global__ void k_type_of_segment(int64_t R, const int32_t* sorted_seg, const uint64_t* seg_key, int32_t* type_of_seg, int64_t* bank, int64_t* length) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) type_of_segment(r, sorted_seg, seg_key, type_of_seg, R, bank, length);
}
__global__ void k_scatter_relation(int64_t total, Geom G, Graphs gr, const int32_t* posn, const uint64_t* cum, const int32_t* type_of_seg,
                                   int64_t* relation) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < total) scatter_relation(e, G, gr, posn, cum, type_of_seg, relation);
}
__global__ void k_cls_cells(Geom G, Graphs gr, int64_t* relation) {
    const int32_t s = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (s < G.S) cls_cells(s, G, gr, relation);
}

__global__ void k_pair_alt_count(Geom G, Graphs gr, Scratch sc, uint32_t* nalt, uint64_t* nalt64) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < G.P) pair_alt_count(p, G, gr, sc, nalt, nalt64);
}
__global__ void k_sizes_count(const uint64_t* cum, const uint32_t* cmax, int64_t P, int32_t* sizes) {
    if (blockIdx.x == 0 && threadIdx.x == 0) sizes_count(cum, cmax, P, sizes);
}
__global__ void k_special_all(Geom G, uint64_t* key, int32_t* posn, int32_t* len_seen) {
    if (blockIdx.x == 0 && threadIdx.x == 0) special_keys_all(G, key, posn, len_seen);
}
__global__ void k_pair_alt_keys(Geom G, Graphs gr, Scratch sc, const uint64_t* cum_alt, const uint32_t* nalt, uint64_t* key, int32_t* posn,
                                int32_t* len_seen) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < G.P) pair_alt_keys(p, G, gr, sc, cum_alt, nalt, key, posn, len_seen);
}
__global__ void k_scatter_relation_all(int64_t total, Geom G, Graphs gr, const int32_t* posn, const uint64_t* cum_flag, const int32_t* type_of_seg,
                                       const uint64_t* cum_alt, int64_t* relation) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < total) scatter_relation_all(e, G, gr, posn, cum_flag, type_of_seg, cum_alt, relation);
}
__global__ void k_cls_cells_all(Geom G, Graphs gr, int64_t* relation) {
    const int32_t s = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (s < G.S) cls_cells_all(s, G, gr, relation);
}

int bits_for(int64_t n) {                 // radix-sort key bits that can be set in values below n
    int b = 1;
    while ((1ll << b) < n && b < 32) ++b;
    return b;
}

}  // namespace

// bytes_out[0] = rocPRIM temporary storage the two phases need for `total` = pairs + 3 elements.
extern "C" int gtos_relbatch_dev_workspace(int64_t total, int64_t* bytes_out) {
    if (total <= 0 || !bytes_out) return -1;
    size_t a = 0, b = 0, c = 0;
    (void)rocprim::radix_sort_pairs(nullptr, a, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr,
                              (size_t)total, 0, 64, (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, b, (const uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)total, rocprim::plus<uint64_t>(), (hipStream_t)0);
    (void)rocprim::radix_sort_pairs(nullptr, c, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr,
                              (size_t)total, 0, 32, (hipStream_t)0);
    size_t m = a > b ? a : b;
    m = m > c ? m : c;
    bytes_out[0] = (int64_t)(m + 256);
    return 0;
}

// Phase A.  geom: int64[GE_COUNT] host integers; tab: host table of device pointers (enum T_* of relbatch_kernels.h); relation is
// zero-filled by the caller.  Afterwards sizes[RZ_R] = distinct paths (bank columns), sizes[RZ_L] = the longest of them.
extern "C" int gtos_relbatch_dev_phase_a(const int64_t* geom, void** tab, void* workspace, size_t workspace_bytes, void* stream) {
    if (!geom || !tab || !workspace) return -1;
    const Geom G = geom_of(geom);
    if (!geom_ok(G) || G.mode == MODE_ALL) return -1;
    for (int k = 0; k <= T_SEG_KEY; ++k) if (!tab[k]) return -1;
    if (!tab[T_LEN_SEEN] || !tab[T_SIZES]) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Graphs gr = graphs_of(tab);
    const Scratch sc = scratch_of(tab);
    uint64_t *key = (uint64_t*)tab[T_KEY], *skey = (uint64_t*)tab[T_SKEY], *seg_key = (uint64_t*)tab[T_SEG_KEY];
    int32_t *posn = (int32_t*)tab[T_POSN], *spos = (int32_t*)tab[T_SPOS], *seg_id = (int32_t*)tab[T_SEG_ID];
    uint64_t *flag = (uint64_t*)tab[T_FLAG], *cum = (uint64_t*)tab[T_CUM];
    uint32_t* first_pos = (uint32_t*)tab[T_FIRST_POS];
    int32_t *len_seen = (int32_t*)tab[T_LEN_SEEN], *sizes = (int32_t*)tab[T_SIZES];
    const int64_t total = G.P + N_SPECIAL;
    GTOS_RB_HIP(hipMemsetAsync(len_seen, 0, 8 * sizeof(int32_t), s));
    GTOS_RB_HIP(hipMemsetAsync(sizes, 0, RZ_TOTAL * sizeof(int32_t), s));
    hipLaunchKernelGGL(k_special, dim3(1), dim3(64), 0, s, G, key, posn, len_seen);
    GTOS_RB_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bfs, grid_for(G.S, 64), dim3(64), 0, s, G, gr, sc);
    GTOS_RB_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pair_key, grid_for(G.P, 256), dim3(256), 0, s, G, gr, sc, key, posn, len_seen);
    GTOS_RB_LAUNCH_CHECK();
    size_t bytes = workspace_bytes;
    GTOS_RB_HIP(rocprim::radix_sort_pairs(workspace, bytes, (const uint64_t*)key, skey, (const int32_t*)posn, spos, (size_t)total, 0, 64, s));
    hipLaunchKernelGGL(k_head_flag, grid_for(total, 256), dim3(256), 0, s, total, (const uint64_t*)skey, flag);
    GTOS_RB_LAUNCH_CHECK();
    bytes = workspace_bytes;
    GTOS_RB_HIP(rocprim::inclusive_scan(workspace, bytes, (const uint64_t*)flag, cum, (size_t)total, rocprim::plus<uint64_t>(), s));
    hipLaunchKernelGGL(k_segment_first, grid_for(total, 256), dim3(256), 0, s, total, (const uint64_t*)skey, (const int32_t*)spos, (const uint64_t*)cum,
                       first_pos, seg_id, seg_key);
    GTOS_RB_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sizes, dim3(1), dim3(64), 0, s, (const uint64_t*)cum, total, (const int32_t*)len_seen, sizes);
    GTOS_RB_LAUNCH_CHECK();
    return 0;
}

// Phase B: R = sizes[RZ_R] as the host read it; bank int64 [8, R] zero-filled, length int64 [R], first_alt / sorted_seg / type_of_seg [R].
extern "C" int gtos_relbatch_dev_phase_b(const int64_t* geom, int64_t R, void** tab, void* workspace, size_t workspace_bytes, void* stream) {
    if (!geom || !tab || !workspace || R < N_SPECIAL) return -1;
    const Geom G = geom_of(geom);
    if (!geom_ok(G) || G.mode == MODE_ALL) return -1;
    for (int k = 0; k <= T_LENGTH; ++k) if (!tab[k]) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Graphs gr = graphs_of(tab);
    const int64_t total = G.P + N_SPECIAL;
    if (R > total) return -1;
    size_t bytes = workspace_bytes;
    GTOS_RB_HIP(rocprim::radix_sort_pairs(workspace, bytes, (const uint32_t*)tab[T_FIRST_POS], (uint32_t*)tab[T_FIRST_ALT], (const int32_t*)tab[T_SEG_ID],
                                          (int32_t*)tab[T_SORTED_SEG], (size_t)R, 0, bits_for(total), s));
    hipLaunchKernelGGL(k_type_of_segment, grid_for(R, 256), dim3(256), 0, s, R, (const int32_t*)tab[T_SORTED_SEG], (const uint64_t*)tab[T_SEG_KEY],
                       (int32_t*)tab[T_TYPE_OF_SEG], (int64_t*)tab[T_BANK], (int64_t*)tab[T_LENGTH]);
    GTOS_RB_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scatter_relation, grid_for(total, 256), dim3(256), 0, s, total, G, gr, (const int32_t*)tab[T_SPOS], (const uint64_t*)tab[T_CUM],
                       (const int32_t*)tab[T_TYPE_OF_SEG], (int64_t*)tab[T_RELATION]);
    GTOS_RB_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_cls_cells, grid_for(G.S, 256), dim3(256), 0, s, G, gr, (int64_t*)tab[T_RELATION]);
    GTOS_RB_LAUNCH_CHECK();
    return 0;
}

// ---- GTOS_PATH_ALL (every shortest path of every pair; relation int64 [n,n,B,K]): three phases with a host read after the first two.
// Counting phase: the searches, the number of shortest paths of every pair, their running sum and maximum.  Afterwards sizes[RZ_T] =
// paths in total (-1: more than 2^31), sizes[RZ_K] = most of one pair; the caller sizes the key arrays (T + 4) from them.
extern "C" int gtos_relbatch_dev_all_count(const int64_t* geom, void** tab, void* workspace, size_t workspace_bytes, void* stream) {
    if (!geom || !tab || !workspace) return -1;
    const Geom G = geom_of(geom);
    if (!geom_ok(G) || G.mode != MODE_ALL) return -1;
    for (int k = 0; k <= T_DLAB; ++k) if (!tab[k]) return -1;
    if (!tab[T_NALT] || !tab[T_CUM_ALT] || !tab[T_CMAX_ALT] || !tab[T_NALT64] || !tab[T_SIZES]) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Graphs gr = graphs_of(tab);
    const Scratch sc = scratch_of(tab);
    uint32_t *nalt = (uint32_t*)tab[T_NALT], *cmax = (uint32_t*)tab[T_CMAX_ALT];
    uint64_t *nalt64 = (uint64_t*)tab[T_NALT64], *cum = (uint64_t*)tab[T_CUM_ALT];
    hipLaunchKernelGGL(k_bfs, grid_for(G.S, 64), dim3(64), 0, s, G, gr, sc);
    GTOS_RB_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pair_alt_count, grid_for(G.P, 256), dim3(256), 0, s, G, gr, sc, nalt, nalt64);
    GTOS_RB_LAUNCH_CHECK();
    size_t bytes = workspace_bytes;
    GTOS_RB_HIP(rocprim::inclusive_scan(workspace, bytes, (const uint64_t*)nalt64, cum, (size_t)G.P, rocprim::plus<uint64_t>(), s));
    bytes = workspace_bytes;
    GTOS_RB_HIP(rocprim::inclusive_scan(workspace, bytes, (const uint32_t*)nalt, cmax, (size_t)G.P, rocprim::maximum<uint32_t>(), s));
    hipLaunchKernelGGL(k_sizes_count, dim3(1), dim3(64), 0, s, (const uint64_t*)cum, (const uint32_t*)cmax, G.P, (int32_t*)tab[T_SIZES]);
    GTOS_RB_LAUNCH_CHECK();
    return 0;
}

// Key phase (geom carries T and K as the host read them): the keys of every path, the key sort, the distinct keys.  Afterwards
// sizes[RZ_R / RZ_L / RZ_N] as after gtos_relbatch_dev_phase_a.
extern "C" int gtos_relbatch_dev_all_keys(const int64_t* geom, void** tab, void* workspace, size_t workspace_bytes, void* stream) {
    if (!geom || !tab || !workspace) return -1;
    const Geom G = geom_of(geom);
    if (!geom_ok(G) || G.mode != MODE_ALL || G.T < G.P || G.K < 1 || G.T + N_SPECIAL_ALL > 0x7fffffffLL) return -1;
    for (int k = 0; k <= T_SEG_KEY; ++k) if (!tab[k]) return -1;
    if (!tab[T_LEN_SEEN] || !tab[T_SIZES] || !tab[T_NALT] || !tab[T_CUM_ALT]) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Graphs gr = graphs_of(tab);
    const Scratch sc = scratch_of(tab);
    uint64_t *key = (uint64_t*)tab[T_KEY], *skey = (uint64_t*)tab[T_SKEY], *seg_key = (uint64_t*)tab[T_SEG_KEY];
    int32_t *posn = (int32_t*)tab[T_POSN], *spos = (int32_t*)tab[T_SPOS], *seg_id = (int32_t*)tab[T_SEG_ID];
    uint64_t *flag = (uint64_t*)tab[T_FLAG], *cum = (uint64_t*)tab[T_CUM];
    uint32_t* first_pos = (uint32_t*)tab[T_FIRST_POS];
    int32_t *len_seen = (int32_t*)tab[T_LEN_SEEN], *sizes = (int32_t*)tab[T_SIZES];
    const int64_t total = G.T + N_SPECIAL_ALL;
    GTOS_RB_HIP(hipMemsetAsync(len_seen, 0, 8 * sizeof(int32_t), s));
    hipLaunchKernelGGL(k_special_all, dim3(1), dim3(64), 0, s, G, key, posn, len_seen);
    GTOS_RB_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pair_alt_keys, grid_for(G.P, 256), dim3(256), 0, s, G, gr, sc, (const uint64_t*)tab[T_CUM_ALT], (const uint32_t*)tab[T_NALT], key,
                       posn, len_seen);
    GTOS_RB_LAUNCH_CHECK();
    size_t bytes = workspace_bytes;
    GTOS_RB_HIP(rocprim::radix_sort_pairs(workspace, bytes, (const uint64_t*)key, skey, (const int32_t*)posn, spos, (size_t)total, 0, 64, s));
    hipLaunchKernelGGL(k_head_flag, grid_for(total, 256), dim3(256), 0, s, total, (const uint64_t*)skey, flag);
    GTOS_RB_LAUNCH_CHECK();
    bytes = workspace_bytes;
    GTOS_RB_HIP(rocprim::inclusive_scan(workspace, bytes, (const uint64_t*)flag, cum, (size_t)total, rocprim::plus<uint64_t>(), s));
    hipLaunchKernelGGL(k_segment_first, grid_for(total, 256), dim3(256), 0, s, total, (const uint64_t*)skey, (const int32_t*)spos, (const uint64_t*)cum,
                       first_pos, seg_id, seg_key);
    GTOS_RB_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sizes, dim3(1), dim3(64), 0, s, (const uint64_t*)cum, total, (const int32_t*)len_seen, sizes);
    GTOS_RB_LAUNCH_CHECK();
    return 0;
}

// Fill phase: R as the host read it; relation int64 [n,n,B,K] and bank int64 [8,R] zero-filled by the caller.
extern "C" int gtos_relbatch_dev_all_fill(const int64_t* geom, int64_t R, void** tab, void* workspace, size_t workspace_bytes, void* stream) {
    if (!geom || !tab || !workspace || R < N_SPECIAL_ALL) return -1;
    const Geom G = geom_of(geom);
    if (!geom_ok(G) || G.mode != MODE_ALL || G.K < 1) return -1;
    for (int k = 0; k < T_TABLE_COUNT; ++k) if (!tab[k]) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Graphs gr = graphs_of(tab);
    const int64_t total = G.T + N_SPECIAL_ALL;
    if (R > total) return -1;
    size_t bytes = workspace_bytes;
    GTOS_RB_HIP(rocprim::radix_sort_pairs(workspace, bytes, (const uint32_t*)tab[T_FIRST_POS], (uint32_t*)tab[T_FIRST_ALT], (const int32_t*)tab[T_SEG_ID],
                                          (int32_t*)tab[T_SORTED_SEG], (size_t)R, 0, bits_for(total), s));
    hipLaunchKernelGGL(k_type_of_segment, grid_for(R, 256), dim3(256), 0, s, R, (const int32_t*)tab[T_SORTED_SEG], (const uint64_t*)tab[T_SEG_KEY],
                       (int32_t*)tab[T_TYPE_OF_SEG], (int64_t*)tab[T_BANK], (int64_t*)tab[T_LENGTH]);
    GTOS_RB_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scatter_relation_all, grid_for(total, 256), dim3(256), 0, s, total, G, gr, (const int32_t*)tab[T_SPOS], (const uint64_t*)tab[T_CUM],
                       (const int32_t*)tab[T_TYPE_OF_SEG], (const uint64_t*)tab[T_CUM_ALT], (int64_t*)tab[T_RELATION]);
    GTOS_RB_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_cls_cells_all, grid_for(G.S, 256), dim3(256), 0, s, G, gr, (int64_t*)tab[T_RELATION]);
    GTOS_RB_LAUNCH_CHECK();
    return 0;
}
