// The relation index of the factored attention operand built on the GPU: the launch glue around the per-thread stages of
// relindex_kernels.h (whose logic tests/test_relindex_dev.py proves equal to csrc_host/relindex.cpp on the host, running the same code
// as serial loops) plus rocPRIM's radix sort and scan.  Two phases with one host read between them (gtos_amd/relindex_hip.py).
//
// STATUS (end of round 3): compiles for gfx950; written after the round's GPU minutes were spent, so it has not run yet.  Opt-in.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include "relindex_kernels.h"

using namespace gtos_relindex_dev;

namespace {

#define GTOS_RI_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return 100 + (int)e_; } while (0)
#define GTOS_RI_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return 100 + (int)e_; } while (0)

inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

__global__ void k_cell_key(Geom G, const int64_t* relation, uint32_t* key, int32_t* val, int32_t* sizes) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < G.P) cell_key(e, G, relation, key, val, sizes);
}
__global__ void k_type_bounds(Geom G, const uint32_t* skey, int32_t* cnt, uint32_t* cum_cnt) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < G.R) type_bounds(t, G, skey, cnt, cum_cnt);
}
__global__ void k_idx_cell(Geom G, const int64_t* relation, const int32_t* cnt, int32_t* idx_q, int32_t* idx_k) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < G.P) idx_cell(p, G, relation, cnt, idx_q, idx_k);
}
__global__ void k_type_counts(Geom G, const int32_t* cnt, uint32_t* nch, uint32_t* heavy) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < G.R) type_counts(t, G, cnt, nch, heavy);
}
__global__ void k_type_chunks(Geom G, const int32_t* cnt, const uint32_t* cum_cnt, const uint32_t* nch, const uint32_t* cum_nch, const uint32_t* heavy,
                              const uint32_t* cum_heavy, const int32_t* pair_sorted, Chunks ch, int32_t* heavy_types) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < G.R) type_chunks(t, G, cnt, cum_cnt, nch, cum_nch, heavy, cum_heavy, pair_sorted, ch, heavy_types);
}
__global__ void k_sizes_a(Geom G, const uint32_t* cum_nch, const uint32_t* cum_heavy, int32_t* sizes) {
    if (blockIdx.x == 0 && threadIdx.x == 0) sizes_a(G, cum_nch, cum_heavy, sizes);
}
__global__ void k_roam_key(int64_t nchunks, Chunks ch, uint32_t* key, int32_t* val, V8* home_cost) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nchunks) roam_key(c, ch, key, val, home_cost);
}
__global__ void k_roam_setup(int64_t nchunks, const uint32_t* rkey_sorted, const V8* cum_cost, unsigned long long* load, int32_t* sizes) {
    if (blockIdx.x == 0 && threadIdx.x == 0) roam_setup(nchunks, rkey_sorted, cum_cost, load, sizes);
}
__global__ void k_roam_cost(int64_t nchunks, const int32_t* roam_sorted, Chunks ch, int32_t* rcost, const int32_t* sizes) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nchunks) roam_cost(q, roam_sorted, ch, rcost, sizes);
}
// greedy_homes of relindex_kernels.h on ONE wave: the recurrence is serial (each placement sees the loads the previous ones left), but a
// lone thread would wait out a memory round trip per chunk (15 k roaming chunks at C2).  Here 64 lanes load 64 costs at once, every
// lane walks the same 64 placements on its own copy of the eight loads (uniform work), lane k keeps the k-th answer, and the tile's
// homes leave as one coalesced store.
__global__ void __launch_bounds__(64) k_greedy_wave(const int32_t* rcost, int32_t* home_q, unsigned long long* load, const int32_t* sizes) {
    const int lane = threadIdx.x;
    const int32_t n_roam = sizes[IZ_NROAM];
    long long l0 = (long long)load[0], l1 = (long long)load[1], l2 = (long long)load[2], l3 = (long long)load[3];
    long long l4 = (long long)load[4], l5 = (long long)load[5], l6 = (long long)load[6], l7 = (long long)load[7];
    for (int32_t base = 0; base < n_roam; base += 64) {
        const int32_t q = base + lane;
        const int32_t mine = q < n_roam ? rcost[q] : 0;
        const int lim = n_roam - base < 64 ? n_roam - base : 64;
        int32_t my_home = 0;
        for (int k = 0; k < lim; ++k) {
            const long long c = (long long)__shfl(mine, k, 64);
            int best = 0;
            long long lo = l0;
            if (l1 < lo) { lo = l1; best = 1; }
            if (l2 < lo) { lo = l2; best = 2; }
            if (l3 < lo) { lo = l3; best = 3; }
            if (l4 < lo) { lo = l4; best = 4; }
            if (l5 < lo) { lo = l5; best = 5; }
            if (l6 < lo) { lo = l6; best = 6; }
            if (l7 < lo) { lo = l7; best = 7; }
            l0 += best == 0 ? c : 0; l1 += best == 1 ? c : 0; l2 += best == 2 ? c : 0; l3 += best == 3 ? c : 0;
            l4 += best == 4 ? c : 0; l5 += best == 5 ? c : 0; l6 += best == 6 ? c : 0; l7 += best == 7 ? c : 0;
            if (lane == k) my_home = best;
        }
        if (q < n_roam) home_q[q] = my_home;
    }
    if (lane == 0) {
        load[0] = (unsigned long long)l0; load[1] = (unsigned long long)l1; load[2] = (unsigned long long)l2; load[3] = (unsigned long long)l3;
        load[4] = (unsigned long long)l4; load[5] = (unsigned long long)l5; load[6] = (unsigned long long)l6; load[7] = (unsigned long long)l7;
    }
}
__global__ void k_scatter_homes(int64_t nchunks, const int32_t* roam_sorted, const int32_t* home_q, Chunks ch, const int32_t* sizes) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nchunks) scatter_homes(q, roam_sorted, home_q, ch, sizes);
}
__global__ void k_final_key(int64_t nchunks, Chunks ch, uint64_t* key, int32_t* val, V8* home_hot) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nchunks) final_key(c, ch, key, val, home_hot);
}
__global__ void k_gather(int64_t nchunks, const int32_t* perm, Chunks ch, int32_t* chunk_type, int32_t* chunk_start, int32_t* chunk_count,
                         int32_t* chunk_slot, const V8* cum_hot, int32_t* xcd_off) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nchunks) gather_chunk(q, perm, ch, chunk_type, chunk_start, chunk_count, chunk_slot);
    if (q == 0) xcd_offsets(nchunks, cum_hot, xcd_off);
}

int bits_for(int64_t n) {                 // radix-sort key bits that can be set in values below n
    int b = 1;
    while ((1ll << b) < n && b < 32) ++b;
    return b;
}

}  // namespace

// bytes_out[0] = rocPRIM temporary storage the two phases need for arrays of up to `n` elements (max(P, R)).
extern "C" int gtos_relindex_dev_workspace(int64_t n, int64_t* bytes_out) {
    if (n <= 0 || !bytes_out) return -1;
    size_t a = 0, b = 0, c = 0, d = 0;
    (void)rocprim::inclusive_scan(nullptr, d, (const V8*)nullptr, (V8*)nullptr, (size_t)n, Add8(), (hipStream_t)0);
    (void)rocprim::radix_sort_pairs(nullptr, a, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)n, 0, 64,
                                    (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, rocprim::plus<uint32_t>(), (hipStream_t)0);
    (void)rocprim::radix_sort_pairs(nullptr, c, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)n, 0, 32,
                                    (hipStream_t)0);
    size_t m = a > b ? a : b;
    m = m > c ? m : c;
    m = m > d ? m : d;
    bytes_out[0] = (int64_t)(m + 256);
    return 0;
}

// Phase A.  geom: int64[GE_COUNT] host integers; tab: host table of device pointers (enum T_* of relindex_kernels.h); sizes zero-filled
// by the caller.  Afterwards sizes = {error flag, chunks, heavy types, 0}.
extern "C" int gtos_relindex_dev_phase_a(const int64_t* geom, void** tab, void* workspace, size_t workspace_bytes, void* stream) {
    if (!geom || !tab || !workspace) return -1;
    const Geom G = geom_of(geom);
    if (!geom_ok(G)) return -1;
    for (int k = 0; k <= T_LAST_OF_PHASE_A; ++k) if (!tab[k]) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t* relation = (const int64_t*)tab[T_RELATION];
    int32_t *cnt = (int32_t*)tab[T_CNT], *sizes = (int32_t*)tab[T_SIZES], *pair_sorted = (int32_t*)tab[T_PAIR_SORTED];
    uint32_t *cum_cnt = (uint32_t*)tab[T_CUM_CNT], *nch = (uint32_t*)tab[T_NCH], *cum_nch = (uint32_t*)tab[T_CUM_NCH];
    uint32_t *heavy = (uint32_t*)tab[T_HEAVY], *cum_heavy = (uint32_t*)tab[T_CUM_HEAVY];
    const Chunks ch = chunks_of(tab);
    hipLaunchKernelGGL(k_cell_key, grid_for(G.P), dim3(256), 0, s, G, relation, (uint32_t*)tab[T_KEY], (int32_t*)tab[T_VAL], sizes);
    GTOS_RI_LAUNCH_CHECK();
    size_t bytes = workspace_bytes;
    GTOS_RI_HIP(rocprim::radix_sort_pairs(workspace, bytes, (const uint32_t*)tab[T_KEY], (uint32_t*)tab[T_SKEY], (const int32_t*)tab[T_VAL], pair_sorted,
                                          (size_t)G.P, 0, bits_for(G.R), s));
    hipLaunchKernelGGL(k_type_bounds, grid_for(G.R), dim3(256), 0, s, G, (const uint32_t*)tab[T_SKEY], cnt, cum_cnt);
    GTOS_RI_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_idx_cell, grid_for(G.P), dim3(256), 0, s, G, relation, (const int32_t*)cnt, (int32_t*)tab[T_IDX_Q], (int32_t*)tab[T_IDX_K]);
    GTOS_RI_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_type_counts, grid_for(G.R), dim3(256), 0, s, G, (const int32_t*)cnt, nch, heavy);
    GTOS_RI_LAUNCH_CHECK();
    bytes = workspace_bytes;
    GTOS_RI_HIP(rocprim::inclusive_scan(workspace, bytes, (const uint32_t*)nch, cum_nch, (size_t)G.R, rocprim::plus<uint32_t>(), s));
    bytes = workspace_bytes;
    GTOS_RI_HIP(rocprim::inclusive_scan(workspace, bytes, (const uint32_t*)heavy, cum_heavy, (size_t)G.R, rocprim::plus<uint32_t>(), s));
    hipLaunchKernelGGL(k_type_chunks, grid_for(G.R), dim3(256), 0, s, G, (const int32_t*)cnt, (const uint32_t*)cum_cnt, (const uint32_t*)nch,
                       (const uint32_t*)cum_nch, (const uint32_t*)heavy, (const uint32_t*)cum_heavy, (const int32_t*)pair_sorted, ch,
                       (int32_t*)tab[T_HEAVY_TYPES]);
    GTOS_RI_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sizes_a, dim3(1), dim3(64), 0, s, G, (const uint32_t*)cum_nch, (const uint32_t*)cum_heavy, sizes);
    GTOS_RI_LAUNCH_CHECK();
    return 0;
}

// Phase B: nchunks = sizes[IZ_NCHUNKS] as the host read it (> 0); fills chunk_type / _start / _count / _slot [nchunks] and xcd_off [9].
extern "C" int gtos_relindex_dev_phase_b(const int64_t* geom, int64_t nchunks, void** tab, void* workspace, size_t workspace_bytes, void* stream) {
    if (!geom || !tab || !workspace || nchunks <= 0) return -1;
    const Geom G = geom_of(geom);
    if (!geom_ok(G) || nchunks > G.R + G.P / G.chunk + 1) return -1;
    for (int k = 0; k < T_TABLE_COUNT; ++k) if (!tab[k]) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Chunks ch = chunks_of(tab);
    int32_t* sizes = (int32_t*)tab[T_SIZES];
    V8 *v8 = (V8*)tab[T_V8], *v8_cum = (V8*)tab[T_V8_CUM];
    hipLaunchKernelGGL(k_roam_key, grid_for(nchunks), dim3(256), 0, s, nchunks, ch, (uint32_t*)tab[T_RKEY], (int32_t*)tab[T_RVAL], v8);
    GTOS_RI_LAUNCH_CHECK();
    size_t bytes = workspace_bytes;
    GTOS_RI_HIP(rocprim::radix_sort_pairs(workspace, bytes, (const uint32_t*)tab[T_RKEY], (uint32_t*)tab[T_RKEY_S], (const int32_t*)tab[T_RVAL],
                                          (int32_t*)tab[T_ROAM_SORTED], (size_t)nchunks, 0, 32, s));
    bytes = workspace_bytes;
    GTOS_RI_HIP(rocprim::inclusive_scan(workspace, bytes, (const V8*)v8, v8_cum, (size_t)nchunks, Add8(), s));
    hipLaunchKernelGGL(k_roam_setup, dim3(1), dim3(64), 0, s, nchunks, (const uint32_t*)tab[T_RKEY_S], (const V8*)v8_cum, (unsigned long long*)tab[T_LOAD],
                       sizes);
    GTOS_RI_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_roam_cost, grid_for(nchunks), dim3(256), 0, s, nchunks, (const int32_t*)tab[T_ROAM_SORTED], ch, (int32_t*)tab[T_RCOST],
                       (const int32_t*)sizes);
    GTOS_RI_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_greedy_wave, dim3(1), dim3(64), 0, s, (const int32_t*)tab[T_RCOST], (int32_t*)tab[T_HOME_Q], (unsigned long long*)tab[T_LOAD],
                       (const int32_t*)sizes);
    GTOS_RI_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scatter_homes, grid_for(nchunks), dim3(256), 0, s, nchunks, (const int32_t*)tab[T_ROAM_SORTED], (const int32_t*)tab[T_HOME_Q], ch,
                       (const int32_t*)sizes);
    GTOS_RI_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_final_key, grid_for(nchunks), dim3(256), 0, s, nchunks, ch, (uint64_t*)tab[T_FKEY], (int32_t*)tab[T_FVAL], v8);
    GTOS_RI_LAUNCH_CHECK();
    bytes = workspace_bytes;
    GTOS_RI_HIP(rocprim::radix_sort_pairs(workspace, bytes, (const uint64_t*)tab[T_FKEY], (uint64_t*)tab[T_FKEY_S], (const int32_t*)tab[T_FVAL],
                                          (int32_t*)tab[T_PERM], (size_t)nchunks, 0, 48, s));
    bytes = workspace_bytes;
    GTOS_RI_HIP(rocprim::inclusive_scan(workspace, bytes, (const V8*)v8, v8_cum, (size_t)nchunks, Add8(), s));
    hipLaunchKernelGGL(k_gather, grid_for(nchunks), dim3(256), 0, s, nchunks, (const int32_t*)tab[T_PERM], ch, (int32_t*)tab[T_CHUNK_TYPE],
                       (int32_t*)tab[T_CHUNK_START], (int32_t*)tab[T_CHUNK_COUNT], (int32_t*)tab[T_CHUNK_SLOT], (const V8*)v8_cum,
                       (int32_t*)tab[T_XCD_OFF]);
    GTOS_RI_LAUNCH_CHECK();
    return 0;
}
