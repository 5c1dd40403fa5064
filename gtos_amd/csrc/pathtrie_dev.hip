// Path tries of a relation bank built on the GPU: the launch glue around the per-thread stages of trie_kernels.h (whose logic
// tests/test_pathtrie.py proves equal to csrc_host/pathtrie.cpp on the host, through oracle/trie_emul.cpp) plus rocPRIM's radix
// sort and scan.  Two phases with one host read between them (gtos_amd/pathtrie_hip.py): phase A is R-sized work that fixes the
// node counts, phase B fills the node- and row-sized arrays the caller then allocates exactly.
//
// STATUS (end of round 3): opt-in (Prefetcher(device_tries="hip"), bench.py --device-tries hip, GTOS_TRIE_DEVICE=hip).  First run on
// an MI355X in the round's last GPU seconds (tools/hip_trie_check.py -> profiles/r3w_hip_trie_check.json): every array equal to the
// host builder's on a 3,000-path bank with duplicates and on the C2 bank (R = 434,624, N = 2,497,192), 1.5 ms per C2 build against
// ~12 ms for the torch-op builder (gtos_amd/pathtrie_device.py) and 56 ms for the host builder on that box; loader in the loop with
// 2 worker processes: 65.8 ms per step (torch-op tries 72.5, host tries 86.5; pre-built batch 61.3).
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "trie_kernels.h"

using namespace gtos_trie;

namespace {

#define GTOS_TRIE_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return 100 + (int)e_; } while (0)
#define GTOS_TRIE_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return 100 + (int)e_; } while (0)

inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

__global__ void k_make_keys(int L, int64_t R, const int64_t* bank, const int64_t* length, uint64_t* key_f, uint64_t* key_b, int32_t* id_f,
                            int32_t* id_b, uint8_t* len8, int32_t* err) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < R) make_keys(s, L, R, bank, length, key_f, key_b, id_f, id_b, len8, err);
}
__global__ void k_open_flags(int64_t R, const uint64_t* key, uint8_t* newmask, V8* bits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) open_flags(i, key, newmask, bits);
}
__global__ void k_level_offsets(const V8* cum, int64_t R, int32_t* lvl, int32_t* sizes_side) {
    if (blockIdx.x == 0 && threadIdx.x == 0) level_offsets(cum, R, lvl, sizes_side);
}
__global__ void k_length_onehot(int64_t R, const int32_t* order_f, const uint8_t* len8, V8* hot) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) length_onehot(i, order_f, len8, hot);
}
__global__ void k_packed_geometry(const V8* cumlen, int64_t R, int32_t* start, int32_t* batch_sizes, int64_t* offs) {
    if (blockIdx.x == 0 && threadIdx.x == 0) packed_geometry(cumlen, R, start, batch_sizes, offs);
}
__global__ void k_packed_position(int64_t R, const int32_t* order_f, const uint8_t* len8, const V8* cumlen, const int32_t* start,
                                  int32_t* seq_order, int32_t* seq_pos, int64_t* seq_order64, int64_t* seq_pos64, int32_t* lexf_of_m) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) packed_position(i, order_f, len8, cumlen, start, seq_order, seq_pos, seq_order64, seq_pos64, lexf_of_m);
}
__global__ void k_lex_position(int64_t R, const int32_t* order_b, int32_t* lexb) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) lex_position(i, order_b, lexb);
}
__global__ void k_write_nodes(int64_t R, Side t) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) write_nodes(i, t);
}
__global__ void k_children(int64_t n, Side t) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) children(v, t);
}
__global__ void k_fill_rows(int64_t R, const int32_t* seq_order, const uint8_t* len8, const int32_t* lexf_of_m, const int32_t* lexb,
                            const int64_t* offs, const int32_t* tab_f, const int32_t* tab_b, int32_t* row_pf, int32_t* row_sf) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < R) fill_rows(m, seq_order, len8, lexf_of_m, lexb, offs, tab_f, tab_b, row_pf, row_sf);
}
__global__ void k_node_offsets(int64_t N, const uint32_t* row_key, int32_t* off, int32_t n) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < N) node_offsets(e, N, row_key, off, n);
}
__global__ void k_node_counts(int64_t n, int chunk, Side t) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u < n) node_counts(u, chunk, t);
}
__global__ void k_node_records(int64_t n, int chunk, Side t) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u < n) node_records(u, chunk, t);
}
__global__ void k_side_sizes(Side t, int32_t* sizes_side) {
    if (blockIdx.x == 0 && threadIdx.x == 0) side_sizes(t, sizes_side);
}
__global__ void k_wave_range(int64_t n_waves, int rows_per_wave, Side t, const int32_t* sizes_side) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w <= n_waves) wave_range(w, n_waves, rows_per_wave, t, sizes_side);
}

int bits_for(int64_t n) {                 // radix-sort key bits that can be set in values below n
    int b = 1;
    while ((1ll << b) < n && b < 32) ++b;
    return b;
}

}  // namespace

// bytes[0] = rocPRIM temporary storage the two phases need for R paths and N rows.
extern "C" int gtos_pathtrie_dev_workspace(int64_t R, int64_t N, int64_t* bytes_out) {
    if (R <= 0 || N <= 0 || !bytes_out) return -1;
    size_t a = 0, b = 0, c = 0;
    (void)rocprim::radix_sort_pairs(nullptr, a, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr,
                              (size_t)R, 0, 64, (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, b, (const V8*)nullptr, (V8*)nullptr, (size_t)(R > N ? R : N), Add8(), (hipStream_t)0);
    (void)rocprim::radix_sort_pairs(nullptr, c, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr,
                              (size_t)N, 0, 32, (hipStream_t)0);
    size_t m = a > b ? a : b;
    m = m > c ? m : c;
    bytes_out[0] = (int64_t)(m + 256);
    return 0;
}

// Phase A: keys, key sorts, "opens a node" scans, level offsets (sizes[SZ_PF / SZ_SF + S_NODES, S_LEVEL..]), packed order.
// sizes[SZ_ERR] != 0 afterwards: a path outside 1..8 labels or a label id outside [0, 255) -- the caller falls back.
extern "C" int gtos_pathtrie_dev_phase_a(int L, int64_t R, const int64_t* bank, const int64_t* length, void** common, void** pf, void** sf,
                                         int32_t* sizes, void* workspace, size_t workspace_bytes, void* stream) {
    if (L <= 0 || R <= 0 || !bank || !length || !common || !pf || !sf || !sizes || !workspace) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Side F = side_of(pf), B = side_of(sf);
    uint8_t* len8 = (uint8_t*)common[C_LEN8];
    V8 *scratch = (V8*)common[C_SCRATCH], *cumlen = (V8*)common[C_CUMLEN];
    uint64_t* key_alt = (uint64_t*)common[C_KEY_ALT];
    int32_t* id_alt = (int32_t*)common[C_ID_ALT];
    GTOS_TRIE_HIP(hipMemsetAsync(sizes, 0, SZ_TOTAL * sizeof(int32_t), s));
    hipLaunchKernelGGL(k_make_keys, grid_for(R), dim3(256), 0, s, L, R, bank, length, F.key, B.key, F.order, B.order, len8, sizes + SZ_ERR);
    GTOS_TRIE_LAUNCH_CHECK();
    int k = 0;
    for (Side* t : {&F, &B}) {
        size_t bytes = workspace_bytes;
        GTOS_TRIE_HIP(rocprim::radix_sort_pairs(workspace, bytes, (const uint64_t*)t->key, key_alt, (const int32_t*)t->order, id_alt, (size_t)R,
                                                0, 64, s));
        GTOS_TRIE_HIP(hipMemcpyAsync(t->key, key_alt, R * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
        GTOS_TRIE_HIP(hipMemcpyAsync(t->order, id_alt, R * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_open_flags, grid_for(R), dim3(256), 0, s, R, (const uint64_t*)t->key, t->newmask, scratch);
        GTOS_TRIE_LAUNCH_CHECK();
        bytes = workspace_bytes;
        GTOS_TRIE_HIP(rocprim::inclusive_scan(workspace, bytes, (const V8*)scratch, t->cum, (size_t)R, Add8(), s));
        hipLaunchKernelGGL(k_level_offsets, dim3(1), dim3(64), 0, s, (const V8*)t->cum, R, t->lvl, sizes + (k++ ? SZ_SF : SZ_PF));
        GTOS_TRIE_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_length_onehot, grid_for(R), dim3(256), 0, s, R, (const int32_t*)F.order, (const uint8_t*)len8, scratch);
    GTOS_TRIE_LAUNCH_CHECK();
    size_t bytes = workspace_bytes;
    GTOS_TRIE_HIP(rocprim::inclusive_scan(workspace, bytes, (const V8*)scratch, cumlen, (size_t)R, Add8(), s));
    hipLaunchKernelGGL(k_packed_geometry, dim3(1), dim3(64), 0, s, (const V8*)cumlen, R, (int32_t*)common[C_START], (int32_t*)common[C_BATCH],
                       (int64_t*)common[C_OFFS]);
    GTOS_TRIE_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_packed_position, grid_for(R), dim3(256), 0, s, R, (const int32_t*)F.order, (const uint8_t*)len8, (const V8*)cumlen,
                       (const int32_t*)common[C_START], (int32_t*)common[C_SEQ_ORDER], (int32_t*)common[C_SEQ_POS],
                       (int64_t*)common[C_SEQ_ORDER64], (int64_t*)common[C_SEQ_POS64], (int32_t*)common[C_LEXF]);
    GTOS_TRIE_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_lex_position, grid_for(R), dim3(256), 0, s, R, (const int32_t*)B.order, (int32_t*)common[C_LEXB]);
    GTOS_TRIE_LAUNCH_CHECK();
    return 0;
}

// Phase B: n_pf / n_sf are the node counts the host read from sizes[] after phase A (the node-sized entries of pf / sf are allocated
// from them; child_off zero-filled).  Fills every remaining array and the rest of sizes[].
extern "C" int gtos_pathtrie_dev_phase_b(int64_t R, int64_t N, int n_pf, int n_sf, int chunk, int rows_per_wave, void** common, void** pf,
                                         void** sf, int32_t* sizes, void* workspace, size_t workspace_bytes, void* stream) {
    if (R <= 0 || N < R || n_pf <= 0 || n_sf <= 0 || chunk <= 0 || chunk > 64 || rows_per_wave <= 0 || !common || !pf || !sf || !sizes ||
        !workspace) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Side F = side_of(pf), B = side_of(sf);
    F.row_node = (int32_t*)common[C_ROW_PF];
    B.row_node = (int32_t*)common[C_ROW_SF];
    const uint8_t* len8 = (const uint8_t*)common[C_LEN8];
    const int32_t* iota = (const int32_t*)common[C_IOTA];
    const int64_t n_waves = (N + rows_per_wave - 1) / rows_per_wave > 0 ? (N + rows_per_wave - 1) / rows_per_wave : 1;
    int k = 0;
    for (Side* t : {&F, &B}) {
        const int64_t n = k++ ? n_sf : n_pf;
        hipLaunchKernelGGL(k_write_nodes, grid_for(R), dim3(256), 0, s, R, *t);
        GTOS_TRIE_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_children, grid_for(n), dim3(256), 0, s, n, *t);
        GTOS_TRIE_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_fill_rows, grid_for(R), dim3(256), 0, s, R, (const int32_t*)common[C_SEQ_ORDER], len8, (const int32_t*)common[C_LEXF],
                       (const int32_t*)common[C_LEXB], (const int64_t*)common[C_OFFS], (const int32_t*)F.node_tab, (const int32_t*)B.node_tab,
                       F.row_node, B.row_node);
    GTOS_TRIE_LAUNCH_CHECK();
    k = 0;
    for (Side* t : {&F, &B}) {
        int32_t* sz = sizes + (k ? SZ_SF : SZ_PF);
        const int64_t n = k++ ? n_sf : n_pf;
        size_t bytes = workspace_bytes;
        // rows sorted by node, stable in the row id (radix sort is stable; only the bits a node id can set are sorted)
        GTOS_TRIE_HIP(rocprim::radix_sort_pairs(workspace, bytes, (const uint32_t*)t->row_node, t->row_key, iota, t->rows, (size_t)N, 0,
                                                bits_for(n), s));
        hipLaunchKernelGGL(k_node_offsets, grid_for(N), dim3(256), 0, s, N, (const uint32_t*)t->row_key, t->off, (int32_t)n);
        GTOS_TRIE_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_node_counts, grid_for(n), dim3(256), 0, s, n, chunk, *t);
        GTOS_TRIE_LAUNCH_CHECK();
        bytes = workspace_bytes;
        GTOS_TRIE_HIP(rocprim::inclusive_scan(workspace, bytes, (const V8*)t->aux, t->aux_cum, (size_t)n, Add8(), s));
        hipLaunchKernelGGL(k_node_records, grid_for(n), dim3(256), 0, s, n, chunk, *t);
        GTOS_TRIE_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_side_sizes, dim3(1), dim3(64), 0, s, *t, sz);
        GTOS_TRIE_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_wave_range, grid_for(n_waves + 1), dim3(256), 0, s, n_waves, rows_per_wave, *t, (const int32_t*)sz);
        GTOS_TRIE_LAUNCH_CHECK();
    }
    return 0;
}
