// HBM bandwidth probe for gfx950: what do streaming, row-gather and mixed read/write patterns reach on an MI355X?
// The GRU step kernels, the segmented sums and the attention kernels of this repo all settle near 4.5 TB/s of measured
// traffic; this probe separates "the access pattern's ceiling" from "the kernel's own inefficiency".
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/membw_probe tools/probes/membw_probe.hip && tools/probes/membw_probe
// Patterns (all 16-byte loads/stores per lane, U loads in flight per lane before the first use):
//   stream_read      : contiguous
//   gather S seq/rnd : rows of S bytes (S = 128 .. 2048) of a table with row pitch P, row ids sequential or a random permutation
//   copy             : contiguous read + contiguous write
//   gather+write     : random rows of S bytes read, S bytes written contiguously (the shape of a GRU step / segment sum)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) uint32_t U128;

// one "item" = 16 bytes: item t of the logical stream lives at byte rows[t / ipr] * pitch + (t % ipr) * 16 (ipr = S / 16)
template <int U, bool WRITE>
__global__ __launch_bounds__(256) void probe_kernel(const char* __restrict__ src, const uint32_t* __restrict__ rows, int ipr_shift,
                                                    int64_t pitch, int64_t n_items, char* __restrict__ dst, uint32_t* __restrict__ sink) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    const int ipr_mask = (1 << ipr_shift) - 1;
    U128 acc = {0u, 0u, 0u, 0u};
    for (int64_t t0 = tid; t0 < n_items; t0 += nth * U) {
        U128 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t t = t0 + (int64_t)u * nth;
            if (t < n_items) {
                const int64_t r = rows ? (int64_t)rows[t >> ipr_shift] : (t >> ipr_shift);
                v[u] = *reinterpret_cast<const U128*>(src + r * pitch + (int64_t)(t & ipr_mask) * 16);
            } else v[u] = U128{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (WRITE) {
                const int64_t t = t0 + (int64_t)u * nth;
                if (t < n_items) *reinterpret_cast<U128*>(dst + t * 16) = v[u];
            } else acc ^= v[u];
        }
    }
    if (!WRITE && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}

// fill only (write ceiling)
__global__ __launch_bounds__(256) void fill_kernel(char* __restrict__ dst, int64_t n_items) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = tid; t < n_items; t += nth) *reinterpret_cast<U128*>(dst + t * 16) = U128{1u, 2u, 3u, 4u};
}

template <int U, bool WRITE>
static float run(const char* src, const uint32_t* rows, int ipr_shift, int64_t pitch, int64_t n_items, char* dst, uint32_t* sink, int blocks) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe_kernel<U, WRITE>), dim3(blocks), dim3(256), 0, 0, src, rows, ipr_shift, pitch, n_items, dst, sink);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe_kernel<U, WRITE>), dim3(blocks), dim3(256), 0, 0, src, rows, ipr_shift, pitch, n_items, dst, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / reps;
}

int main() {
    const int64_t pitch = 2048;                       // bytes per table row (a d4 / gates row of the GRU: 4h bf16)
    const int64_t n_rows = 3 << 20;                   // 3 M rows x 2 KB = 6 GB table: far past L2 (32 MB) and the Infinity Cache (256 MB)
    const int64_t tab_bytes = n_rows * pitch;
    char *src, *dst;
    uint32_t *seq_d, *rnd_d, *blk_d, *sink;
    CK(hipMalloc(&src, tab_bytes)); CK(hipMalloc(&dst, tab_bytes));
    CK(hipMemset(src, 1, tab_bytes)); CK(hipMemset(dst, 0, tab_bytes));
    CK(hipMalloc(&sink, 64)); CK(hipMemset(sink, 0, 64));
    std::vector<uint32_t> perm(n_rows);
    for (int64_t i = 0; i < n_rows; ++i) perm[i] = (uint32_t)i;
    std::mt19937_64 rng(12345);
    std::shuffle(perm.begin(), perm.end(), rng);
    CK(hipMalloc(&rnd_d, n_rows * 4)); CK(hipMemcpy(rnd_d, perm.data(), n_rows * 4, hipMemcpyHostToDevice));
    // "blocked" permutation: runs of 8 consecutive rows at random places (rows of one trie node inside a length class)
    std::vector<uint32_t> blk(n_rows);
    { std::vector<uint32_t> heads(n_rows / 8); for (size_t i = 0; i < heads.size(); ++i) heads[i] = (uint32_t)i; std::shuffle(heads.begin(), heads.end(), rng);
      for (int64_t i = 0; i < n_rows; ++i) blk[i] = heads[i / 8] * 8 + (uint32_t)(i % 8); }
    CK(hipMalloc(&blk_d, n_rows * 4)); CK(hipMemcpy(blk_d, blk.data(), n_rows * 4, hipMemcpyHostToDevice));
    (void)seq_d;
    printf("%-44s %10s %10s\n", "pattern", "ms", "TB/s");
    auto report = [&](const char* name, float ms, double bytes) { printf("%-44s %10.3f %10.2f\n", name, ms, bytes / ms / 1e9); fflush(stdout); };

    const int64_t items_full = tab_bytes / 16;
    for (int blocks : {256 * 4, 256 * 8, 256 * 16}) {
        char nm[128];
        snprintf(nm, sizeof nm, "stream_read U=4 blocks=%d", blocks);
        report(nm, run<4, false>(src, nullptr, 7, pitch, items_full, dst, sink, blocks), (double)tab_bytes);
        snprintf(nm, sizeof nm, "stream_read U=8 blocks=%d", blocks);
        report(nm, run<8, false>(src, nullptr, 7, pitch, items_full, dst, sink, blocks), (double)tab_bytes);
        snprintf(nm, sizeof nm, "copy U=4 blocks=%d (r+w bytes)", blocks);
        report(nm, run<4, true>(src, nullptr, 7, pitch, items_full, dst, sink, blocks), 2.0 * tab_bytes);
        snprintf(nm, sizeof nm, "copy U=8 blocks=%d (r+w bytes)", blocks);
        report(nm, run<8, true>(src, nullptr, 7, pitch, items_full, dst, sink, blocks), 2.0 * tab_bytes);
    }
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, dst, items_full);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, dst, items_full);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        report("fill (write only)", ms / 5, (double)tab_bytes);
    }
    // row gathers: S bytes of every 2 KB row (so S = 2048 reads everything, S = 128 one 128-byte piece per row)
    for (int shift = 3; shift <= 7; ++shift) {            // S = 128, 256, 512, 1024, 2048
        const int S = 16 << shift;
        const int64_t items = n_rows << shift;
        const double bytes = (double)n_rows * S;
        char nm[128];
        for (int blocks : {256 * 8, 256 * 16}) {
            snprintf(nm, sizeof nm, "gather S=%4d sequential rows U=8 blocks=%d", S, blocks);
            report(nm, run<8, false>(src, nullptr, shift, pitch, items, dst, sink, blocks), bytes);
            snprintf(nm, sizeof nm, "gather S=%4d random rows     U=8 blocks=%d", S, blocks);
            report(nm, run<8, false>(src, rnd_d, shift, pitch, items, dst, sink, blocks), bytes);
        }
        snprintf(nm, sizeof nm, "gather S=%4d runs of 8 rows  U=8 blocks=4096", S);
        report(nm, run<8, false>(src, blk_d, shift, pitch, items, dst, sink, 4096), bytes);
        snprintf(nm, sizeof nm, "gather S=%4d random rows     U=4 blocks=4096", S);
        report(nm, run<4, false>(src, rnd_d, shift, pitch, items, dst, sink, 4096), bytes);
        snprintf(nm, sizeof nm, "gather+write S=%4d random U=8 (r+w bytes)", S);
        report(nm, run<8, true>(src, rnd_d, shift, pitch, items, dst, sink, 4096), 2.0 * bytes);
        snprintf(nm, sizeof nm, "gather+write S=%4d seq    U=8 (r+w bytes)", S);
        report(nm, run<8, true>(src, nullptr, shift, pitch, items, dst, sink, 4096), 2.0 * bytes);
    }
    return 0;
}
