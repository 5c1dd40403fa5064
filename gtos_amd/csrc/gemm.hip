// MFMA GEMM for gfx950 with fused bias / ReLU / dropout epilogue and split-K accumulation.
//
//   C[M,N] (+)= act( opA(A)[M,K] . opB(B)[K,N] + bias[N] )           (all row-major)
//
// It replaces the ATen F.linear / mm calls of the reference's projections and FFN
// (generator/graph_transformer.py:61-63,106-122,166,176-197) and their autograd mm's:
//   forward  Y = X W^T      -> TA=0, TB=1 (W stored [N,K] like nn.Linear.weight)
//   dX = dY W               -> TA=0, TB=0
//   dW = dY^T X             -> TA=1, TB=0, split-K, fp32 atomic accumulate
//
// Structure of the main kernel (one 256-thread workgroup = 4 waves, 2x2, per 128x128 output tile; each wave 64x64 =
// 4x4 MFMA tiles; bf16 inputs: v_mfma_f32_16x16x32_bf16, fp32 inputs: the exact-fp32 v_mfma_f32_16x16x4_f32):
//   * bf16 operands of EVERY layout go global -> LDS directly (global_load_lds_dwordx4), one 32 KB stage, up to 4
//     workgroups per CU covering each other's load latency: K-contiguous operands as 128-byte rows of 8 16-byte chunks
//     with the chunk index XOR-swizzled by ((row>>1)&7) ^ ((row>>4)&3) on the DMA's source address (conflict-free
//     fragment ds_read_b128), M/N-contiguous operands as a [k][128] tile read with ds_read_b64_tr_b16;
//   * fp32 operands that are not K-contiguous keep a register-staged path: 512 threads, two LDS stages, prefetch
//     distance 2, branch-free loads (out-of-range vectors read a block of zeros), 4x8 in-register transposes and
//     ds_write_b64 instead of 2-byte scatter writes;
//   * MFMA operands are swapped (D = B.A^T) so a lane holds 4 consecutive output columns; bf16 outputs go through
//     LDS and leave as full 128-byte row segments; bias / ReLU / dropout are applied in the epilogue;
//   * split-K (weight gradients) writes per-split partial tiles to a workspace, splitk_reduce_kernel adds them into C;
//   * XCD-aware tile order (block id % 8 = XCD): an XCD walks the N tiles of an M panel / all tiles of one K split;
//   * gemm256_nt_kernel: 256x256 macro tile with a two-stage DMA pipeline for deep-K products (see its comment).
#include "common.h"
#include <stdlib.h>

// 256 bytes of zeros in GLOBAL memory (allocated once per process): target of out-of-range tile loads.  A __device__
// constant would make the selected pointer generic and turn the tile loads into flat loads.  Shared with gru_step.hip.
__attribute__((visibility("hidden"))) const void* gtos_zero_block() {
    static void* z = nullptr;
    if (!z) {
        if (hipMalloc(&z, 256) != hipSuccess) return nullptr;
        if (hipMemset(z, 0, 256) != hipSuccess) return nullptr;
    }
    return z;
}

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 128;                       // bytes per LDS row (BK elements)

template <typename T> struct GemmCfg;
template <> struct GemmCfg<bf16_t> { static constexpr int BK = 64, VEC = 8; };
template <> struct GemmCfg<float>  { static constexpr int BK = 32, VEC = 4; };

// Workgroup shapes: 256 threads = 4 waves (2x2, 64x64 per wave), 512 threads = 8 waves (2x4, 64x32 per wave).
template <int NTH> struct TCfg {
    static constexpr int NT = NTH, WAVES_N = NTH / 128, WTN = BN / WAVES_N, NTW = WTN / 16;
    static constexpr int ITERS = 1024 / NTH;    // 16-byte vectors per thread per operand tile (128 rows x 128 B / NTH / 16)
};

inline const void* zero_block() { return gtos_zero_block(); }

struct GemmArgs {
    const void* A; const void* B; void* C; const float* bias;
    int M, N, K; int64_t lda, ldb, ldc;
    int vecA, vecB, vecC;     // 16-byte vector access allowed (alignment checked on the host)
    int relu, accumulate, splitk;
    float p_drop; uint64_t seed;
    const void* zeros;        // 16 zero bytes in global memory
    float* ws;                // split-K partial tiles [splitk, M, N] (fp32), or null: fp32 atomics straight into C
};

// f(row) = ((row>>1)&7) ^ ((row>>4)&3).  ds_read_b128 is serviced in 16-lane groups that pair the rows {0-3,12-15} of one
// 16-byte chunk with the rows {4-11} of the next one: (row>>1) makes those 16 accesses hit 16 distinct 16-byte slots of
// the 256-byte bank row; the (row>>4) term spreads the 8-row-strided transposed ds_write_b64s (rows rc*8+i) over all chunks.
__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 3); }
// byte offset of 16-byte chunk c of row r inside an operand tile
__device__ __forceinline__ int lds_off(int r, int c) { return r * ROWB + ((c ^ swz(r)) << 4); }

// ---- global -> register tile load.  KC=true: operand rows are K-contiguous ([rows, K], ld): thread vector v covers
//      row v/8, chunk v%8.  KC=false: operand stored [K, rows] (rows contiguous): thread t covers 4 consecutive k
//      and one chunk of VEC consecutive rows (to be transposed on the LDS write).
template <typename T, bool KC, bool FAST, int NTH>
__device__ __forceinline__ void load_tile(const T* __restrict__ base, const U128* __restrict__ zeros, int64_t ld, int rows_total,
                                          int row0, int k0, int kend, U128 (&regs)[TCfg<NTH>::ITERS]) {
    constexpr int VEC = GemmCfg<T>::VEC, NT = NTH, ITERS = TCfg<NTH>::ITERS;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        int r, k;
        if (KC) { const int v = it * NT + threadIdx.x; r = row0 + (v >> 3); k = k0 + (v & 7) * VEC; }
        else {  // BM/VEC row chunks x (NT*VEC/BM) k groups; each thread ITERS consecutive k of one row chunk
            const int rc = threadIdx.x % (BM / VEC), kq = threadIdx.x / (BM / VEC);
            r = row0 + rc * VEC; k = k0 + kq * ITERS + it;
        }
        if constexpr (FAST) {
            // out-of-range vectors read a 16-byte block of zeros: the address is selected BEFORE the load, nothing is
            // selected after it, so no wait is forced between the loads and the MFMA phase that hides their latency
            const bool ok = KC ? (r < rows_total && k + VEC <= kend) : (k < kend && r + VEC <= rows_total);
            const int64_t off = KC ? ((int64_t)r * ld + k) : ((int64_t)k * ld + r);
            const U128* p = ok ? reinterpret_cast<const U128*>(base + off) : zeros;
            regs[it] = *p;
        } else {
            U128 val = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const bool ok = KC ? (r < rows_total && k + i < kend) : (k < kend && r + i < rows_total);
                uint32_t bits = 0u;
                if (ok) {
                    const T x = KC ? base[(int64_t)r * ld + k + i] : base[(int64_t)k * ld + r + i];
                    if constexpr (sizeof(T) == 2) bits = (uint32_t)x; else bits = __float_as_uint(x);
                }
                if constexpr (sizeof(T) == 2) val[i >> 1] |= bits << (16 * (i & 1)); else val[i] = bits;
            }
            regs[it] = val;
        }
    }
}

// ---- global -> LDS directly (K-contiguous operands, fast path): global_load_lds_dwordx4 writes wave-uniform base +
//      lane*16, i.e. 8 whole 128-byte rows per wave-instruction; the chunk swizzle is applied on the per-lane SOURCE
//      address (lane p of a row fetches logical chunk p ^ f(row)).  No staging VGPRs, no ds_write issue slots.
template <typename T, int NTH>
__device__ __forceinline__ void glds_tile(const T* __restrict__ base, const U128* __restrict__ zeros, int64_t ld, int rows_total,
                                          int row0, int k0, int kend, char* lds_tile) {
    constexpr int VEC = GemmCfg<T>::VEC, NT = NTH, ITERS = TCfg<NTH>::ITERS;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int blk = it * (NT / 64) + wave;                    // 1 KB block = rows blk*8 .. blk*8+7
        const int rl = blk * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz(rl);
        const int r = row0 + rl, k = k0 + c * VEC;
        const bool ok = r < rows_total && k + VEC <= kend;
        const void* src = ok ? static_cast<const void*>(base + (int64_t)r * ld + k) : static_cast<const void*>(zeros);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds_tile + blk * 1024), 16, 0, 0);
    }
}

// ---- M/N-contiguous bf16 operands ([K, rows] in memory: dY and X of dW = dY^T X, W of dX = dY W) also go global -> LDS
//      directly, as a [BK k][128 rows] tile (256-byte k-rows), and the MFMA fragments (8 consecutive k of one row) are
//      fetched with the hardware transpose read ds_read_b64_tr_b16: within a 16-lane group lane i supplies the address of
//      4 contiguous bf16 (row i/4, cols (i%4)*4..) of a 4x16 block and lane l receives column l&15 -- measured on gfx950
//      with tools/probes/tr_b16_probe.hip.  32-byte granules are XOR-swizzled with (k & 3) so the 4 k-rows of one read
//      (256 B apart = same banks) land on different banks; the swizzle is applied on the DMA's source address.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

template <typename T, int NTH>
__device__ __forceinline__ void glds_tile_km(const T* __restrict__ base, const U128* __restrict__ zeros, int64_t ld, int rows_total,
                                             int row0, int k0, int kend, char* lds_tile) {
    static_assert(sizeof(T) == 2, "transpose-read path is bf16 only");
    constexpr int NT = NTH, ITERS = TCfg<NTH>::ITERS;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int blk = it * (NT / 64) + wave;                    // 1 KB block = k-rows blk*4 .. blk*4+3
        const int kr = lane >> 4, pp = lane & 15;                 // k-row inside the block, physical 16-byte piece
        const int lp = ((((pp >> 1) ^ kr) & 7) << 1) | (pp & 1);  // logical piece (8 rows) stored at this physical slot
        const int k = k0 + blk * 4 + kr, r = row0 + lp * 8;
        const bool ok = k < kend && r + 8 <= rows_total;
        const void* src = ok ? static_cast<const void*>(base + (int64_t)k * ld + r) : static_cast<const void*>(zeros);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds_tile + blk * 1024), 16, 0, 0);
    }
}

// fragment (8 consecutive k = ks*32 + fq*8 .. +7) of row/column c0 + fr from a [k][128] tile
__device__ __forceinline__ bf16x8_t tr_fragment(const char* tile, int c0, int ks, int fr, int fq) {
    const int kq = fr >> 2;                                       // this lane addresses k-row kbase + kq (== k & 3)
    const int col = (((c0 >> 4) ^ kq) << 5) + (fr & 3) * 8;       // swizzled 32-byte granule + 8-byte piece
    const char* p0 = tile + (ks * 32 + fq * 8 + kq) * 256 + col;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 4 * 256));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

// ---- register -> LDS.  `lds` is the byte base of this operand's tile in the target stage.
template <typename T, bool KC, int NTH>
__device__ __forceinline__ void store_tile(char* __restrict__ lds, const U128 (&regs)[TCfg<NTH>::ITERS]) {
    constexpr int VEC = GemmCfg<T>::VEC, NT = NTH, ITERS = TCfg<NTH>::ITERS;
    if (KC) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int v = it * NT + threadIdx.x;
            *reinterpret_cast<U128*>(lds + lds_off(v >> 3, v & 7)) = regs[it];
        }
    } else {
        const int rc = threadIdx.x % (BM / VEC), kq = threadIdx.x / (BM / VEC);
        const int kk = kq * ITERS;                               // first k of this thread inside the tile
        if constexpr (sizeof(T) == 2) {
            // regs[j] = 8 rows (m) at k = kk + j.  Transpose ITERS(k) x 8(m): row m gets its ITERS k values
            // (ITERS*2 bytes) at chunk kk/8, byte (kk%8)*2.
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int w = i >> 1;
                char* dst = lds + lds_off(rc * 8 + i, kk >> 3) + (kk & 7) * 2;
                uint32_t lo, hi = 0u;
                if (i & 1) lo = (regs[0][w] >> 16) | (regs[1][w] & 0xffff0000u);
                else       lo = (regs[0][w] & 0xffffu) | (regs[1][w] << 16);
                if constexpr (ITERS == 4) {
                    if (i & 1) hi = (regs[2][w] >> 16) | (regs[3][w] & 0xffff0000u);
                    else       hi = (regs[2][w] & 0xffffu) | (regs[3][w] << 16);
                    *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
                } else {
                    *reinterpret_cast<uint32_t*>(dst) = lo;
                }
            }
        } else {
            // fp32: regs[j] = 4 rows at k = kk + j; row m gets ITERS k values at chunk kk/4, byte (kk%4)*4
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                char* dst = lds + lds_off(rc * 4 + i, kk >> 2) + (kk & 3) * 4;
                if constexpr (ITERS == 4) {
                    const U128 o = {regs[0][i], regs[1][i], regs[2][i], regs[3][i]};
                    *reinterpret_cast<U128*>(dst) = o;
                } else {
                    *reinterpret_cast<uint2*>(dst) = make_uint2(regs[0][i], regs[1][i]);
                }
            }
        }
    }
}

// NTH/STAGES: 512 threads + 2 LDS stages (register-staged or mixed operands), or 256 threads + ONE stage when both
// operands go global->LDS directly (forward Linear): 32 KB of LDS and ~110 VGPRs let 4 workgroups share a CU, whose
// interleaving hides the load latency, and a 64x64 wave tile needs a third fewer LDS fragment reads per MFMA.
template <typename T, typename TO, bool TA, bool TB, bool FAST, int NTH, int STAGES>
__global__ __launch_bounds__(NTH, 4) void gemm_kernel(GemmArgs a) {
    if (a.p_drop > 0.f) a.seed = live_seed(a.seed);
    constexpr int BK = GemmCfg<T>::BK;
    constexpr int NT = NTH, WAVES_N = TCfg<NTH>::WAVES_N, WTN = TCfg<NTH>::WTN, NTW = TCfg<NTH>::NTW, ITERS = TCfg<NTH>::ITERS;
    constexpr int STAGE = (BM + BN) * ROWB;                       // 32 KB
    constexpr bool BF = sizeof(T) == 2;
    constexpr bool GA = FAST && (!TA || BF), GB = FAST && (TB || BF);   // operand goes global -> LDS by DMA
    constexpr bool TRA = GA && TA, TRB = GB && !TB;                      // ... as a [k][rows] tile read with ds_read_b64_tr_b16
    __shared__ __attribute__((aligned(16))) char lds[STAGES * STAGE];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave / WAVES_N) * 64, wn = (wave % WAVES_N) * WTN;
    // XCD-aware tile order.  The dispatcher deals block ids round-robin to the 8 XCDs (private L2 each), so give every
    // XCD its own M panels and let consecutive blocks of one XCD walk the N tiles of one panel: the streamed operand
    // (A rows: activations, M up to millions) is then fetched from HBM once instead of once per N tile.
    const int nN = (a.N + BN - 1) / BN;
    int m0, n0, bz;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    if (a.splitk > 1) {
        // weight-gradient shape (small MxN, huge K): all output tiles of ONE K split run together on one XCD, so the
        // K-slice rows of both operands are fetched from HBM once and shared through that XCD's L2
        const int nM = (a.M + BM - 1) / BM, tiles = nM * nN;
        bz = (sq / tiles) * 8 + xcd;
        const int t = sq % tiles;
        m0 = (t / nN) * BM; n0 = (t % nN) * BN;
        if (bz >= a.splitk) return;
    } else {
        bz = 0;
        m0 = ((sq / nN) * 8 + xcd) * BM; n0 = (sq % nN) * BN;
        if (m0 >= a.M) return;
    }
    // split-K range (whole BK tiles per split)
    const int ktiles = (a.K + BK - 1) / BK;
    const int tps = (ktiles + a.splitk - 1) / a.splitk;
    const int kbeg = bz * tps * BK;
    const int kend = min(a.K, kbeg + tps * BK);
    if (kbeg >= kend && a.splitk > 1) return;

    const T* A = static_cast<const T*>(a.A);
    const T* B = static_cast<const T*>(a.B);
    const U128* Z = static_cast<const U128*>(a.zeros);

    f32x4_t acc[4][NTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // Software pipeline, prefetch distance 2: while tile t is multiplied out of LDS stage `cur`, tile t+1 (loaded one
    // step earlier) waits in one register set and tile t+2 is being issued into the other.  The step is written out
    // twice so the two register sets are named, never indexed.  Only the OLDER set is waited for at the end of a step
    // (s_waitcnt vmcnt(8) leaves the 8 newest loads in flight), so HBM latency has two steps of MFMAs to hide under.
    const int fr = lane & 15, fq = lane >> 4;
    auto dma_a = [&](int k0x, char* dst) {
        if constexpr (TRA) glds_tile_km<T, NTH>(A, Z, a.lda, a.M, m0, k0x, kend, dst);
        else glds_tile<T, NTH>(A, Z, a.lda, a.M, m0, k0x, kend, dst);
    };
    auto dma_b = [&](int k0x, char* dst) {
        if constexpr (TRB) glds_tile_km<T, NTH>(B, Z, a.ldb, a.N, n0, k0x, kend, dst);
        else glds_tile<T, NTH>(B, Z, a.ldb, a.N, n0, k0x, kend, dst);
    };
    auto mma_stage = [&](const char* As, const char* Bs) {
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {               // BK = 64 = 2 MFMA k-steps of 32; lane chunk = ks*4 + fq
                bf16x8_t fa[4], fb[NTW];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if constexpr (TRA) fa[t] = tr_fragment(As, wm + t * 16, ks, fr, fq);
                    else fa[t] = *reinterpret_cast<const bf16x8_t*>(As + lds_off(wm + t * 16 + fr, ks * 4 + fq));
                }
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    if constexpr (TRB) fb[t] = tr_fragment(Bs, wn + t * 16, ks, fr, fq);
                    else fb[t] = *reinterpret_cast<const bf16x8_t*>(Bs + lds_off(wn + t * 16 + fr, ks * 4 + fq));
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)   // swapped operands: rows of D <-> n, cols <-> m
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {               // BK = 32 = 8 MFMA k-steps of 4; element k = ks*4 + fq
                float fa[4], fb[NTW];
#pragma unroll
                for (int t = 0; t < 4; ++t) fa[t] = *reinterpret_cast<const float*>(As + lds_off(wm + t * 16 + fr, ks) + fq * 4);
#pragma unroll
                for (int t = 0; t < NTW; ++t) fb[t] = *reinterpret_cast<const float*>(Bs + lds_off(wn + t * 16 + fr, ks) + fq * 4);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
            }
        }
    };

    if constexpr (STAGES == 1) {
        // both operands by global_load_lds into the single stage; the other resident workgroups cover the wait
        static_assert(!(STAGES == 1) || (GA && GB), "single-stage variant needs both operands by DMA");
        if constexpr (!TRA && !TRB) {
            // Forward shape (both operands K-contiguous): the per-lane part of every DMA source address is computed ONCE
            // (counters: the generic helpers spend ~180 VALU instructions per k tile and wave on it, more issue cycles than
            // the tile's 32 MFMAs).  Source = uniform tile base + k0 + 32-bit lane offset; rows past the end re-read the
            // last valid row (their products land in rows / columns that are never stored); only a partial last k tile
            // selects the block of zeros per lane.
            const char* Ab = reinterpret_cast<const char*>(A + (int64_t)m0 * a.lda);
            const char* Bb = reinterpret_cast<const char*>(B + (int64_t)n0 * a.ldb);
            const int amax = a.M - 1 - m0, bmax = a.N - 1 - n0;
            const uint32_t lda2 = (uint32_t)a.lda * (uint32_t)sizeof(T), ldb2 = (uint32_t)a.ldb * (uint32_t)sizeof(T);
            const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            uint32_t offA[ITERS], offB[ITERS];
            int cc[ITERS];
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int rl = (it * (NT / 64) + wv) * 8 + (lane >> 3);
                cc[it] = (lane & 7) ^ swz(rl);
                offA[it] = (uint32_t)min(rl, amax) * lda2 + (uint32_t)(cc[it] << 4);
                offB[it] = (uint32_t)min(rl, bmax) * ldb2 + (uint32_t)(cc[it] << 4);
            }
            constexpr int VEC = GemmCfg<T>::VEC;
            for (int k0 = kbeg; k0 < kend; k0 += BK) {
                const char* ak = Ab + (int64_t)k0 * (int)sizeof(T);
                const char* bk = Bb + (int64_t)k0 * (int)sizeof(T);
                if (k0 + BK <= kend) {
#pragma unroll
                    for (int it = 0; it < ITERS; ++it) {
                        const int blk = it * (NT / 64) + wv;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ak + offA[it]),
                                                         (__attribute__((address_space(3))) void*)(lds + blk * 1024), 16, 0, 0);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bk + offB[it]),
                                                         (__attribute__((address_space(3))) void*)(lds + BM * ROWB + blk * 1024), 16, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < ITERS; ++it) {
                        const int blk = it * (NT / 64) + wv;
                        const bool ok = k0 + cc[it] * VEC + VEC <= kend;
                        const void* sa = ok ? static_cast<const void*>(ak + offA[it]) : static_cast<const void*>(Z);
                        const void* sb = ok ? static_cast<const void*>(bk + offB[it]) : static_cast<const void*>(Z);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sa,
                                                         (__attribute__((address_space(3))) void*)(lds + blk * 1024), 16, 0, 0);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sb,
                                                         (__attribute__((address_space(3))) void*)(lds + BM * ROWB + blk * 1024), 16, 0, 0);
                    }
                }
                __syncthreads();                           // hipcc drains the DMA (vmcnt(0)) in front of the barrier
                mma_stage(lds, lds + BM * ROWB);
                __syncthreads();                           // every wave is done reading before the next tile lands
            }
        } else {
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            dma_a(k0, lds);
            dma_b(k0, lds + BM * ROWB);
            __syncthreads();                               // hipcc drains the DMA (vmcnt(0)) in front of the barrier
            mma_stage(lds, lds + BM * ROWB);
            __syncthreads();                               // every wave is done reading before the next tile lands
        }
        }
    } else {
    // K-contiguous operands on the fast path go global -> LDS directly (prefetch distance 1: the barrier drains the
    // DMA); the others through the two register sets (distance 2).
    U128 ra0[ITERS], rb0[ITERS], ra1[ITERS], rb1[ITERS];
    if constexpr (GA) dma_a(kbeg, lds);
    else { load_tile<T, !TA, FAST, NTH>(A, Z, a.lda, a.M, m0, kbeg, kend, ra0); store_tile<T, !TA, NTH>(lds, ra0);
           load_tile<T, !TA, FAST, NTH>(A, Z, a.lda, a.M, m0, kbeg + BK, kend, ra0); }
    if constexpr (GB) dma_b(kbeg, lds + BM * ROWB);
    else { load_tile<T, TB, FAST, NTH>(B, Z, a.ldb, a.N, n0, kbeg, kend, rb0); store_tile<T, TB, NTH>(lds + BM * ROWB, rb0);
           load_tile<T, TB, FAST, NTH>(B, Z, a.ldb, a.N, n0, kbeg + BK, kend, rb0); }
    __syncthreads();

    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += 2 * BK) {
        {   // even step: tile k0 in stage cur, tile k0+BK in flight in set 0, issue tile k0+2BK into set 1
            const char* As = lds + cur * STAGE;
            char* An = lds + (cur ^ 1) * STAGE;
            if constexpr (GA) dma_a(k0 + BK, An);
            else load_tile<T, !TA, FAST, NTH>(A, Z, a.lda, a.M, m0, k0 + 2 * BK, kend, ra1);
            if constexpr (GB) dma_b(k0 + BK, An + BM * ROWB);
            else load_tile<T, TB, FAST, NTH>(B, Z, a.ldb, a.N, n0, k0 + 2 * BK, kend, rb1);
            __builtin_amdgcn_sched_barrier(0);             // keep the loads ahead of the MFMA block (hipcc sinks them)
            mma_stage(As, As + BM * ROWB);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!GA) store_tile<T, !TA, NTH>(An, ra0); // write late: the other stage is free since the last barrier
            if constexpr (!GB) store_tile<T, TB, NTH>(An + BM * ROWB, rb0);
            __syncthreads();
            cur ^= 1;
        }
        if (k0 + BK >= kend) break;
        {   // odd step: roles of the register sets swapped
            const char* As = lds + cur * STAGE;
            char* An = lds + (cur ^ 1) * STAGE;
            if constexpr (GA) dma_a(k0 + 2 * BK, An);
            else load_tile<T, !TA, FAST, NTH>(A, Z, a.lda, a.M, m0, k0 + 3 * BK, kend, ra0);
            if constexpr (GB) dma_b(k0 + 2 * BK, An + BM * ROWB);
            else load_tile<T, TB, FAST, NTH>(B, Z, a.ldb, a.N, n0, k0 + 3 * BK, kend, rb0);
            __builtin_amdgcn_sched_barrier(0);
            mma_stage(As, As + BM * ROWB);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!GA) store_tile<T, !TA, NTH>(An, ra1);
            if constexpr (!GB) store_tile<T, TB, NTH>(An + BM * ROWB, rb1);
            __syncthreads();
            cur ^= 1;
        }
    }

    }

    // ---- epilogue: lane holds C[m = .. + fr][n = .. + fq*4 + 0..3]
    TO* C = static_cast<TO*>(a.C);
    const float keep_scale = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;

    if constexpr (sizeof(TO) == 2) {
        if (a.vecC && a.splitk == 1) {
            // bf16 output: bias/act in registers, 64x64 wave tile -> LDS (8-byte writes) -> 16-byte row-contiguous stores
            constexpr int CP = WTN * 2 + 16;                           // bytes per staged row (WTN bf16 + pad)
            constexpr int LPR = WTN / 8, RPP = 64 / LPR;               // lanes per row, rows per pass
            char* cs = lds + wave * 32 * CP;                           // private to the wave; 32 rows at a time so that
            //                                                            4 waves x 32 x 144 B fit the one-stage 32 KB
            if (!a.bias && !a.relu && !(a.p_drop > 0.f) && !a.accumulate && m0 + BM <= a.M && n0 + BN <= a.N) {
                // interior tile of a plain product (the relation projections): no per-element conditions, addresses
                // computed once, wave-level ordering only (the staging rows are private to the wave)
                const int rrow = lane / LPR, rcol = (lane % LPR) * 8;
                char* wr = cs + fr * CP + fq * 8;
                const char* rd = cs + rrow * CP + rcol * 2;
                bf16_t* cp0 = reinterpret_cast<bf16_t*>(C) + (int64_t)(m0 + wm + rrow) * a.ldc + n0 + wn + rcol;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt) {
                            const f32x4_t v = acc[half * 2 + mh][nt];
                            *reinterpret_cast<uint2*>(wr + mh * 16 * CP + nt * 32) = make_uint2(pack_bf(v[0], v[1]), pack_bf(v[2], v[3]));
                        }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0)
#pragma unroll
                    for (int pass = 0; pass < 32 / RPP; ++pass) {
                        const uint4 val = *reinterpret_cast<const uint4*>(rd + pass * RPP * CP);
                        *reinterpret_cast<uint4*>(cp0 + (int64_t)(half * 32 + pass * RPP) * a.ldc) = val;
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);                // read back before the next half overwrites the rows
                }
                return;
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (half) __syncthreads();                             // the first half has been read back
#pragma unroll
                for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        const int mt = half * 2 + mh;
                        const int m = m0 + wm + mt * 16 + fr, n = n0 + wn + nt * 16 + fq * 4;
                        float v[4] = {acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (a.bias && n + i < a.N) v[i] += a.bias[n + i];
                            if (a.relu) v[i] = fmaxf(v[i], 0.f);
                            if (a.p_drop > 0.f)
                                v[i] = drop_keep(a.seed, (uint64_t)m * (uint64_t)a.N + (uint64_t)(n + i), a.p_drop) ? v[i] * keep_scale : 0.f;
                        }
                        *reinterpret_cast<uint2*>(cs + (mh * 16 + fr) * CP + (nt * 16 + fq * 4) * 2) =
                            make_uint2(pack_bf(v[0], v[1]), pack_bf(v[2], v[3]));
                    }
                __syncthreads();
#pragma unroll
                for (int pass = 0; pass < 32 / RPP; ++pass) {
                    const int row = pass * RPP + lane / LPR, col = (lane % LPR) * 8;
                    const int m = m0 + wm + half * 32 + row, n = n0 + wn + col;
                    if (m >= a.M || n >= a.N) continue;
                    uint4 val = *reinterpret_cast<const uint4*>(cs + row * CP + col * 2);
                    bf16_t* cp = reinterpret_cast<bf16_t*>(C) + (int64_t)m * a.ldc + n;
                    if (n + 8 <= a.N) {
                        if (a.accumulate) {
                            const uint4 old = *reinterpret_cast<const uint4*>(cp);
                            val = make_uint4(pack_bf(lo_bf(val.x) + lo_bf(old.x), hi_bf(val.x) + hi_bf(old.x)),
                                             pack_bf(lo_bf(val.y) + lo_bf(old.y), hi_bf(val.y) + hi_bf(old.y)),
                                             pack_bf(lo_bf(val.z) + lo_bf(old.z), hi_bf(val.z) + hi_bf(old.z)),
                                             pack_bf(lo_bf(val.w) + lo_bf(old.w), hi_bf(val.w) + hi_bf(old.w)));
                        }
                        *reinterpret_cast<uint4*>(cp) = val;
                    } else {
                        const bf16_t* e = reinterpret_cast<const bf16_t*>(cs + row * CP + col * 2);
                        for (int i = 0; i < 8 && n + i < a.N; ++i) {
                            float o = bf2f(e[i]);
                            if (a.accumulate) o += bf2f(cp[i]);
                            cp[i] = f2bf(o);
                        }
                    }
                }
            }
            return;
        }
    }

#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm + mt * 16 + fr;
        if (m >= a.M) continue;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int n = n0 + wn + nt * 16 + fq * 4;
            if (n >= a.N) continue;
            float v[4] = {acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]};
            TO* cp = C + (int64_t)m * a.ldc + n;
            if (a.splitk > 1) {
                if constexpr (sizeof(TO) == 4) {
                    if (a.ws) {     // N % 4 == 0 (host): plain 16-byte stores of this split's partial tile, reduced afterwards
                        *reinterpret_cast<float4*>(a.ws + ((int64_t)bz * a.M + m) * a.N + n) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (n + i < a.N) atomicAdd(reinterpret_cast<float*>(cp) + i, v[i]);
                    }
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (a.bias && n + i < a.N) v[i] += a.bias[n + i];
                if (a.relu) v[i] = fmaxf(v[i], 0.f);
                if (a.p_drop > 0.f)
                    v[i] = drop_keep(a.seed, (uint64_t)m * (uint64_t)a.N + (uint64_t)(n + i), a.p_drop) ? v[i] * keep_scale : 0.f;
            }
            if (sizeof(TO) == 4 && a.vecC && n + 3 < a.N) {
                float4 o = make_float4(v[0], v[1], v[2], v[3]);
                if (a.accumulate) { const float4 c = *reinterpret_cast<const float4*>(cp); o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
                *reinterpret_cast<float4*>(cp) = o;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (n + i < a.N) {
                    float o = v[i];
                    if (a.accumulate) o += to_f<TO>(cp[i]);
                    cp[i] = from_f<TO>(o);
                }
            }
        }
    }
}

// C[m,n] += sum over the K splits of the partial tiles (deterministic alternative to the atomic epilogue: device-scope
// fp32 atomics from 8 XCDs resolve at the memory side and cost more than the GEMM itself when K per split is short)
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int M, int N, float* __restrict__ C, int64_t ldc) {
    const int64_t n4 = N / 4, total = (int64_t)M * n4, plane = (int64_t)M * N;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(t / n4), n = (int)(t % n4) * 4;
        const float* p = ws + (int64_t)m * N + n;
        float4 acc = *reinterpret_cast<const float4*>(p);
        for (int s = 1; s < splits; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(p + s * plane);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        float* cp = C + (int64_t)m * ldc + n;
        cp[0] += acc.x; cp[1] += acc.y; cp[2] += acc.z; cp[3] += acc.w;
    }
}


// ---- 256x256 macro tile for big K-contiguous bf16 products with a DEEP reduction (K >= 1024).  Measured on MI355X
//      (tools/bench_gemm.py): [434624,4096]x[1024,4096]^T 1001 TF/s vs 832 for the 128x128 kernel, 8192^3 1004 vs 705;
//      at K = 512 both sit at ~690 TF/s because a tile's output write (not overlapped inside a 1-workgroup-per-CU
//      kernel) costs as much as its 8 k tiles, so the dispatcher keeps K < 2048 on the 128x128 kernel (at K = 1024, N = 512 the macro tile measured 603 vs 653 TF/s in the training step).  8 waves (2 x 4), each 128 x 64 = 8 x 4 MFMA
//      tiles (128 accumulator registers), two 64 KB LDS stages.  Per k tile: barrier (drains this stage's DMA), ALL
//      fragments of the tile are read into registers, THEN the next tile's DMA is issued into the other stage, then the
//      64 MFMAs run while it lands -- hipcc waits for every outstanding LDS-DMA in front of any ds_read, so the reads
//      have to precede the prefetch for the two to overlap.  One barrier per k tile.
constexpr int BM2 = 256, BN2 = 256, STAGE2 = (BM2 + BN2) * ROWB;
const bool g_use256 = !(getenv("GTOS_GEMM256") && getenv("GTOS_GEMM256")[0] == '0');

__global__ __launch_bounds__(512, 2) void gemm256_nt_kernel(GemmArgs a) {
    if (a.p_drop > 0.f) a.seed = live_seed(a.seed);
    extern __shared__ __attribute__((aligned(16))) char lds2[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
    const int nN = (a.N + BN2 - 1) / BN2;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    const int m0 = ((sq / nN) * 8 + xcd) * BM2, n0 = (sq % nN) * BN2;
    if (m0 >= a.M) return;
    const bf16_t* A = static_cast<const bf16_t*>(a.A);
    const bf16_t* B = static_cast<const bf16_t*>(a.B);
    const U128* Z = static_cast<const U128*>(a.zeros);

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto dma = [&](int k0, char* st) {
#pragma unroll
        for (int op = 0; op < 2; ++op) {
            const bf16_t* base = op ? B : A;
            const int64_t ld = op ? a.ldb : a.lda;
            const int rows_total = op ? a.N : a.M, row0 = op ? n0 : m0;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int blk = it * 8 + wave, rl = blk * 8 + (lane >> 3);
                const int c = (lane & 7) ^ swz(rl);
                const int r = row0 + rl, k = k0 + c * 8;
                const bool ok = r < rows_total && k + 8 <= a.K;
                const void* src = ok ? static_cast<const void*>(base + (int64_t)r * ld + k) : static_cast<const void*>(Z);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(st + op * (BM2 * ROWB) + blk * 1024), 16, 0, 0);
            }
        }
    };

    const int nk = (a.K + 63) / 64;
    dma(0, lds2);
    for (int kt = 0; kt < nk; ++kt) {
        const char* As = lds2 + (kt & 1) * STAGE2;
        const char* Bs = As + BM2 * ROWB;
        __syncthreads();                                   // this stage has landed (hipcc drains vmcnt in front of it)
        bf16x8_t fa[2][8], fb[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int t = 0; t < 8; ++t) fa[ks][t] = *reinterpret_cast<const bf16x8_t*>(As + lds_off(wm + t * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int t = 0; t < 4; ++t) fb[ks][t] = *reinterpret_cast<const bf16x8_t*>(Bs + lds_off(wn + t * 16 + fr, ks * 4 + fq));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) dma((kt + 1) * 64, lds2 + ((kt + 1) & 1) * STAGE2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][nt], fa[ks][mt], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue (bf16 out): bias/act in registers, 32 rows x 64 columns of the wave tile at a time through LDS
    __syncthreads();
    bf16_t* C = static_cast<bf16_t*>(a.C);
    const float keep_scale = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
    constexpr int CP = 64 * 2 + 16;                        // bytes per staged row
    char* cs = lds2 + wave * 32 * CP;                      // private to the wave
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int mt = q4 * 2 + mh;
                const int m = m0 + wm + mt * 16 + fr, n = n0 + wn + nt * 16 + fq * 4;
                float v[4] = {acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (a.bias && n + i < a.N) v[i] += a.bias[n + i];
                    if (a.relu) v[i] = fmaxf(v[i], 0.f);
                    if (a.p_drop > 0.f)
                        v[i] = drop_keep(a.seed, (uint64_t)m * (uint64_t)a.N + (uint64_t)(n + i), a.p_drop) ? v[i] * keep_scale : 0.f;
                }
                *reinterpret_cast<uint2*>(cs + (mh * 16 + fr) * CP + (nt * 16 + fq * 4) * 2) = make_uint2(pack_bf(v[0], v[1]), pack_bf(v[2], v[3]));
            }
        // wave-private region: only this wave's own LDS writes have to be visible to its reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 8 + (lane >> 3), col = (lane & 7) * 8;
            const int m = m0 + wm + q4 * 32 + row, n = n0 + wn + col;
            if (m >= a.M || n >= a.N) continue;
            uint4 val = *reinterpret_cast<const uint4*>(cs + row * CP + col * 2);
            bf16_t* cp = C + (int64_t)m * a.ldc + n;
            if (n + 8 <= a.N) {
                if (a.accumulate) {
                    const uint4 old = *reinterpret_cast<const uint4*>(cp);
                    val = make_uint4(pack_bf(lo_bf(val.x) + lo_bf(old.x), hi_bf(val.x) + hi_bf(old.x)),
                                     pack_bf(lo_bf(val.y) + lo_bf(old.y), hi_bf(val.y) + hi_bf(old.y)),
                                     pack_bf(lo_bf(val.z) + lo_bf(old.z), hi_bf(val.z) + hi_bf(old.z)),
                                     pack_bf(lo_bf(val.w) + lo_bf(old.w), hi_bf(val.w) + hi_bf(old.w)));
                }
                *reinterpret_cast<uint4*>(cp) = val;
            } else {
                const bf16_t* e = reinterpret_cast<const bf16_t*>(cs + row * CP + col * 2);
                for (int i = 0; i < 8 && n + i < a.N; ++i) {
                    float o = bf2f(e[i]);
                    if (a.accumulate) o += bf2f(cp[i]);
                    cp[i] = f2bf(o);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                // the reads are done before the next quarter overwrites the rows
    }
}

// ---- Software-pipelined 256x256 kernels (gemm256q_nt_kernel below, gemm256p_tn_kernel further down).  Two findings of round 2 make them possible:
//      * hipcc's wait-count pass tracks LDS-DMA per underlying LDS OBJECT: with the stages / half-tiles as SEPARATE static __shared__ arrays a
//        ds_read of one does not wait for the DMA in flight into the others (with ONE LDS array it drains vmcnt(0) in front of every ds_read --
//        what limits gemm256_nt_kernel to one tile of prefetch, issued after the fragment reads).  The waits are explicit counted vmcnt(n);
//        hipcc accepts them and adds none of its own (checked in the ISA).
//      * Waves 4-7 run one barrier behind waves 0-3: on every SIMD one wave multiplies while its partner issues DMA pieces (60-180 issue
//        cycles each) and fragment reads.
//      Round 2's NT kernel of this family -- gemm256p_nt_kernel: four 32-k stages of 64-byte rows, 32 MFMAs per compute segment, 1.06-1.18 PF/s on
//      deep K -- was replaced by the 8-phase kernel in round 6 (+10-16 %: whole-line DMA pieces, shorter load segments; K % 64 == 32 through a
//      zero-filled half of the last k tile) and removed; the TN kernel keeps the four-stage form (its stage rows are 512 bytes: whole lines already).
constexpr int ROW3 = 64, A3 = BM2 * ROW3, ST3 = (BM2 + BN2) * ROW3;      // the TN kernel's stage: [32 k][256 m] + [32 k][256 n]
constexpr int g_min_k256 = 2048;                     // (rounds 3-5 swept these thresholds with environment switches: profiles/r3..r5_ab_switches.txt)
const bool g_use_pipe = !(getenv("GTOS_GEMM_PIPE") && getenv("GTOS_GEMM_PIPE")[0] == '0');
const bool g_use_pipe_tn = !(getenv("GTOS_GEMM_PIPE_TN") && getenv("GTOS_GEMM_PIPE_TN")[0] == '0');
constexpr int g_min_kpipe = 1024;
// (Small products -- a few thousand rows: the graph layers' and the decoder's projections -- on a 256x256 kernel: 17-25 us against 10-19 us on the
// single-stage 128x128 kernel, step +1.1 ms; measured in round 3, never enabled.)

#define GTOS_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))

// (Round 4's two opt-in experiments lived here: gemm_p2_nt_kernel -- two role-split 4-wave workgroups per CU on 128x256 tiles for
//  K <= 1023 -- and gemm_w4_nt_kernel -- one wave per SIMD with a 256-register accumulator named by hand.  Both were bit-checked on the
//  GPU and measured at or below the default kernels (profiles/r4c_gemm_p2_*.txt, r4d_*, r4w_gemm_w4.txt; DESIGN.md section 0 keeps the
//  findings: a 1 KB LDS-DMA piece costs its wave ~60 issue cycles whether it hits or not, whole 128-byte rows per stage matter).  They
//  are not part of the library any more; the last commit that carries them is 4bd4edb.)

// ---- gemm256q_nt_kernel (round 6): the same 256x256 tile / 8 waves (2 x 4, 128 x 64 each) on 64-k tiles cut into HALF-TILES, eight
//      phases per two k tiles.  Against round 2's four-stage kernel (gemm256p_nt_kernel, removed): (1) a DMA piece is 8 rows x 128 bytes -- whole cache lines -- where the 32-k
//      stages fetch 16 rows x 64 bytes (half lines: twice the line requests on the CU's vector-memory path for the same bytes; what the
//      forward GRU step gained by moving to 64-k stages, and what profiles/r6p_gru_bwd_regfed.txt lost by fetching half lines); (2) a
//      compute segment is 16 MFMAs on one 64 x 32 quadrant of the wave tile, its fragment reads 4..12 ds_read_b128 (quadrant order
//      (0,0) (0,1) (1,1) (1,0): one operand's fragments stay in registers from phase to phase), so the partner wave of a SIMD loads in
//      shorter, more even segments.
//      Measured (round 6, call r6q, same box, old -> new; hipBLASLt beside): [434624,4096]x[1024,4096]^T 1061 -> 1158 TF/s (1168);
//      [434624,8192]x[512,8192]^T 1130 -> 1265-1290 (1335); 8192^3 1100 (two-stage kernel) -> 1265-1335 (1285-1325); K = 1024 820 -> 850-875
//      (880-940); at K = 512 713 against 727-760 for the 128x128 kernel (four workgroups per CU overlap their output writes): the
//      threshold stays at K >= 1024.  s_setprio around the MFMAs and a lookahead of five half-tiles measured the same and are not kept.
//      * Half-tiles: A-half h = rows {0..63, 128..191} + 64 h of the tile (the rows quadrant row h of BOTH wave rows needs), B-half h =
//        columns {0..31, 64..95, 128..159, 192..223} + 32 h; 128 rows x 128 bytes = 16 KB each, 8 separate LDS objects [k tile parity][A0 B0
//        B1 A1] (hipcc tracks LDS-DMA per object).  Load order per k tile A0 B0 B1 A1 = the order the phases first read them.
//      * One half-tile (2 pieces per wave) is issued per phase, four half-tiles ahead (H(g + 4) in phase g): a buffer is refilled at
//        least two phases after its last read.  Before a phase's first barrier a wave waits vmcnt(4): everything up to H(g + 2) -- what
//        phase g + 1 reads -- has landed; the barrier makes that true for every wave ONE PHASE BEFORE the read.
//      * Waves 4-7 run one barrier behind waves 0-3 (the two waves of a SIMD alternate between loading and multiplying).
//      * LDS rows are 128 bytes; chunk c of row r sits at c ^ ((r >> 1) & 7) (conflict-free ds_read_b128, see swz; the (r >> 4) term of
//        swz is not needed without transposed writes): a per-lane constant, so fragment addresses are two lane offsets (ks = 0, 1) plus
//        immediates and the DMA sources a per-lane offset per piece.
//      * K % 32 == 0; with K % 64 == 32 the lanes that would fetch the upper four chunks of the last k tile fetch a block of zeros instead
//        ([434624,2016]x[1024,2016]^T: 838 TF/s on the four-stage kernel -> 970, hipBLASLt 980-1005).  Rows / columns past the end re-read the last
//        valid one; half-tiles past the end of K re-read the last k tile (never multiplied): the number of DMAs in flight is the same in every phase.
constexpr int HB = 128 * ROWB;                             // half-tile: 128 rows x 128 bytes
constexpr int HB_EPI = 4 * 32 * (64 * 2 + 16);             // the two buffers the epilogue stages its rows in

__global__ __launch_bounds__(512) void gemm256q_nt_kernel(GemmArgs a) {
    if (a.p_drop > 0.f) a.seed = live_seed(a.seed);
    __shared__ __attribute__((aligned(16))) char a00[HB_EPI];
    __shared__ __attribute__((aligned(16))) char a01[HB];
    __shared__ __attribute__((aligned(16))) char b00[HB_EPI];
    __shared__ __attribute__((aligned(16))) char b01[HB];
    __shared__ __attribute__((aligned(16))) char a10[HB];
    __shared__ __attribute__((aligned(16))) char a11[HB];
    __shared__ __attribute__((aligned(16))) char b10[HB];
    __shared__ __attribute__((aligned(16))) char b11[HB];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;
    const int wm = wr * 128, wn = wc * 64;
    const int nN = (a.N + BN2 - 1) / BN2;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    const int m0 = ((sq / nN) * 8 + xcd) * BM2, n0 = (sq % nN) * BN2;
    if (m0 >= a.M) return;
    const char* Ab = reinterpret_cast<const char*>(static_cast<const bf16_t*>(a.A) + (int64_t)m0 * a.lda);
    const char* Bb = reinterpret_cast<const char*>(static_cast<const bf16_t*>(a.B) + (int64_t)n0 * a.ldb);
    const int amax = a.M - 1 - m0, bmax = a.N - 1 - n0;
    const uint32_t lda2 = (uint32_t)a.lda * 2u, ldb2 = (uint32_t)a.ldb * 2u;
    const int nk = (a.K + 63) / 64;
    const bool tail32 = (a.K & 63) != 0;                   // K % 64 == 32: the upper four chunks of the last k tile come from the block of zeros
    const char* Zl = static_cast<const char*>(a.zeros) + (lane & 15) * 16;
    bool hi[2];
    // DMA: piece p (= wave, wave + 8) of a half-tile = buffer rows p*8 .. p*8+7; lane l -> row p*8 + (l >> 3), physical chunk l & 7
    uint32_t aoff[2][2], boff[2][2];                       // [half][piece]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rb = (wave + 8 * j) * 8 + (lane >> 3);                                   // buffer row
        const uint32_t ch = (uint32_t)(((lane & 7) ^ ((rb >> 1) & 7)) << 4);              // source chunk of this LDS position
        hi[j] = ch >= 64;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            aoff[h][j] = (uint32_t)min((rb >> 6) * 128 + h * 64 + (rb & 63), amax) * lda2 + ch;
            boff[h][j] = (uint32_t)min((rb >> 5) * 64 + h * 32 + (rb & 31), bmax) * ldb2 + ch;
        }
    }
    // fragment reads: buffer row (.. + fr), logical chunk ks*4 + fq
    const int fo0 = fr * ROWB + ((fq ^ ((fr >> 1) & 7)) << 4), fo1 = fr * ROWB + (((4 + fq) ^ ((fr >> 1) & 7)) << 4);
    const int fa_base = wr * 64 * ROWB, fb_base = wc * 32 * ROWB;

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t fa[4][2], fb[2][2];

#define GTOS_QDMA(src, dst)                                                                                                   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                                    \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
// half-tile j (0 A0, 1 B0, 2 B1, 3 A1) of k tile kt_ into buf
#define GTOS_QISSUE(buf, isA, h, kt_)                                                                                         \
    {                                                                                                                         \
        const int kb_ = min((kt_), nk - 1) * 128;                                                                             \
        const bool tl_ = tail32 && (kt_) >= nk - 1;                                                                           \
        const char* s0_ = ((isA) ? Ab : Bb) + kb_ + ((isA) ? aoff[h][0] : boff[h][0]);                                        \
        const char* s1_ = ((isA) ? Ab : Bb) + kb_ + ((isA) ? aoff[h][1] : boff[h][1]);                                        \
        GTOS_QDMA((tl_ && hi[0]) ? Zl : s0_, (buf) + wave * 1024);                                                            \
        GTOS_QDMA((tl_ && hi[1]) ? Zl : s1_, (buf) + (8 + wave) * 1024);                                                      \
    }
#define GTOS_QREAD_A(buf)                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                           \
        fa[i][0] = *reinterpret_cast<const bf16x8_t*>((buf) + fa_base + i * 16 * ROWB + fo0);                                 \
        fa[i][1] = *reinterpret_cast<const bf16x8_t*>((buf) + fa_base + i * 16 * ROWB + fo1);                                 \
    }
#define GTOS_QREAD_B(buf)                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                           \
        fb[j][0] = *reinterpret_cast<const bf16x8_t*>((buf) + fb_base + j * 16 * ROWB + fo0);                                 \
        fb[j][1] = *reinterpret_cast<const bf16x8_t*>((buf) + fb_base + j * 16 * ROWB + fo1);                                 \
    }
// one phase: READS (this phase's new fragments), the DMA of the half-tile four ahead, the counted wait, barrier, 16 MFMAs on
// quadrant (mh, nh), barrier
#define GTOS_QPHASE(READS, ISSUE, mh, nh)                                                                                     \
    {                                                                                                                         \
        READS;                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        ISSUE;                                                                                                                \
        GTOS_VMCNT(4);                                     /* this wave's pieces of everything the NEXT phase reads */        \
        __builtin_amdgcn_s_barrier();                                                                                         \
        __builtin_amdgcn_s_waitcnt(0xc07f);                /* lgkmcnt(0): this phase's fragments */                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                      \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                     \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
                    acc[(mh) * 4 + i][(nh) * 2 + j] =                                                                         \
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][ks], fa[i][ks], acc[(mh) * 4 + i][(nh) * 2 + j], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        __builtin_amdgcn_s_barrier();                                                                                         \
    }
// the four phases of k tile kt_ held in buffers (A0_, B0_, B1_, A1_); the half-tiles of k tile kt_ + 1 go into (nA0, nB0, nB1, nA1)
#define GTOS_QTILE(A0_, B0_, B1_, A1_, nA0, nB0, nB1, nA1, kt_)                                                               \
    GTOS_QPHASE({ GTOS_QREAD_B(B0_); __builtin_amdgcn_sched_barrier(0); GTOS_QREAD_A(A0_); }, GTOS_QISSUE(nA0, true, 0, (kt_) + 1), 0, 0); \
    GTOS_QPHASE({ GTOS_QREAD_B(B1_); }, GTOS_QISSUE(nB0, false, 0, (kt_) + 1), 0, 1);                                         \
    GTOS_QPHASE({ GTOS_QREAD_A(A1_); }, GTOS_QISSUE(nB1, false, 1, (kt_) + 1), 1, 1);                                         \
    GTOS_QPHASE({ GTOS_QREAD_B(B0_); }, GTOS_QISSUE(nA1, true, 1, (kt_) + 1), 1, 0);

    GTOS_QISSUE(a00, true, 0, 0);
    GTOS_QISSUE(b00, false, 0, 0);
    GTOS_QISSUE(b01, false, 1, 0);
    GTOS_QISSUE(a01, true, 1, 0);
    GTOS_VMCNT(4);                                         // own pieces of A0, B0 of k tile 0
    __builtin_amdgcn_s_barrier();                          // everybody's
    if (wave >= 4) __builtin_amdgcn_s_barrier();           // the second wave of every SIMD runs one barrier behind
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
        GTOS_QTILE(a00, b00, b01, a01, a10, b10, b11, a11, kt);
        GTOS_QTILE(a10, b10, b11, a11, a00, b00, b01, a01, kt + 1);
    }
    if (kt < nk) { GTOS_QTILE(a00, b00, b01, a01, a10, b10, b11, a11, kt); }
    if (wave < 4) __builtin_amdgcn_s_barrier();            // same number of barriers for both halves
    GTOS_VMCNT(0);                                         // the dummy prefetches of the last phases
    __syncthreads();                                       // every wave is done with the buffers: a00 / b00 become the output staging
#undef GTOS_QTILE
#undef GTOS_QPHASE
#undef GTOS_QREAD_A
#undef GTOS_QREAD_B
#undef GTOS_QISSUE
#undef GTOS_QDMA

    // ---- epilogue (bf16 out), as in gemm256_nt_kernel: 32 rows x 64 columns of the wave tile at a time through LDS
    bf16_t* C = static_cast<bf16_t*>(a.C);
    const float keep_scale = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
    constexpr int CP = 64 * 2 + 16;
    char* cs = (wave < 4 ? a00 : b00) + (wave & 3) * 32 * CP;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int mt = q4 * 2 + mh;
                const int m = m0 + wm + mt * 16 + fr, n = n0 + wn + nt * 16 + fq * 4;
                float v[4] = {acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (a.bias && n + i < a.N) v[i] += a.bias[n + i];
                    if (a.relu) v[i] = fmaxf(v[i], 0.f);
                    if (a.p_drop > 0.f)
                        v[i] = drop_keep(a.seed, (uint64_t)m * (uint64_t)a.N + (uint64_t)(n + i), a.p_drop) ? v[i] * keep_scale : 0.f;
                }
                *reinterpret_cast<uint2*>(cs + (mh * 16 + fr) * CP + (nt * 16 + fq * 4) * 2) = make_uint2(pack_bf(v[0], v[1]), pack_bf(v[2], v[3]));
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 8 + (lane >> 3), col = (lane & 7) * 8;
            const int m = m0 + wm + q4 * 32 + row, n = n0 + wn + col;
            if (m >= a.M || n >= a.N) continue;
            uint4 val = *reinterpret_cast<const uint4*>(cs + row * CP + col * 2);
            bf16_t* cp = C + (int64_t)m * a.ldc + n;
            if (n + 8 <= a.N) {
                if (a.accumulate) {
                    const uint4 old = *reinterpret_cast<const uint4*>(cp);
                    val = make_uint4(pack_bf(lo_bf(val.x) + lo_bf(old.x), hi_bf(val.x) + hi_bf(old.x)),
                                     pack_bf(lo_bf(val.y) + lo_bf(old.y), hi_bf(val.y) + hi_bf(old.y)),
                                     pack_bf(lo_bf(val.z) + lo_bf(old.z), hi_bf(val.z) + hi_bf(old.z)),
                                     pack_bf(lo_bf(val.w) + lo_bf(old.w), hi_bf(val.w) + hi_bf(old.w)));
                }
                *reinterpret_cast<uint4*>(cp) = val;
            } else {
                const bf16_t* e = reinterpret_cast<const bf16_t*>(cs + row * CP + col * 2);
                for (int i = 0; i < 8 && n + i < a.N; ++i) {
                    float o = bf2f(e[i]);
                    if (a.accumulate) o += bf2f(cp[i]);
                    cp[i] = f2bf(o);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

int launch256p(const GemmArgs& a, hipStream_t s) {
    const long long nMt = (a.M + BM2 - 1) / BM2, nNt = (a.N + BN2 - 1) / BN2;
    const long long nblk = ((nMt + 7) / 8) * 8 * nNt;
    if (nblk > 0x7fffffffLL) return -6;
    hipLaunchKernelGGL(gemm256q_nt_kernel, dim3((unsigned)nblk), dim3(512), 0, s, a);
    GTOS_CHECK_LAUNCH();
    return 0;
}

// ---- gemm256p_tn_kernel: the weight-gradient shape dW = A^T B (A [K, M], B [K, N] row-major, K in the hundreds of
//      thousands, split-K into the fp32 partial-tile workspace) on the same 256x256 tile, four 32-k stages and ping-pong
//      schedule as round 2's NT kernel (see the family comment above).  A 256x256 tile re-reads each operand row from L2 half as often as the 128x128
//      tile (these products stream both operands from HBM once and are bound by the L2 -> LDS amplification).
//      A stage holds [32 k][256 m] + [32 k][256 n] (512-byte k-rows); MFMA fragments (8 consecutive k of one column) come
//      from ds_read_b64_tr_b16 as in gemm_kernel's transpose path: 32-byte granules XOR-swizzled with (k & 3), applied
//      on the DMA's source address.  Rows k >= kend read a block of zeros; columns past M / N re-read the last 8 valid
//      ones (never stored); stages past the end of the split re-read its last stage (never multiplied).
//      GROUPED (round 5): the tiles of ONE launch come from a table over TWO B operands -- the GRU weight gradients of a layer and
//      direction, d4^T [x | h_prev] with d4 = [d r | d z | d n_x | d n_h]: dW_ih = d4[:, 0:3hs]^T x and dW_hh = d4[:, {0:2hs, 3hs:4hs}]^T
//      h_prev used to be three products (three passes over d4, two over h_prev); here every (A column block, B column block) pair that
//      holds a needed element is one tile, all tiles of a K split run on one XCD, and d4, x and h_prev are each fetched from HBM once.
//      Partial tiles go to ws[split][tile][256][256]; gru_dw_reduce_kernel adds the needed elements into the two gradients.
struct TnGroup {
    const bf16_t* B1; int64_t ldb1; int N1;        // second B operand [K, N1]; GemmArgs.B / ldb / N describe the first
    int tiles;
    int tmn[64];                                   // tile t: A column block (bits 0-7), B operand (8-15), its column block (16-23), x 256
};

//      BATCH (round 6, MODE 2): the tiles of one launch come from a table of INDEPENDENT products -- the weight gradients of the model's small linear
//      layers (dY^T X over a few thousand rows: 63 such launches of ~22 us each per C2 step, each followed by a split-K reduction, a bias column
//      sum and often a fill: 3.6 ms of latency-bound launches on the main stream).  A workgroup takes one 256x256 tile of one job over the job's
//      WHOLE K (no split: nobody else writes the tile) and adds it to the job's fp32 target in place.
struct TnJob { const bf16_t* A; const bf16_t* B; float* C; int lda, ldb, ldc, M, N, K, tile_end, nN; };     // tile_end: exclusive prefix sum of the jobs' tile counts
constexpr int TN_MAX_JOBS = 48;
struct TnBatch { int njobs, ntiles; TnJob job[TN_MAX_JOBS]; };

template <int MODE, typename G>
__global__ __launch_bounds__(512) void gemm256p_tn_kernel(GemmArgs a, G g) {
    constexpr bool GROUPED = MODE == 1;
    __shared__ __attribute__((aligned(16))) char st0[ST3];
    __shared__ __attribute__((aligned(16))) char st1[ST3];
    __shared__ __attribute__((aligned(16))) char st2[ST3];
    __shared__ __attribute__((aligned(16))) char st3[ST3];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
    const int nN = (a.N + BN2 - 1) / BN2, nM = (a.M + BM2 - 1) / BM2;
    int tiles = nM * nN;
    if constexpr (GROUPED) tiles = g.tiles;
    if constexpr (MODE == 2) tiles = 1;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    int bz = (sq / tiles) * 8 + xcd, t_ = sq % tiles;              // all tiles of one K split on one XCD
    if constexpr (MODE == 2) { bz = 0; t_ = blockIdx.x; if (t_ >= g.ntiles) return; }
    if (bz >= a.splitk) return;
    int m0, n0, Nv;
    const bf16_t* Bp;
    int64_t ldb, wld;
    float* wst;                                                    // this split's partial tile: element (m, n) at wst[(m - m0) * wld + n - n0]
    if constexpr (MODE == 2) {
        int j = 0;
        while (j + 1 < g.njobs && t_ >= g.job[j].tile_end) ++j;
        const TnJob& jb = g.job[j];
        const int tl = t_ - (j ? g.job[j - 1].tile_end : 0);
        m0 = (tl / jb.nN) * BM2; n0 = (tl % jb.nN) * BN2;
        a.A = jb.A; a.lda = jb.lda; a.M = jb.M; a.K = jb.K; a.splitk = 1;
        Bp = jb.B; ldb = jb.ldb; Nv = jb.N;
        wst = jb.C + (int64_t)m0 * jb.ldc + n0; wld = jb.ldc;
    } else if constexpr (GROUPED) {
        const int e = g.tmn[t_], src = (e >> 8) & 255;
        m0 = (e & 255) * BM2; n0 = ((e >> 16) & 255) * BN2;
        Bp = src ? g.B1 : static_cast<const bf16_t*>(a.B); ldb = src ? g.ldb1 : a.ldb; Nv = src ? g.N1 : a.N;
        wst = a.ws + ((int64_t)bz * tiles + t_) * (BM2 * BN2); wld = BN2;
    } else {
        m0 = (t_ / nN) * BM2; n0 = (t_ % nN) * BN2;
        Bp = static_cast<const bf16_t*>(a.B); ldb = a.ldb; Nv = a.N;
        wst = a.ws + ((int64_t)bz * a.M + m0) * a.N + n0; wld = a.N;
    }
    const int ktiles = (a.K + 63) / 64, tps = (ktiles + a.splitk - 1) / a.splitk;   // the split boundaries of gemm_kernel
    const int kbeg = bz * tps * 64, kend = min(a.K, kbeg + tps * 64);
    if (kbeg >= kend) return;
    const int nk = (kend - kbeg + 31) / 32;
    const char* Z = static_cast<const char*>(a.zeros);
    // DMA: a wave instruction fills 1 KB = 2 k-rows x 512 B; lane l -> k-row l >> 5, physical 16-byte piece l & 31.
    // Wave w owns blocks w and w + 8 of each operand: k-rows 2w, 2w+1 and 2w+16, 2w+17 (k & 3 the same for both).
    const int kr = lane >> 5, pp = lane & 31;
    const int kl0 = 2 * wave + kr, kq_d = kl0 & 3;
    const int lp = ((((pp >> 1) ^ kq_d)) << 1) | (pp & 1);        // logical piece (8 columns) stored at this physical slot
    const int acol = min(m0 + lp * 8, a.M - 8), bcol = min(n0 + lp * 8, Nv - 8);
    // running per-lane source pointers of the NEXT stage to fetch (k-row kcur and kcur + 16 of each operand); rows at or
    // past kend -- also every row of the dummy stages after the split's last one -- read the block of zeros instead
    const char* pa = reinterpret_cast<const char*>(static_cast<const bf16_t*>(a.A) + (int64_t)(kbeg + kl0) * a.lda + acol);
    const char* pb = reinterpret_cast<const char*>(Bp + (int64_t)(kbeg + kl0) * ldb + bcol);
    const int64_t a16 = 32 * a.lda, b16 = 32 * ldb;        // bytes: 16 k-rows
    int kcur = kbeg + kl0;
    // fragment reads (tr_fragment of gemm_kernel with 512-byte k-rows): lane addresses k-row fq*8 + (fr >> 2) (+4), the
    // 32-byte granule (column / 16) ^ (fr >> 2), piece fr & 3; column offsets are multiples of 64 elements + t*16
    const int kq = fr >> 2;
    const int fbase = (fq * 8 + kq) * 512 + (fr & 3) * 8;

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t fa[8], fb[4];

#define GTOS_DMA1(src, dst)                                                                                                   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                                    \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
#define GTOS_DMA4(stage)                                                                                                      \
    {                                                                                                                         \
        const bool ok0 = kcur < kend, ok1 = kcur + 16 < kend;                                                                 \
        GTOS_DMA1(ok0 ? pa : Z, (stage) + wave * 1024);                                                                       \
        GTOS_DMA1(ok1 ? pa + a16 : Z, (stage) + (8 + wave) * 1024);                                                           \
        GTOS_DMA1(ok0 ? pb : Z, (stage) + A3 + wave * 1024);                                                                  \
        GTOS_DMA1(ok1 ? pb + b16 : Z, (stage) + A3 + (8 + wave) * 1024);                                                      \
        kcur += 32; pa += 2 * a16; pb += 2 * b16;                                                                             \
    }
#define GTOS_TRF(tile, c0)                                                                                                    \
    ([&]() -> bf16x8_t {                                                                                                      \
        const char* p0 = (tile) + fbase + ((((c0) >> 4) ^ kq) << 5);                                                          \
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0));         \
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 4 * 512)); \
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                                           \
        return __builtin_bit_cast(bf16x8_t, v);                                                                               \
    }())
#define GTOS_STEP(slot_s, slot_d, s_)                                                                                         \
    {                                                                                                                         \
        GTOS_DMA4(slot_d);                                                                                                  \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) fb[t] = GTOS_TRF((slot_s) + A3, wn + t * 16);                           \
        _Pragma("unroll") for (int t = 0; t < 8; ++t) fa[t] = GTOS_TRF((slot_s), wm + t * 16);                                \
        __builtin_amdgcn_s_waitcnt(0x0078);                /* vmcnt(8) lgkmcnt(0): fragments here, own pieces of s_+1 landed */ \
        __builtin_amdgcn_s_barrier();                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        _Pragma("unroll") for (int mt = 0; mt < 8; ++mt)                                                                      \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                                  \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        __builtin_amdgcn_s_barrier();                                                                                         \
    }

    GTOS_DMA4(st0);
    GTOS_DMA4(st1);
    GTOS_DMA4(st2);
    GTOS_VMCNT(8);
    __builtin_amdgcn_s_barrier();
    if (wave >= 4) __builtin_amdgcn_s_barrier();           // the second wave of every SIMD starts one segment late
    int s = 0;
    for (; s + 4 <= nk; s += 4) {
        GTOS_STEP(st0, st3, s);
        GTOS_STEP(st1, st0, s + 1);
        GTOS_STEP(st2, st1, s + 2);
        GTOS_STEP(st3, st2, s + 3);
    }
    if (s < nk) {
        GTOS_STEP(st0, st3, s);
        if (s + 1 < nk) {
            GTOS_STEP(st1, st0, s + 1);
            if (s + 2 < nk) GTOS_STEP(st2, st1, s + 2);
        }
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();            // same number of barriers for both halves
#undef GTOS_STEP
#undef GTOS_TRF
#undef GTOS_DMA4
#undef GTOS_DMA1

    // ---- this split's partial tile -> workspace (N % 4 == 0: plain 16-byte stores), reduced by splitk_reduce_kernel / gru_dw_reduce_kernel
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int m = m0 + wm + mt * 16 + fr;
        if (m >= a.M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn + nt * 16 + fq * 4;
            if (n >= Nv) continue;
            float4* dst = reinterpret_cast<float4*>(wst + (int64_t)(m - m0) * wld + (n - n0));
            if constexpr (MODE == 2) {                     // the job's gradient itself: this workgroup is the tile's only writer
                const float4 o = *dst;
                *dst = make_float4(o.x + acc[mt][nt][0], o.y + acc[mt][nt][1], o.z + acc[mt][nt][2], o.w + acc[mt][nt][3]);
            } else {
                *dst = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
            }
        }
    }
}

// column sums of the A operands of a batch (the layers' bias gradients): block (x, y) of job j sums rows y*rpb .. of columns x*512 .. x*512+511
struct ColJob { const bf16_t* A; float* out; int lda, M, K, blk_end, gx; };
struct ColBatch { int njobs, nblocks, rpb; ColJob job[TN_MAX_JOBS]; };
__global__ __launch_bounds__(256) void colsum_batch_kernel(ColBatch g) {
    __shared__ float red[4][64][8];
    int t_ = blockIdx.x, j = 0;
    if (t_ >= g.nblocks) return;
    while (j + 1 < g.njobs && t_ >= g.job[j].blk_end) ++j;
    const ColJob& jb = g.job[j];
    const int tl = t_ - (j ? g.job[j - 1].blk_end : 0);
    const int bx = tl % jb.gx, by = tl / jb.gx;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = (bx * 64 + lane) * 8;
    const int r0 = by * g.rpb, r1 = min(jb.K, r0 + g.rpb);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < jb.M) {                                        // M % 8 == 0: whole vectors
        for (int r = r0 + w; r < r1; r += 4) {
            float v[8];
            Vec8<bf16_t>::load(jb.A + (int64_t)r * jb.lda + c, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[w][lane][e] = acc[e];
    __syncthreads();
    if (w == 0 && c < jb.M) {
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(jb.out + c + e, red[0][lane][e] + red[1][lane][e] + red[2][lane][e] + red[3][lane][e]);
    }
}

// dW_ih[(g, c), j] += sum over splits of the partial element (A column i = g * hs + c, x column j) for g in {r, z, n_x};
// dW_hh[(g', c), j] += ... (A column i, h_prev column j) for g in {r, z, n_h} (n_h = gate block 3 of d4 -> row block 2 of W_hh)
struct GruDwOut { float* dwih; int64_t ld_ih; float* dwhh; int64_t ld_hh; int hs, in_dim; };
__global__ void gru_dw_reduce_kernel(const float* __restrict__ ws, int splits, TnGroup g, GruDwOut o) {
    const int t_ = blockIdx.x >> 6, q = (blockIdx.x & 63) * 256 + threadIdx.x;        // 64 blocks x 256 threads x float4 = one 256 x 256 tile
    const int e = g.tmn[t_], src = (e >> 8) & 255;
    const int r = q >> 6, c = (q & 63) * 4;
    const int i = (e & 255) * BM2 + r, j = ((e >> 16) & 255) * BN2 + c;
    if (i >= 4 * o.hs) return;
    const int gate = i / o.hs, ch = i - gate * o.hs;
    float* dst;
    if (src == 0) {
        if (gate == 3 || j >= o.in_dim) return;
        dst = o.dwih + (int64_t)(gate * o.hs + ch) * o.ld_ih + j;
    } else {
        if (gate == 2 || j >= o.hs) return;
        dst = o.dwhh + (int64_t)((gate == 3 ? 2 : gate) * o.hs + ch) * o.ld_hh + j;
    }
    const float* p = ws + (int64_t)t_ * (BM2 * BN2) + r * BN2 + c;
    const int64_t plane = (int64_t)g.tiles * (BM2 * BN2);
    float4 acc = *reinterpret_cast<const float4*>(p);
    for (int s = 1; s < splits; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(p + s * plane);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    dst[0] += acc.x; dst[1] += acc.y; dst[2] += acc.z; dst[3] += acc.w;
}

int launch256p_tn(const GemmArgs& a, hipStream_t s) {
    const long long tiles = ((long long)(a.M + BM2 - 1) / BM2) * ((a.N + BN2 - 1) / BN2);
    const long long nblk = tiles * 8 * ((a.splitk + 7) / 8);
    if (nblk > 0x7fffffffLL) return -6;
    hipLaunchKernelGGL((gemm256p_tn_kernel<0, TnGroup>), dim3((unsigned)nblk), dim3(512), 0, s, a, TnGroup{});
    GTOS_CHECK_LAUNCH();
    return 0;
}

int launch256(const GemmArgs& a, hipStream_t s) {
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_nt_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * STAGE2) != hipSuccess) return -7;
        configured = true;
    }
    const long long nMt = (a.M + BM2 - 1) / BM2, nNt = (a.N + BN2 - 1) / BN2;
    const long long nblk = ((nMt + 7) / 8) * 8 * nNt;
    if (nblk > 0x7fffffffLL) return -6;
    hipLaunchKernelGGL(gemm256_nt_kernel, dim3((unsigned)nblk), dim3(512), 2 * STAGE2, s, a);
    GTOS_CHECK_LAUNCH();
    return 0;
}

template <typename T, typename TO>
int launch(const GemmArgs& a, int transA, int transB, hipStream_t s) {
    const long long nMt = (a.M + BM - 1) / BM, nNt = (a.N + BN - 1) / BN;
    const long long nblk = a.splitk > 1 ? nMt * nNt * 8 * ((a.splitk + 7) / 8) : ((nMt + 7) / 8) * 8 * nNt;
    if (nblk > 0x7fffffffLL) return -6;
    dim3 grid((unsigned)nblk);
    const bool fast = a.vecA && a.vecB;          // both operands 16-byte aligned with whole vectors in range
    constexpr bool BF = sizeof(T) == 2;          // bf16: every layout is all-DMA (transpose reads) -> single-stage kernel
    if (!transA && transB) {
        if (fast) hipLaunchKernelGGL((gemm_kernel<T, TO, false, true, true, 256, 1>), grid, dim3(256), 0, s, a);
        else      hipLaunchKernelGGL((gemm_kernel<T, TO, false, true, false, 512, 2>), grid, dim3(512), 0, s, a);
    } else if (!transA && !transB) {
        if (fast && BF) hipLaunchKernelGGL((gemm_kernel<T, TO, false, false, true, 256, BF ? 1 : 2>), grid, dim3(256), 0, s, a);
        else if (fast)  hipLaunchKernelGGL((gemm_kernel<T, TO, false, false, true, 512, 2>), grid, dim3(512), 0, s, a);
        else            hipLaunchKernelGGL((gemm_kernel<T, TO, false, false, false, 512, 2>), grid, dim3(512), 0, s, a);
    } else if (transA && !transB) {
        if (fast && BF) hipLaunchKernelGGL((gemm_kernel<T, TO, true, false, true, 256, BF ? 1 : 2>), grid, dim3(256), 0, s, a);
        else if (fast)  hipLaunchKernelGGL((gemm_kernel<T, TO, true, false, true, 512, 2>), grid, dim3(512), 0, s, a);
        else            hipLaunchKernelGGL((gemm_kernel<T, TO, true, false, false, 512, 2>), grid, dim3(512), 0, s, a);
    } else return -2;
    GTOS_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int gtos_gemm(int in_dtype, int out_dtype, int transA, int transB, int M, int N, int K,
                         const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                         const float* bias, int relu, float p_drop, uint64_t seed, int accumulate,
                         int splitk, void* workspace, int64_t workspace_bytes, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0) return -3;
    const int es = in_dtype == GTOS_BF16 ? 2 : 4, vec = 16 / es;
    const int eo = out_dtype == GTOS_BF16 ? 2 : 4;
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    // K-contiguous operands need K % vec == 0, M/N-contiguous ones need M (resp. N) % vec == 0 (whole vectors in range)
    a.vecA = ((uintptr_t)A % 16 == 0) && (lda % vec == 0) && ((transA ? M : K) % vec == 0);
    a.vecB = ((uintptr_t)B % 16 == 0) && (ldb % vec == 0) && ((transB ? K : N) % vec == 0);
    a.vecC = ((uintptr_t)C % 16 == 0) && (ldc % (16 / eo) == 0);
    a.relu = relu; a.accumulate = accumulate; a.p_drop = p_drop; a.seed = seed;
    a.zeros = zero_block();
    if (!a.zeros) return -5;
    if (splitk < 1) splitk = 1;
    const int BKc = in_dtype == GTOS_BF16 ? GemmCfg<bf16_t>::BK : GemmCfg<float>::BK;
    const int ktiles = (K + BKc - 1) / BKc;
    if (splitk > ktiles) splitk = ktiles;
    if (splitk > 1) { const int tps = (ktiles + splitk - 1) / splitk; splitk = (ktiles + tps - 1) / tps; }   // no empty split
    a.splitk = splitk;
    if (splitk > 1 && (out_dtype != GTOS_F32 || bias || relu || p_drop > 0.f || !accumulate)) return -4;
    hipStream_t s = static_cast<hipStream_t>(stream);
    a.ws = nullptr;
    if (splitk > 1 && workspace && N % 4 == 0 && (uintptr_t)workspace % 16 == 0 && (uintptr_t)C % 4 == 0 &&
        (int64_t)splitk * M * N * 4 <= workspace_bytes) {
        a.ws = static_cast<float*>(workspace);
        // long-K weight gradients with at least one full 256x256 tile: the four-stage ping-pong kernel
        const int kps = ((ktiles + splitk - 1) / splitk) * BKc;            // k per split
        const bool tn256 = g_use_pipe_tn && in_dtype == GTOS_BF16 && transA && !transB && a.vecA && a.vecB && M >= 256 && N >= 256 &&
                           M % 8 == 0 && N % 8 == 0 && kps >= 256;
        int rc = tn256 ? launch256p_tn(a, s)
                       : (in_dtype == GTOS_BF16 ? launch<bf16_t, float>(a, transA, transB, s) : launch<float, float>(a, transA, transB, s));
        if (rc) return rc;
        const int64_t total = (int64_t)M * (N / 4);
        const int nb = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb), dim3(256), 0, s, a.ws, splitk, M, N, static_cast<float*>(C), ldc);
        GTOS_CHECK_LAUNCH();
        return 0;
    }
    if (in_dtype == GTOS_BF16 && out_dtype == GTOS_BF16) {
        // big forward-shaped products go to the 256x256 macro tile (enough tiles to give every CU several)
        const long long t256 = ((long long)(M + BM2 - 1) / BM2) * ((N + BN2 - 1) / BN2);
        // K % 32 == 0 and enough macro tiles: the software-pipelined 256x256 kernel
        if (g_use_pipe && !transA && transB && a.vecA && a.vecB && a.vecC && splitk == 1 && N >= 256 && K % 32 == 0 &&
            lda < (1 << 22) && ldb < (1 << 22) &&
            (t256 >= 512 && K >= g_min_kpipe))
            return launch256p(a, s);
        if (g_use256 && !transA && transB && a.vecA && a.vecB && a.vecC && splitk == 1 && t256 >= 1024 && N >= 256 && K >= g_min_k256)
            return launch256(a, s);
        return launch<bf16_t, bf16_t>(a, transA, transB, s);
    }
    if (in_dtype == GTOS_BF16 && out_dtype == GTOS_F32)  return launch<bf16_t, float>(a, transA, transB, s);
    if (in_dtype == GTOS_F32 && out_dtype == GTOS_F32)   return launch<float, float>(a, transA, transB, s);
    return -1;
}


extern "C" int gtos_gru_weight_grads(int rows, int hs, int in_dim, int in_valid, const void* d4, const void* x, int64_t ldx, const void* hprev, int64_t ldh,
                                     float* dwih, int64_t ld_dwih, float* dwhh, int64_t ld_dwhh, void* workspace, int64_t workspace_bytes,
                                     void* stream) {
    if (rows <= 0) return 0;
    if (hs <= 0 || hs % 64 || in_dim < 8 || in_dim % 8 || ldx < in_dim || ldx % 8 || ldh < hs || ldh % 8 || in_valid < 1 || in_valid > in_dim ||
        in_valid % 4) return -22;
    if (!d4 || !x || !hprev || !dwih || !dwhh || !workspace) return -23;
    if ((uintptr_t)d4 % 16 || (uintptr_t)x % 16 || (uintptr_t)hprev % 16 || (uintptr_t)dwih % 4 || (uintptr_t)dwhh % 4 || (uintptr_t)workspace % 16 ||
        ld_dwih < in_valid || ld_dwhh < hs) return -25;
    GemmArgs a{};
    a.A = d4; a.lda = 4 * (int64_t)hs; a.M = 4 * hs; a.B = x; a.ldb = ldx; a.N = in_dim; a.K = rows; a.vecA = a.vecB = 1;
    a.accumulate = 1; a.zeros = zero_block();
    if (!a.zeros) return -5;
    TnGroup g{};
    g.B1 = static_cast<const bf16_t*>(hprev); g.ldb1 = ldh; g.N1 = hs;
    int nt = 0;
    for (int mb = 0; mb * BM2 < 4 * hs; ++mb) {
        const int lo = mb * BM2, hi = (lo + BM2 < 4 * hs ? lo + BM2 : 4 * hs) - 1;
        const int g_lo = lo / hs, g_hi = hi / hs;
        const bool needs_x = g_lo <= 2, needs_h = !(g_lo == 2 && g_hi == 2);
        for (int src = 0; src < 2; ++src) {
            if (!(src ? needs_h : needs_x)) continue;
            const int ncols = src ? hs : in_dim;
            for (int nb = 0; nb * BN2 < ncols; ++nb) {
                if (nt >= 64) return -28;
                g.tmn[nt++] = mb | (src << 8) | (nb << 16);
            }
        }
    }
    g.tiles = nt;
    // whole splits per XCD (its 32 CUs hold one 128 KB-LDS workgroup each), at least four 64-row k tiles per split
    const int ktiles = (rows + 63) / 64;
    const long long tile_bytes = (long long)BM2 * BN2 * 4;
    long long sk = 8LL * (32 / nt > 0 ? 32 / nt : 1);
    // (a cap on this product's workgroups -- 96 .. 192 instead of its 216-240 -- was measured in round 6: no gain, 77.8-79.1 vs 77.6-77.9 ms per
    //  step; beside the step kernels of the other stream the two are work-conserving.  profiles/r6_ab_switches.txt)
    if (sk > ktiles / 4) sk = ktiles / 4;
    if (sk * nt * tile_bytes > workspace_bytes) sk = workspace_bytes / (nt * tile_bytes);
    if (sk < 1) { if (nt * tile_bytes > workspace_bytes) return -4; sk = 1; }
    const int tps = (int)((ktiles + sk - 1) / sk);
    a.splitk = (ktiles + tps - 1) / tps;                                   // no empty split
    a.ws = static_cast<float*>(workspace);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long nblk = (long long)nt * 8 * ((a.splitk + 7) / 8);
    hipLaunchKernelGGL((gemm256p_tn_kernel<1, TnGroup>), dim3((unsigned)nblk), dim3(512), 0, s, a, g);
    GTOS_CHECK_LAUNCH();
    GruDwOut o{dwih, ld_dwih, dwhh, ld_dwhh, hs, in_valid};            // columns in_valid .. in_dim of x are the zero padding of its rows
    hipLaunchKernelGGL(gru_dw_reduce_kernel, dim3((unsigned)(nt * 64)), dim3(256), 0, s, a.ws, a.splitk, g, o);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_gemm_tn_batch(int n, const void* const* A, const int64_t* lda, const int* M, const void* const* B, const int64_t* ldb,
                                  const int* N, const int* K, float* const* C, const int64_t* ldc, float* const* bias, void* stream) {
    if (n <= 0) return 0;
    if (!A || !lda || !M || !B || !ldb || !N || !K || !C || !ldc) return -23;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const void* zeros = zero_block();
    if (!zeros) return -5;
    for (int j = 0; j < n; ++j) {
        if (M[j] < 8 || M[j] % 8 || N[j] < 8 || N[j] % 8 || K[j] < 1 || lda[j] < M[j] || lda[j] % 8 || ldb[j] < N[j] || ldb[j] % 8 || ldc[j] < N[j] ||
            ldc[j] % 4 || lda[j] >= (1LL << 31) || ldb[j] >= (1LL << 31) || ldc[j] >= (1LL << 31)) return -22;
        if (!A[j] || !B[j] || !C[j] || (uintptr_t)A[j] % 16 || (uintptr_t)B[j] % 16 || (uintptr_t)C[j] % 16) return -25;
        if (bias && bias[j] && (uintptr_t)bias[j] % 4) return -25;
    }
    for (int j0 = 0; j0 < n; j0 += TN_MAX_JOBS) {
        const int nj = n - j0 < TN_MAX_JOBS ? n - j0 : TN_MAX_JOBS;
        TnBatch tb{};
        ColBatch cb{};
        int tiles = 0, blocks = 0, ncol = 0;
        cb.rpb = 512;
        for (int j = 0; j < nj; ++j) {
            const int q = j0 + j;
            TnJob& t = tb.job[j];
            t.A = static_cast<const bf16_t*>(A[q]); t.B = static_cast<const bf16_t*>(B[q]); t.C = C[q];
            t.lda = (int)lda[q]; t.ldb = (int)ldb[q]; t.ldc = (int)ldc[q]; t.M = M[q]; t.N = N[q]; t.K = K[q];
            t.nN = (N[q] + BN2 - 1) / BN2;
            tiles += ((M[q] + BM2 - 1) / BM2) * t.nN;
            t.tile_end = tiles;
            if (bias && bias[q]) {
                ColJob& c = cb.job[ncol++];
                c.A = t.A; c.out = bias[q]; c.lda = t.lda; c.M = t.M; c.K = t.K;
                c.gx = (t.M + 511) / 512;
                blocks += c.gx * ((t.K + cb.rpb - 1) / cb.rpb);
                c.blk_end = blocks;
            }
        }
        tb.njobs = nj; tb.ntiles = tiles;
        GemmArgs a{};
        a.splitk = 1; a.zeros = zeros; a.vecA = a.vecB = 1; a.accumulate = 1;
        hipLaunchKernelGGL((gemm256p_tn_kernel<2, TnBatch>), dim3((unsigned)tiles), dim3(512), 0, s, a, tb);
        GTOS_CHECK_LAUNCH();
        if (ncol) {
            cb.njobs = ncol; cb.nblocks = blocks;
            hipLaunchKernelGGL(colsum_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, s, cb);
            GTOS_CHECK_LAUNCH();
        }
    }
    return 0;
}

GTOS_SEED_EPOCH_SETTER(gemm)
