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
#include "gemm_w4_gen.h"
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

// ---- gemm256p_nt_kernel: the same 256x256 tile / 8 waves with FOUR 32-k LDS stages and a ping-pong schedule, for
//      K % 32 == 0, K >= 1024, N <= 2048 (measured: 1.18 vs 1.10 PF/s at [434624,4096]x[1024,4096]^T, +9..11 % over the
//      128x128 kernel at K = 1024..2016; below K ~ 700 the 128x128 kernel's four co-resident workgroups win, and with a
//      wide N (8192^3) the 64-byte rows cost L2 efficiency: those shapes stay on the kernels above).
//      * The stages are SEPARATE static LDS arrays: hipcc's wait-count pass tracks LDS-DMA per underlying LDS object, so
//        a ds_read of one stage does not wait for the DMA in flight into the others (with ONE LDS array it drains
//        vmcnt(0) in front of every ds_read -- what limits gemm256_nt_kernel to one tile of prefetch, issued after the
//        fragment reads).  The waits are explicit: s_waitcnt vmcnt(8) leaves a wave's 8 newest DMA instructions (its
//        pieces of two stages) in flight; hipcc accepts it and adds none of its own (checked in the ISA).
//      * A step of a wave = a LOAD segment (its 4 DMA pieces of stage s+3 into the slot of stage s-1, the 12 fragment
//        reads of stage s, the wait that retires its pieces of stage s+1) and a COMPUTE segment (32 MFMAs), each closed
//        by s_barrier.  Waves 4-7 run one segment behind waves 0-3: on every SIMD one wave multiplies while its partner
//        issues DMA / LDS reads (a DMA piece costs its wave 60-180 issue cycles).  Three stages (96 KB) in flight per CU.
//      * LDS rows are 64 bytes (32 k); 16-byte chunk c of row r sits at chunk c ^ ((-(r >> 2)) & 3): the 16 accesses the
//        LDS serves together (rows {0-3,12-15} of one chunk, rows {4-11} of the next) hit 16 different 16-byte slots.
//        The swizzle is a per-lane constant for both the DMA source address and the fragment reads (row offsets are
//        multiples of 16): fragments are one address register + immediate offsets, DMA sources uniform base + 32-bit
//        lane offset.
//      * Rows past the end re-read the last valid row (their products are never stored); stages past the end of K
//        re-read the last stage (never multiplied), so the number of DMAs in flight is the same in every step.
constexpr int ROW3 = 64, A3 = BM2 * ROW3, ST3 = (BM2 + BN2) * ROW3;
const int g_min_k256 = getenv("GTOS_GEMM256_MINK") ? atoi(getenv("GTOS_GEMM256_MINK")) : 2048;   // A/B switch (tools/bench_gemm.py)
const bool g_use_pipe = !(getenv("GTOS_GEMM_PIPE") && getenv("GTOS_GEMM_PIPE")[0] == '0');
const bool g_use_pipe_tn = !(getenv("GTOS_GEMM_PIPE_TN") && getenv("GTOS_GEMM_PIPE_TN")[0] == '0');
const int g_max_npipe = getenv("GTOS_GEMM_PIPE_MAXN") ? atoi(getenv("GTOS_GEMM_PIPE_MAXN")) : 2048;
const int g_min_kpipe = getenv("GTOS_GEMM_PIPE_MINK") ? atoi(getenv("GTOS_GEMM_PIPE_MINK")) : 1024;
// Small products (a few thousand rows: the graph layers' and the decoder's projections) leave the single-stage 128x128 kernel with
// less than one workgroup per CU, so nothing covers its load latency: 16 k tiles = 16 exposed round trips, 20-27 us for 3 GFLOP.
// The four-stage kernel keeps its own loads in flight; with <= g_pipe_small_max macro tiles it runs them on a few dozen CUs.
const int g_pipe_small_max = getenv("GTOS_GEMM_PIPE_SMALL") ? atoi(getenv("GTOS_GEMM_PIPE_SMALL")) : 0;

#define GTOS_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))

__global__ __launch_bounds__(512) void gemm256p_nt_kernel(GemmArgs a) {
    if (a.p_drop > 0.f) a.seed = live_seed(a.seed);
    __shared__ __attribute__((aligned(16))) char st0[ST3];
    __shared__ __attribute__((aligned(16))) char st1[ST3];
    __shared__ __attribute__((aligned(16))) char st2[ST3];
    __shared__ __attribute__((aligned(16))) char st3[ST3];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
    const int nN = (a.N + BN2 - 1) / BN2;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    const int m0 = ((sq / nN) * 8 + xcd) * BM2, n0 = (sq % nN) * BN2;
    if (m0 >= a.M) return;
    const char* Ab = reinterpret_cast<const char*>(static_cast<const bf16_t*>(a.A) + (int64_t)m0 * a.lda);
    const char* Bb = reinterpret_cast<const char*>(static_cast<const bf16_t*>(a.B) + (int64_t)n0 * a.ldb);
    const int amax = a.M - 1 - m0, bmax = a.N - 1 - n0;
    const uint32_t lda2 = (uint32_t)a.lda * 2u, ldb2 = (uint32_t)a.ldb * 2u;
    const int nk = a.K / 32;
    // DMA: a wave instruction fills 1 KB = 16 rows x 64 B; lane l -> row l >> 2, physical chunk l & 3
    const int drow = lane >> 2;
    const uint32_t dchunk = (uint32_t)(((lane & 3) ^ ((-(lane >> 4)) & 3)) << 4);
    const uint32_t aoff0 = (uint32_t)min(wave * 16 + drow, amax) * lda2 + dchunk;
    const uint32_t aoff1 = (uint32_t)min((8 + wave) * 16 + drow, amax) * lda2 + dchunk;
    const uint32_t boff0 = (uint32_t)min(wave * 16 + drow, bmax) * ldb2 + dchunk;
    const uint32_t boff1 = (uint32_t)min((8 + wave) * 16 + drow, bmax) * ldb2 + dchunk;
    // fragment reads: row (.. + fr), logical chunk fq
    const int foff = fr * ROW3 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t fa[8], fb[4];

#define GTOS_DMA1(src, dst)                                                                                                   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                                    \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
#define GTOS_DMA4(stage, s_)                                                                                                  \
    {                                                                                                                         \
        const int kb_ = min((s_), nk - 1) * 64;            /* byte offset of the stage's k range; past the end: dummy re-read */ \
        GTOS_DMA1(Ab + kb_ + aoff0, (stage) + wave * 1024);                                                                   \
        GTOS_DMA1(Ab + kb_ + aoff1, (stage) + (8 + wave) * 1024);                                                             \
        GTOS_DMA1(Bb + kb_ + boff0, (stage) + A3 + wave * 1024);                                                              \
        GTOS_DMA1(Bb + kb_ + boff1, (stage) + A3 + (8 + wave) * 1024);                                                        \
    }
// one step of one wave = a LOAD segment (this wave's four DMA pieces of stage s_+3 into the slot of stage s_-1, the 12
// fragment reads of stage s_, then the wait that retires this wave's pieces of stage s_+1) and a COMPUTE segment (32
// MFMAs), each closed by a barrier.  Waves 4-7 run one segment behind waves 0-3, so on every SIMD one wave computes
// while its partner loads.
#define GTOS_STEP(slot_s, slot_d, s_)                                                                                         \
    {                                                                                                                         \
        GTOS_DMA4(slot_d, (s_) + 3);                                                                                          \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                         \
            fb[t] = *reinterpret_cast<const bf16x8_t*>((slot_s) + A3 + (wn + t * 16) * ROW3 + foff);                          \
        _Pragma("unroll") for (int t = 0; t < 8; ++t)                                                                         \
            fa[t] = *reinterpret_cast<const bf16x8_t*>((slot_s) + (wm + t * 16) * ROW3 + foff);                               \
        __builtin_amdgcn_s_waitcnt(0x0078);                /* vmcnt(8) lgkmcnt(0): fragments here, own pieces of s_+1 landed */ \
        __builtin_amdgcn_s_barrier();                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        _Pragma("unroll") for (int mt = 0; mt < 8; ++mt)                                                                      \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                                  \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        __builtin_amdgcn_s_barrier();                                                                                         \
    }

    GTOS_DMA4(st0, 0);
    GTOS_DMA4(st1, 1);
    GTOS_DMA4(st2, 2);
    GTOS_VMCNT(8);                                         // own pieces of stage 0
    __builtin_amdgcn_s_barrier();                          // everybody's
    if (wave >= 4) __builtin_amdgcn_s_barrier();           // the second wave of every SIMD starts one segment late
    int s = 0;
    for (; s + 4 <= nk; s += 4) {                          // whole quads only: one path through the body, so the wait-count
        GTOS_STEP(st0, st3, s);                            // pass sees the same DMA order on the back edge as on entry
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
    GTOS_VMCNT(0);                                         // the dummy prefetches of the last steps
    __syncthreads();                                       // every wave is done with the stages: st0 / st1 become the output staging
#undef GTOS_STEP
#undef GTOS_READ
#undef GTOS_DMA4
#undef GTOS_DMA1

    // ---- epilogue (bf16 out), as in gemm256_nt_kernel: 32 rows x 64 columns of the wave tile at a time through LDS
    bf16_t* C = static_cast<bf16_t*>(a.C);
    const float keep_scale = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
    constexpr int CP = 64 * 2 + 16;
    char* cs = (wave < 4 ? st0 : st1) + (wave & 3) * 32 * CP;
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

// ---- gemm_p2_nt_kernel (round 4): the forward shape with a SHORT reduction (K = 128 .. 1023: relation projection, GRU gate tables,
//      RelationEncoder output projection), where a tile's output write costs as much as its k loop.  Measured picture behind it
//      (tools/probes/membw_probe.hip): the chip streams reads at 6.0-6.3 TB/s but WRITES at 4.4 TB/s, so [R,1024] bf16 = 0.89 GB of
//      output is 0.20 ms of HBM time against 0.19 ms of MFMA time at the peak -- the two must overlap, and inside one workgroup they
//      cannot (the accumulators are the data being stored).  So: TWO independent 4-wave workgroups per CU (one wave of each per SIMD),
//      each on its own 128 x 256 tile; they drift apart by themselves, and while one drains its tile through the store queue the
//      other has the matrix pipes.  First version (three 32-k stages, every wave loading its share of A and B): 0.65 ms at the
//      relation projection, no better than the single-stage kernel -- the A operand streams from HBM (2+ us under load) and two
//      stages of prefetch cover 0.4 us of MFMA time, so every step waited.  A wave's vmcnt retires loads IN ORDER, so one wave
//      cannot hold a deep A prefetch and a shallow B one at once; hence the ROLE SPLIT: waves 0-1 load only A into a SIX-slot ring
//      (five 8 KB stages = 40 KB of HBM reads in flight per workgroup), waves 2-3 load only B (the weight, L2-resident: two 16 KB
//      slots are enough); all four multiply.  80 KB of LDS per workgroup.  Stages are separate static arrays and every wait is
//      explicit, as in gemm256p_nt_kernel (hipcc would otherwise drain all LDS-DMA in front of each ds_read).  Step s: loaders wait
//      for their own pieces of stage s (A: vmcnt(16) = four newer stages stay in flight; B: vmcnt(0)), barrier (stage s complete,
//      every wave past its reads of stage s-1), A(s+5) / B(s+1) issued into the slots of stage s-1, 12 fragment reads, 32 MFMAs.
//      Rows past M / N re-read the last valid row (never stored); stages past K re-read the last stage (never multiplied).
constexpr int P2_BM = 128, P2_BN = 256, P2_AS = P2_BM * ROW3, P2_BS = P2_BN * ROW3;   // 8 KB / 16 KB per stage
// Measured (profiles/r4c_gemm_p2_*.txt, r4d_gemm_p2_*.txt; same box, back to back): relation projection 0.653 ms (three-stage
// version, every wave loading A and B) and 0.706 ms (this role-split version) against 0.637-0.644 ms for the single-stage 128x128
// kernel and 0.65-0.69 ms for torch.matmul; t(K) slope 1.04 us per k against 0.96 (0.36 at the MFMA peak) -- the deep A ring changed
// nothing, so the k loop of all these kernels is not waiting for HBM latency; in the training step: graph encoder forward -0.3..-0.5 ms,
// RelationEncoder +0.2, step unchanged (61.0 / 60.8 vs 60.95 ms).  Kept as an opt-in (GTOS_GEMM_P2=1) with its parity test.
const bool g_use_p2 = getenv("GTOS_GEMM_P2") && getenv("GTOS_GEMM_P2")[0] == '1';
const int g_p2_maxk = getenv("GTOS_GEMM_P2_MAXK") ? atoi(getenv("GTOS_GEMM_P2_MAXK")) : 1023;

__global__ __launch_bounds__(256, 2) void gemm_p2_nt_kernel(GemmArgs a) {
    if (a.p_drop > 0.f) a.seed = live_seed(a.seed);
    __shared__ __attribute__((aligned(16))) char a0[P2_AS];
    __shared__ __attribute__((aligned(16))) char a1[P2_AS];
    __shared__ __attribute__((aligned(16))) char a2[P2_AS];
    __shared__ __attribute__((aligned(16))) char a3[P2_AS];
    __shared__ __attribute__((aligned(16))) char a4[P2_AS];
    __shared__ __attribute__((aligned(16))) char a5[P2_AS];
    __shared__ __attribute__((aligned(16))) char b0[P2_BS];
    __shared__ __attribute__((aligned(16))) char b1[P2_BS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 128;
    const bool isA = wave < 2;                             // loader role (wave-uniform)
    const int lw = wave & 1;                               // loader index inside its role
    const int nN = (a.N + P2_BN - 1) / P2_BN;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    const int m0 = ((sq / nN) * 8 + xcd) * P2_BM, n0 = (sq % nN) * P2_BN;       // an XCD walks the N tiles of its M panels: A rows stay in its L2
    if (m0 >= a.M) return;
    const char* Ab = reinterpret_cast<const char*>(static_cast<const bf16_t*>(a.A) + (int64_t)m0 * a.lda);
    const char* Bb = reinterpret_cast<const char*>(static_cast<const bf16_t*>(a.B) + (int64_t)n0 * a.ldb);
    const char* Lb = isA ? Ab : Bb;                        // this wave's operand
    const int lmax = isA ? a.M - 1 - m0 : a.N - 1 - n0;
    const uint32_t ld2 = (uint32_t)(isA ? a.lda : a.ldb) * 2u;
    const int nk = a.K / 32;
    // DMA: a wave instruction fills 1 KB = 16 rows x 64 B; lane l -> row l >> 2, physical chunk l & 3.  Loader lw of a role owns the
    // 16-row blocks lw, lw + 2, ... of its operand's stage: 4 of A's 8, 8 of B's 16.
    const int drow = lane >> 2;
    const uint32_t dchunk = (uint32_t)(((lane & 3) ^ ((-(lane >> 4)) & 3)) << 4);
    uint32_t off[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) off[i] = (uint32_t)min((lw + 2 * i) * 16 + drow, lmax) * ld2 + dchunk;
    const int foff = fr * ROW3 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);

    f32x4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t fa[4], fb[8];

#define GTOS_DMA1(src, dst)                                                                                                   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                                    \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
#define GTOS_P2_DMA_A(slot, s_)                                                                                               \
    {                                                                                                                         \
        const int kb_ = min((s_), nk - 1) * 64;            /* byte offset of the stage's k range; past the end: dummy re-read */ \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) GTOS_DMA1(Lb + kb_ + off[i_], (slot) + (lw + 2 * i_) * 1024);        \
    }
#define GTOS_P2_DMA_B(slot, s_)                                                                                               \
    {                                                                                                                         \
        const int kb_ = min((s_), nk - 1) * 64;                                                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) GTOS_DMA1(Lb + kb_ + off[i_], (slot) + (lw + 2 * i_) * 1024);        \
    }
// sa: A slot of stage s_, da: A slot that takes stage s_+5 (= slot of s_-1); sb / db likewise for B's two slots
#define GTOS_P2_STEP(sa, da, sb, db, s_)                                                                                      \
    {                                                                                                                         \
        if (isA) { GTOS_VMCNT(16); } else { GTOS_VMCNT(0); }   /* own pieces of stage s_ (A: 4 newer stages stay in flight) */ \
        __builtin_amdgcn_s_barrier();                      /* everybody's; and every wave has read its fragments of s_-1 */   \
        if (isA) { GTOS_P2_DMA_A(da, (s_) + 5); } else { GTOS_P2_DMA_B(db, (s_) + 1); }                                       \
        {                                                                                                                     \
            const uint32_t ab_ = GTOS_LDS_ADDR(sb) + fbase_b, aa_ = GTOS_LDS_ADDR(sa) + fbase_a;                              \
            GTOS_DSR128(fb[0], ab_, 0); GTOS_DSR128(fb[1], ab_, 1024); GTOS_DSR128(fb[2], ab_, 2048); GTOS_DSR128(fb[3], ab_, 3072);   \
            GTOS_DSR128(fb[4], ab_, 4096); GTOS_DSR128(fb[5], ab_, 5120); GTOS_DSR128(fb[6], ab_, 6144); GTOS_DSR128(fb[7], ab_, 7168); \
            GTOS_DSR128(fa[0], aa_, 0); GTOS_DSR128(fa[1], aa_, 1024); GTOS_DSR128(fa[2], aa_, 2048); GTOS_DSR128(fa[3], aa_, 3072);   \
        }                                                                                                                     \
        __builtin_amdgcn_s_waitcnt(0xc07f);                /* lgkmcnt(0): fragments in registers */                            \
        asm volatile("" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]),   \
                          "+v"(fb[4]), "+v"(fb[5]), "+v"(fb[6]), "+v"(fb[7]));   /* the MFMAs below consume values defined AFTER the wait */ \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                                      \
            _Pragma("unroll") for (int nt = 0; nt < 8; ++nt)                                                                  \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
    }

    // The fragment reads are inline assembly: hipcc's wait-count pass, which cannot know that a wave only ever issues DMA into ITS
    // role's slots, put s_waitcnt vmcnt(0) in front of every other step's ds_read (checked in the ISA) and with it drained the deep
    // A ring.  Reads it does not see get no waits; the waits needed are the explicit ones of GTOS_P2_STEP.
    const uint32_t fbase_a = (uint32_t)(wm * ROW3 + foff), fbase_b = (uint32_t)(wn * ROW3 + foff);
#define GTOS_LDS_ADDR(p_) ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(p_))
#define GTOS_DSR128(dst_, addr_, imm_) asm volatile("ds_read_b128 %0, %1 offset:" #imm_ : "=v"(dst_) : "v"(addr_))

    if (isA) {
        GTOS_P2_DMA_A(a0, 0); GTOS_P2_DMA_A(a1, 1); GTOS_P2_DMA_A(a2, 2); GTOS_P2_DMA_A(a3, 3); GTOS_P2_DMA_A(a4, 4);
    } else {
        GTOS_P2_DMA_B(b0, 0);
    }
    int s = 0;
    for (; s + 6 <= nk; s += 6) {                          // whole sextuples: one path through the body for the wait-count pass
        GTOS_P2_STEP(a0, a5, b0, b1, s);
        GTOS_P2_STEP(a1, a0, b1, b0, s + 1);
        GTOS_P2_STEP(a2, a1, b0, b1, s + 2);
        GTOS_P2_STEP(a3, a2, b1, b0, s + 3);
        GTOS_P2_STEP(a4, a3, b0, b1, s + 4);
        GTOS_P2_STEP(a5, a4, b1, b0, s + 5);
    }
    if (s < nk) {
        GTOS_P2_STEP(a0, a5, b0, b1, s);
        if (s + 1 < nk) {
            GTOS_P2_STEP(a1, a0, b1, b0, s + 1);
            if (s + 2 < nk) {
                GTOS_P2_STEP(a2, a1, b0, b1, s + 2);
                if (s + 3 < nk) {
                    GTOS_P2_STEP(a3, a2, b1, b0, s + 3);
                    if (s + 4 < nk) GTOS_P2_STEP(a4, a3, b0, b1, s + 4);
                }
            }
        }
    }
    GTOS_VMCNT(0);                                         // the dummy prefetches of the last steps
    __syncthreads();                                       // every wave is done with the stages: a0..a2 become the output staging
#undef GTOS_P2_STEP
#undef GTOS_DSR128
#undef GTOS_LDS_ADDR
#undef GTOS_P2_DMA_A
#undef GTOS_P2_DMA_B
#undef GTOS_DMA1
    // ---- epilogue (bf16 out): 16 rows x 128 columns of the wave tile at a time through LDS (wave-private rows), 256-byte row segments out
    bf16_t* C = static_cast<bf16_t*>(a.C);
    const float keep_scale = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
    constexpr int CP = 128 * 2 + 16;                       // bytes per staged row
    char* cs = wave == 0 ? a0 : (wave == 1 ? a1 : (wave == 2 ? a2 : a3));      // 16 x 272 B of a free 8 KB stage slot per wave
    const bool plain = !a.bias && !a.relu && !(a.p_drop > 0.f);
    if (plain && !a.accumulate && m0 + P2_BM <= a.M && n0 + P2_BN <= a.N) {
        // interior tile of a plain product (the relation projections, the gate tables): no per-element conditions, addresses once
        char* wr = cs + fr * CP + fq * 8;
        const char* rd = cs + (lane >> 4) * CP + (lane & 15) * 16;
        bf16_t* cp0 = C + (int64_t)(m0 + wm + (lane >> 4)) * a.ldc + n0 + wn + (lane & 15) * 8;
        const int64_t ld4 = 4 * a.ldc;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                const f32x4_t v = acc[mt][nt];
                *reinterpret_cast<uint2*>(wr + nt * 32) = make_uint2(pack_bf(v[0], v[1]), pack_bf(v[2], v[3]));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0)
            const U128 v0 = *reinterpret_cast<const U128*>(rd), v1 = *reinterpret_cast<const U128*>(rd + 4 * CP);
            const U128 v2 = *reinterpret_cast<const U128*>(rd + 8 * CP), v3 = *reinterpret_cast<const U128*>(rd + 12 * CP);
            __builtin_amdgcn_s_waitcnt(0xc07f);            // read back before the next row block overwrites the staging rows
            *reinterpret_cast<U128*>(cp0 + (mt * 4 + 0) * ld4) = v0;
            *reinterpret_cast<U128*>(cp0 + (mt * 4 + 1) * ld4) = v1;
            *reinterpret_cast<U128*>(cp0 + (mt * 4 + 2) * ld4) = v2;
            *reinterpret_cast<U128*>(cp0 + (mt * 4 + 3) * ld4) = v3;
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            float v[4] = {acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]};
            if (!plain) {
                const int m = m0 + wm + mt * 16 + fr, n = n0 + wn + nt * 16 + fq * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (a.bias && n + i < a.N) v[i] += a.bias[n + i];
                    if (a.relu) v[i] = fmaxf(v[i], 0.f);
                    if (a.p_drop > 0.f)
                        v[i] = drop_keep(a.seed, (uint64_t)m * (uint64_t)a.N + (uint64_t)(n + i), a.p_drop) ? v[i] * keep_scale : 0.f;
                }
            }
            *reinterpret_cast<uint2*>(cs + fr * CP + (nt * 16 + fq * 4) * 2) = make_uint2(pack_bf(v[0], v[1]), pack_bf(v[2], v[3]));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 4 + (lane >> 4), col = (lane & 15) * 8;
            const int m = m0 + wm + mt * 16 + row, n = n0 + wn + col;
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
        __builtin_amdgcn_s_waitcnt(0xc07f);                // the reads are done before the next row block overwrites the staging rows
    }
}

int launch_p2(const GemmArgs& a, hipStream_t s) {
    const long long nMt = (a.M + P2_BM - 1) / P2_BM, nNt = (a.N + P2_BN - 1) / P2_BN;
    const long long nblk = ((nMt + 7) / 8) * 8 * nNt;
    if (nblk > 0x7fffffffLL) return -6;
    hipLaunchKernelGGL(gemm_p2_nt_kernel, dim3((unsigned)nblk), dim3(256), 0, s, a);
    GTOS_CHECK_LAUNCH();
    return 0;
}

// ---- gemm_w4_nt_kernel (round 4, OPT-IN: GTOS_GEMM_W4=1; an experiment that reaches the default kernel's rate, not more): the deep-K
//      forward shape (the K = L*2d bank-gradient slab: 3.65 TFLOP, 3.2-3.5 ms of every C2 step) on the 256 x 256 tile of
//      gemm256p_nt_kernel with FOUR waves of 128 x 128 (all 256 accumulation registers of a lane, one wave per SIMD) instead of eight of
//      128 x 64 -- the shape of the hipBLASLt kernel that reaches 1.33-1.41 PF/s on these products where gemm256p_nt_kernel reaches
//      1.07-1.19.  Five builds, all bit-checked by tests/test_hip_parity.py::test_gemm_deep_k_one_wave_per_simd:
//        1. LDS-DMA three 32-k stages ahead, fragments of the next step in a second register set      0.97-1.02 PF/s
//        2. the same, four stages ahead (the register set as the fifth)                                0.97-1.02
//        3. global loads into registers two steps ahead, ds_write into two LDS slots                   1.01-1.02
//        4. 64-k stages = one whole cache line per matrix row and stage (was: half a line twice)       1.00-1.12 (square 8192^3: 0.63 -> 1.05-1.10)
//        5. A two stages deep in registers, every memory instruction between two MFMAs, one load per
//           8-MFMA group, non-temporal hint on A                                                       1.05-1.14
//      with the measuring switches of the kernel (GTOS_GEMM_W4_DBG): MFMAs + barriers alone 1.81-2.06 PF/s; with the LDS traffic
//      1.59-1.81; with the global loads 1.05-1.19 EVEN WHEN every load hits the L1 (all stages redirected to the lines of stage 0) and
//      whatever their form (global / buffer / LDS-DMA), depth (1-4 stages) or spacing: a 1 KB wave load costs its wave ~60 cycles in
//      which it issues nothing else, 16 of them per 128 MFMAs, and with one wave per SIMD nobody else feeds the MFMA pipe meanwhile.
//      The two-waves-per-SIMD ping-pong of gemm256p_nt_kernel hides about as much as this kernel saves.  Left in the tree as the
//      measured record of that; profiles/r4w_gemm_w4.txt.
//      hipcc cannot allocate a 256-register accumulator (it permutes tiles through VGPRs and scratch at the loop header, and the scratch
//      traffic breaks hand-counted vmcnt waits): the MFMAs name a0..a255 themselves (gemm_w4_gen.h, written by tools/gen_gemm_w4.py;
//      tools/check_gemm_w4_isa.py checks that the compiler stays out of those registers).
const bool g_use_w4 = getenv("GTOS_GEMM_W4") && getenv("GTOS_GEMM_W4")[0] == '1';
const int g_w4_mink = getenv("GTOS_GEMM_W4_MINK") ? atoi(getenv("GTOS_GEMM_W4_MINK")) : 2048;

constexpr int W4_ROW = 128;                                // bytes per staged row: 64 k, one whole cache line per row and stage
constexpr int W4_A = 256 * W4_ROW;                         // bytes of the A half of a stage (the B half follows)
constexpr int W4_ST = 2 * W4_A;                            // 64 KB per stage

template <int DBG> __global__ __launch_bounds__(256, 1) void gemm_w4_nt_kernel(GemmArgs a) {
    if (a.p_drop > 0.f) a.seed = live_seed(a.seed);
    // Two LDS slots of one 64-k stage each (stage t in slot t & 1).  A stage row is 128 B = ONE whole cache line per matrix row: with
    // the 64-byte rows of the 32-k stages of gemm256p_nt_kernel every line is requested twice, in consecutive steps, through a vector
    // L1 that a step's 64 KB of lines has flushed in between -- measured on this kernel's first builds: MFMAs + barriers alone 1.9-2.06
    // PF/s, with the LDS traffic 1.54-1.69, with the global loads 0.63-1.02 whatever the prefetch depth (LDS-DMA three or four stages
    // ahead, or registers two steps ahead).  The stage in flight lives in REGISTERS (16 x 16 B per lane) and reaches the LDS by
    // ds_write one step after its loads were issued.
    __shared__ __attribute__((aligned(16))) char st[2 * W4_ST];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int wm = (wave >> 1) * 128, wn = (wave & 1) * 128;
    const int nN = (a.N + BN2 - 1) / BN2;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    const int m0 = ((sq / nN) * 8 + xcd) * BM2, n0 = (sq % nN) * BN2;
    if (m0 >= a.M) return;
    const char* Ab = reinterpret_cast<const char*>(static_cast<const bf16_t*>(a.A) + (int64_t)m0 * a.lda);
    const char* Bb = reinterpret_cast<const char*>(static_cast<const bf16_t*>(a.B) + (int64_t)n0 * a.ldb);
    const int amax = a.M - 1 - m0, bmax = a.N - 1 - n0;
    const uint32_t lda2 = (uint32_t)a.lda * 2u, ldb2 = (uint32_t)a.ldb * 2u;
    const int nk = a.K / 64;
    // a wave instruction moves 1 KB = 8 rows x 128 B: lane l -> row l >> 3 of the piece; its 16 B land at lane * 16 of the piece
    // (physical chunk l & 7) and come from logical chunk (l & 7) ^ (row & 7) of the line.  Wave w owns the 8-row pieces w, w + 4, ...,
    // w + 28 of each operand's 32.
    const int drow = lane >> 3;
    const uint32_t dchunk = (uint32_t)(((lane & 7) ^ drow) << 4);
    uint32_t goff[16];                                     // pieces 0..7: A, 8..15: B
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        goff[i] = (uint32_t)min((wave + 4 * i) * 8 + drow, amax) * lda2 + dchunk;
        goff[8 + i] = (uint32_t)min((wave + 4 * i) * 8 + drow, bmax) * ldb2 + dchunk;
    }
#define GTOS_LDS_ADDR(p_) ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(p_))
    // fragment of 16 rows x 32 k (k half h of the stage): lane (fr, fq) reads logical chunk h * 4 + fq of row fr
    const uint32_t fsw0 = (uint32_t)(fr * W4_ROW + ((fq ^ (fr & 7)) << 4)), fsw1 = (uint32_t)(fr * W4_ROW + (((4 + fq) ^ (fr & 7)) << 4));
    const uint32_t lds0 = GTOS_LDS_ADDR(st);
    const uint32_t wbase = lds0 + (uint32_t)(wave * 1024 + lane * 16);    // + 4096 * i for A piece i, + W4_A + 4096 * i for B piece i

    // the 128 x 128 fp32 wave tile lives in a0..a255, named by the inline assembly of gemm_w4_gen.h (tile (mt, nt) = a[(mt*8+nt)*4 ..]);
    // the compiler never sees an accumulator value, and must not use an accumulation register of its own in this kernel
    // (tools/check_gemm_w4_isa.py checks the compiled kernel for that)
    GTOS_W4_ZERO();
    bf16x8_t fa[8], fbx[8], fby[8];                        // A fragments: one set, refilled row block by row block; B: two sets
    // stages in flight, in registers: A two steps deep (set t & 1 holds the lane's 8 pieces of stage t), B one step.  A is the operand
    // that streams from HBM (in the products this kernel is for, B is a weight matrix that stays in the L2); with one 64 KB stage in
    // flight per CU the loads set the pace -- all stages redirected to the lines of stage 0 (L1 hits) or of stages 0..7 (L2 hits): 1.8-2.0
    // PF/s; real addresses: 1.0-1.1.
    U128 ga0[8], ga1[8], gb[8];

// All memory instructions of the loop are inline assembly, issued where they are written, with the waits written out: hipcc sinks
// ordinary LDS reads to just before their use and drains lgkmcnt(0) there (four exposed LDS latencies per step in the first build).
#define GTOS_DSR128(dst_, addr_, imm_) if (DBG != 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "n"(imm_))
#define GTOS_DSW128(addr_, src_, imm_) if (DBG != 2) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr_), "v"(src_), "n"(imm_) : "memory")
#define GTOS_GLD128(dst_, off_, base_) if (DBG != 1 && DBG != 2) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst_) : "v"(off_), "s"(base_))
// A streams through the L2 once per block row: its loads carry the non-temporal hint (worth 0-7 % on the bank-gradient product)
#define GTOS_GLD128A(dst_, off_, base_) if (DBG != 1 && DBG != 2) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst_) : "v"(off_), "s"(base_))
#define GTOS_GLD128B(dst_, off_, base_) GTOS_GLD128(dst_, off_, base_)
// DBG (GTOS_GEMM_W4_DBG, a measuring aid; results are garbage for DBG != 0): 1 = no global loads, 2 = no LDS traffic either (MFMAs +
// barriers), 4 = every stage loads the k range of stage 0 (L1 hits)
#define GTOS_W4_KB(s_) (DBG == 4 ? 0 : min((s_), nk - 1) * 128)          /* byte offset of a stage's k range; past the end: dummy re-read */
#define GTOS_W4_KBA(s_) GTOS_W4_KB(s_)
#define GTOS_W4_KBB(s_) GTOS_W4_KB(s_)
#define GTOS_W4_LOAD_A(GA, s_)                                                                                                \
    {                                                                                                                         \
        const char* p_ = Ab + GTOS_W4_KBA(s_);                                                                                \
        GTOS_GLD128A(GA[0], goff[0], p_); GTOS_GLD128A(GA[1], goff[1], p_); GTOS_GLD128A(GA[2], goff[2], p_); GTOS_GLD128A(GA[3], goff[3], p_); \
        GTOS_GLD128A(GA[4], goff[4], p_); GTOS_GLD128A(GA[5], goff[5], p_); GTOS_GLD128A(GA[6], goff[6], p_); GTOS_GLD128A(GA[7], goff[7], p_); \
    }
#define GTOS_W4_LOAD_B(s_)                                                                                                    \
    {                                                                                                                         \
        const char* p_ = Bb + GTOS_W4_KBB(s_);                                                                                \
        GTOS_GLD128B(gb[0], goff[8], p_); GTOS_GLD128B(gb[1], goff[9], p_); GTOS_GLD128B(gb[2], goff[10], p_); GTOS_GLD128B(gb[3], goff[11], p_);   \
        GTOS_GLD128B(gb[4], goff[12], p_); GTOS_GLD128B(gb[5], goff[13], p_); GTOS_GLD128B(gb[6], goff[14], p_); GTOS_GLD128B(gb[7], goff[15], p_); \
    }
#define GTOS_W4_DEP8(F) asm volatile("" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7]))
// group mt of the FIRST half step (k half 0 of stage s in fa / fbx): B fragment mt of k half 1 of the same stage into fby; A piece mt
// and B piece mt of stage s+1 from their registers into the other slot; the load of B piece mt-1 of stage s+2 into the registers
// written one group earlier; the row block's A fragment of k half 1 over the one just consumed.  Group mt of the SECOND half step
// (k half 1 of stage s): B fragment mt and A fragment mt of k half 0 of stage s+1, and the load of A piece mt of stage s+3.  One load
// per group: the four waves of a workgroup run in lock step, and 16 loads per wave inside one half step keep the CU's address unit
// busy for that whole half step (every wave stalls at its issue), while the other half step sends it nothing.
// The loads of a step are issued B(s+2) b0..7, then A(s+3) a0..7; at the write of A piece mt the loads that may stay in flight are
// 7 - mt + B(s+1) + A(s+2) + mt - 1 = 22, at the write of B piece mt 7 - mt + A(s+2) + mt - 1 = 14.
// Every memory instruction sits between two MFMAs: a wave issues in order, and while it issues memory instructions in a row its SIMD's
// MFMA pipe, which nobody else feeds in this kernel, runs dry.
#define GTOS_W4_GROUP0(mt, GA)                                                                                                \
    {                                                                                                                         \
        GTOS_W4_MFMA(mt, 0, fa[mt], fbx);                                                                                     \
        GTOS_DSR128(fby[mt], rb1_, (mt) * 2048);                                                                              \
        GTOS_W4_MFMA(mt, 1, fa[mt], fbx);                                                                                     \
        GTOS_VMCNT(22);                                                                                                       \
        asm volatile("" : "+v"(GA[mt]));                                                                                      \
        GTOS_DSW128(w_, GA[mt], (mt) * 4096);                                                                                 \
        GTOS_W4_MFMA(mt, 2, fa[mt], fbx);                                                                                     \
        GTOS_VMCNT(14);                                                                                                       \
        asm volatile("" : "+v"(gb[mt]));                                                                                      \
        GTOS_DSW128(w_, gb[mt], W4_A + (mt) * 4096);                                                                          \
        GTOS_W4_MFMA(mt, 3, fa[mt], fbx);                                                                                     \
        if ((mt) > 0) GTOS_GLD128B(gb[(mt) > 0 ? (mt) - 1 : 0], goff[8 + ((mt) > 0 ? (mt) - 1 : 0)], pb_);                    \
        GTOS_W4_MFMA(mt, 4, fa[mt], fbx);                                                                                     \
        GTOS_W4_MFMA(mt, 5, fa[mt], fbx);                                                                                     \
        GTOS_W4_MFMA(mt, 6, fa[mt], fbx);                                                                                     \
        GTOS_W4_MFMA(mt, 7, fa[mt], fbx);                                                                                     \
        GTOS_DSR128(fa[mt], ra1_, (mt) * 2048);                                                                               \
    }
#define GTOS_W4_GROUP1(mt, GA)                                                                                                \
    {                                                                                                                         \
        GTOS_W4_MFMA(mt, 0, fa[mt], fby);                                                                                     \
        GTOS_DSR128(fbx[mt], nb0_, (mt) * 2048);                                                                              \
        GTOS_W4_MFMA(mt, 1, fa[mt], fby);                                                                                     \
        GTOS_GLD128A(GA[mt], goff[mt], pa_);                                                                                  \
        GTOS_W4_MFMA(mt, 2, fa[mt], fby); GTOS_W4_MFMA(mt, 3, fa[mt], fby); GTOS_W4_MFMA(mt, 4, fa[mt], fby);                 \
        GTOS_W4_MFMA(mt, 5, fa[mt], fby); GTOS_W4_MFMA(mt, 6, fa[mt], fby); GTOS_W4_MFMA(mt, 7, fa[mt], fby);                 \
        GTOS_DSR128(fa[mt], na0_, (mt) * 2048);                                                                               \
    }
// step s_: k half 0 of stage s_ is in fa / fbx.  First half step: MFMAs on it; k half 1 of stage s_ (slot s_ & 1) into fa / fby; stage
// s_+1 from the registers (GA = its A set) into slot (s_+1) & 1 (free since the middle of step s_-1); loads of A(s_+3), B(s_+2).
// Barrier.  Second half step: MFMAs on fa / fby; k half 0 of stage s_+1 into fa / fbx.
#define GTOS_W4_STEP(s_, GA)                                                                                                  \
    {                                                                                                                         \
        const uint32_t cur = lds0 + (uint32_t)(((s_) & 1) * W4_ST), nxt = lds0 + (uint32_t)((((s_) + 1) & 1) * W4_ST);        \
        const uint32_t ra1_ = cur + (uint32_t)(wm * W4_ROW) + fsw1, rb1_ = cur + (uint32_t)(W4_A + wn * W4_ROW) + fsw1;      \
        const uint32_t na0_ = nxt + (uint32_t)(wm * W4_ROW) + fsw0, nb0_ = nxt + (uint32_t)(W4_A + wn * W4_ROW) + fsw0;      \
        const uint32_t w_ = wbase + (uint32_t)((((s_) + 1) & 1) * W4_ST);                                                     \
        const char* pa_ = Ab + GTOS_W4_KBA((s_) + 3);                                                                         \
        const char* pb_ = Bb + GTOS_W4_KBB((s_) + 2);                                                                         \
        GTOS_W4_GROUP0(0, GA);                                                                                                \
        __builtin_amdgcn_s_waitcnt(0xc47f);                /* lgkmcnt(4): the refill of fa[7], the previous half step's last read */ \
        GTOS_W4_GROUP0(1, GA);                                                                                                \
        GTOS_W4_GROUP0(2, GA);                                                                                                \
        GTOS_W4_GROUP0(3, GA);                                                                                                \
        GTOS_W4_GROUP0(4, GA);                                                                                                \
        GTOS_W4_GROUP0(5, GA);                                                                                                \
        GTOS_W4_GROUP0(6, GA);                                                                                                \
        GTOS_W4_GROUP0(7, GA);                                                                                                \
        GTOS_GLD128B(gb[7], goff[15], pb_);                                                                                   \
        __builtin_amdgcn_s_waitcnt(0xc17f);                /* lgkmcnt(1): all but the refill of fa[7] -- own writes, fby, fa[0..6] */ \
        GTOS_W4_DEP8(fa);                                                                                                     \
        GTOS_W4_DEP8(fby);                                                                                                    \
        __builtin_amdgcn_s_barrier();                      /* stage s_+1 is in its slot for everybody */                       \
        GTOS_W4_GROUP1(0, GA);                                                                                                  \
        __builtin_amdgcn_s_waitcnt(0xc27f);                /* lgkmcnt(2): the refill of fa[7] */                               \
        GTOS_W4_GROUP1(1, GA);                                                                                                  \
        GTOS_W4_GROUP1(2, GA);                                                                                                  \
        GTOS_W4_GROUP1(3, GA);                                                                                                  \
        GTOS_W4_GROUP1(4, GA);                                                                                                  \
        GTOS_W4_GROUP1(5, GA);                                                                                                  \
        GTOS_W4_GROUP1(6, GA);                                                                                                  \
        GTOS_W4_GROUP1(7, GA);                                                                                                  \
        __builtin_amdgcn_s_waitcnt(0xc17f);                /* lgkmcnt(1) */                                                    \
        GTOS_W4_DEP8(fa);                                                                                                     \
        GTOS_W4_DEP8(fbx);                                                                                                    \
    }

    GTOS_W4_LOAD_A(ga0, 0);
    GTOS_W4_LOAD_B(0);
    GTOS_VMCNT(0);
    GTOS_W4_DEP8(ga0);
    GTOS_W4_DEP8(gb);
    GTOS_DSW128(wbase, ga0[0], 0); GTOS_DSW128(wbase, ga0[1], 4096); GTOS_DSW128(wbase, ga0[2], 8192); GTOS_DSW128(wbase, ga0[3], 12288);
    GTOS_DSW128(wbase, ga0[4], 16384); GTOS_DSW128(wbase, ga0[5], 20480); GTOS_DSW128(wbase, ga0[6], 24576); GTOS_DSW128(wbase, ga0[7], 28672);
    GTOS_DSW128(wbase, gb[0], W4_A); GTOS_DSW128(wbase, gb[1], W4_A + 4096); GTOS_DSW128(wbase, gb[2], W4_A + 8192); GTOS_DSW128(wbase, gb[3], W4_A + 12288);
    GTOS_DSW128(wbase, gb[4], W4_A + 16384); GTOS_DSW128(wbase, gb[5], W4_A + 20480); GTOS_DSW128(wbase, gb[6], W4_A + 24576); GTOS_DSW128(wbase, gb[7], W4_A + 28672);
    // the load queue the first step expects: A(1), B(1), A(2)
    GTOS_W4_LOAD_A(ga1, 1);
    GTOS_W4_LOAD_B(1);
    GTOS_W4_LOAD_A(ga0, 2);
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // own writes of stage 0 done
    __builtin_amdgcn_s_barrier();
    {
        const uint32_t ra_ = lds0 + (uint32_t)(wm * W4_ROW) + fsw0, rb_ = lds0 + (uint32_t)(W4_A + wn * W4_ROW) + fsw0;
        GTOS_DSR128(fbx[0], rb_, 0); GTOS_DSR128(fbx[1], rb_, 2048); GTOS_DSR128(fbx[2], rb_, 4096); GTOS_DSR128(fbx[3], rb_, 6144);
        GTOS_DSR128(fbx[4], rb_, 8192); GTOS_DSR128(fbx[5], rb_, 10240); GTOS_DSR128(fbx[6], rb_, 12288); GTOS_DSR128(fbx[7], rb_, 14336);
        GTOS_DSR128(fa[0], ra_, 0); GTOS_DSR128(fa[1], ra_, 2048); GTOS_DSR128(fa[2], ra_, 4096); GTOS_DSR128(fa[3], ra_, 6144);
        GTOS_DSR128(fa[4], ra_, 8192); GTOS_DSR128(fa[5], ra_, 10240); GTOS_DSR128(fa[6], ra_, 12288); GTOS_DSR128(fa[7], ra_, 14336);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    GTOS_W4_DEP8(fa);
    GTOS_W4_DEP8(fbx);
    int s = 0;
    for (; s + 2 <= nk; s += 2) {                          // A(s+1) lives in register set (s+1) & 1
        GTOS_W4_STEP(s, ga1);
        GTOS_W4_STEP(s + 1, ga0);
    }
    if (s < nk) GTOS_W4_STEP(s, ga1);
    GTOS_VMCNT(0);                                         // the dummy prefetches of the last steps
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // the destination registers of those last loads and reads stay allocated until here: an output nobody reads is a dead value to the
    // compiler, which hands its registers to something else while the load is still in flight (the odd-stage-count tail of the second
    // build put a B fragment where a prefetch landed)
    GTOS_W4_DEP8(ga0); GTOS_W4_DEP8(ga1); GTOS_W4_DEP8(gb); GTOS_W4_DEP8(fa); GTOS_W4_DEP8(fbx); GTOS_W4_DEP8(fby);
    __syncthreads();                                       // every wave is done with the slots: they become the output staging
#undef GTOS_W4_STEP
#undef GTOS_W4_GROUP1
#undef GTOS_W4_GROUP0
#undef GTOS_W4_DEP8
#undef GTOS_W4_LOAD_B
#undef GTOS_W4_LOAD_A
#undef GTOS_GLD128B
#undef GTOS_GLD128A
#undef GTOS_W4_KBB
#undef GTOS_W4_KBA
#undef GTOS_W4_KB
#undef GTOS_GLD128
#undef GTOS_DSW128
#undef GTOS_DSR128
#undef GTOS_LDS_ADDR

    // ---- epilogue (bf16 out): 32 rows x 128 columns of the wave tile at a time through the wave's own stage slot
    bf16_t* C = static_cast<bf16_t*>(a.C);
    const float keep_scale = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
    constexpr int CP = 128 * 2 + 16;
    char* cs = st + wave * 16384;                          // 32 x 272 B per wave
    const bool plain = !a.bias && !a.relu && !(a.p_drop > 0.f);
    // MFMA results are not interlocked against v_accvgpr_read issued from inline assembly: the loop's last MFMAs retire first
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
    auto put = [&](int mt, int nt, float v0, float v1, float v2, float v3) {
        float v[4] = {v0, v1, v2, v3};
        if (!plain) {
            const int m = m0 + wm + mt * 16 + fr, n = n0 + wn + nt * 16 + fq * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (a.bias && n + i < a.N) v[i] += a.bias[n + i];
                if (a.relu) v[i] = fmaxf(v[i], 0.f);
                if (a.p_drop > 0.f)
                    v[i] = drop_keep(a.seed, (uint64_t)m * (uint64_t)a.N + (uint64_t)(n + i), a.p_drop) ? v[i] * keep_scale : 0.f;
            }
        }
        *reinterpret_cast<uint2*>(cs + ((mt & 1) * 16 + fr) * CP + (nt * 16 + fq * 4) * 2) = make_uint2(pack_bf(v[0], v[1]), pack_bf(v[2], v[3]));
    };
    auto flush = [&](int q4) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0)
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int row = pass * 4 + (lane >> 4), col = (lane & 15) * 8;
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
    };
    GTOS_W4_TILES_0(put); flush(0);
    GTOS_W4_TILES_1(put); flush(1);
    GTOS_W4_TILES_2(put); flush(2);
    GTOS_W4_TILES_3(put); flush(3);
}

int launch_w4(const GemmArgs& a, hipStream_t s) {
    const long long nMt = (a.M + BM2 - 1) / BM2, nNt = (a.N + BN2 - 1) / BN2;
    const long long nblk = ((nMt + 7) / 8) * 8 * nNt;
    if (nblk > 0x7fffffffLL) return -6;
    static const int dbg = getenv("GTOS_GEMM_W4_DBG") ? atoi(getenv("GTOS_GEMM_W4_DBG")) : 0;
    if (dbg == 1) hipLaunchKernelGGL(gemm_w4_nt_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    else if (dbg == 2) hipLaunchKernelGGL(gemm_w4_nt_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    else if (dbg == 4) hipLaunchKernelGGL(gemm_w4_nt_kernel<4>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(gemm_w4_nt_kernel<0>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    GTOS_CHECK_LAUNCH();
    return 0;
}

int launch256p(const GemmArgs& a, hipStream_t s) {
    const long long nMt = (a.M + BM2 - 1) / BM2, nNt = (a.N + BN2 - 1) / BN2;
    const long long nblk = ((nMt + 7) / 8) * 8 * nNt;
    if (nblk > 0x7fffffffLL) return -6;
    hipLaunchKernelGGL(gemm256p_nt_kernel, dim3((unsigned)nblk), dim3(512), 0, s, a);
    GTOS_CHECK_LAUNCH();
    return 0;
}

// ---- gemm256p_tn_kernel: the weight-gradient shape dW = A^T B (A [K, M], B [K, N] row-major, K in the hundreds of
//      thousands, split-K into the fp32 partial-tile workspace) on the same 256x256 tile, four 32-k stages and ping-pong
//      schedule as gemm256p_nt_kernel.  A 256x256 tile re-reads each operand row from L2 half as often as the 128x128
//      tile (these products stream both operands from HBM once and are bound by the L2 -> LDS amplification).
//      A stage holds [32 k][256 m] + [32 k][256 n] (512-byte k-rows); MFMA fragments (8 consecutive k of one column) come
//      from ds_read_b64_tr_b16 as in gemm_kernel's transpose path: 32-byte granules XOR-swizzled with (k & 3), applied
//      on the DMA's source address.  Rows k >= kend read a block of zeros; columns past M / N re-read the last 8 valid
//      ones (never stored); stages past the end of the split re-read its last stage (never multiplied).
__global__ __launch_bounds__(512) void gemm256p_tn_kernel(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) char st0[ST3];
    __shared__ __attribute__((aligned(16))) char st1[ST3];
    __shared__ __attribute__((aligned(16))) char st2[ST3];
    __shared__ __attribute__((aligned(16))) char st3[ST3];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
    const int nN = (a.N + BN2 - 1) / BN2, nM = (a.M + BM2 - 1) / BM2, tiles = nM * nN;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    const int bz = (sq / tiles) * 8 + xcd, t_ = sq % tiles;        // all tiles of one K split on one XCD
    if (bz >= a.splitk) return;
    const int m0 = (t_ / nN) * BM2, n0 = (t_ % nN) * BN2;
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
    const int acol = min(m0 + lp * 8, a.M - 8), bcol = min(n0 + lp * 8, a.N - 8);
    // running per-lane source pointers of the NEXT stage to fetch (k-row kcur and kcur + 16 of each operand); rows at or
    // past kend -- also every row of the dummy stages after the split's last one -- read the block of zeros instead
    const char* pa = reinterpret_cast<const char*>(static_cast<const bf16_t*>(a.A) + (int64_t)(kbeg + kl0) * a.lda + acol);
    const char* pb = reinterpret_cast<const char*>(static_cast<const bf16_t*>(a.B) + (int64_t)(kbeg + kl0) * a.ldb + bcol);
    const int64_t a16 = 32 * a.lda, b16 = 32 * a.ldb;      // bytes: 16 k-rows
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

    // ---- this split's partial tile -> workspace (N % 4 == 0: plain 16-byte stores), reduced by splitk_reduce_kernel
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int m = m0 + wm + mt * 16 + fr;
        if (m >= a.M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn + nt * 16 + fq * 4;
            if (n >= a.N) continue;
            *reinterpret_cast<float4*>(a.ws + ((int64_t)bz * a.M + m) * a.N + n) =
                make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
        }
    }
}

int launch256p_tn(const GemmArgs& a, hipStream_t s) {
    const long long tiles = ((long long)(a.M + BM2 - 1) / BM2) * ((a.N + BN2 - 1) / BN2);
    const long long nblk = tiles * 8 * ((a.splitk + 7) / 8);
    if (nblk > 0x7fffffffLL) return -6;
    hipLaunchKernelGGL(gemm256p_tn_kernel, dim3((unsigned)nblk), dim3(512), 0, s, a);
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
        if (g_use_w4 && !transA && transB && a.vecA && a.vecB && a.vecC && splitk == 1 && N >= 256 && K % 64 == 0 && K >= g_w4_mink &&
            lda < (1 << 22) && ldb < (1 << 22) && t256 >= 512)
            return launch_w4(a, s);
        if (g_use_pipe && !transA && transB && a.vecA && a.vecB && a.vecC && splitk == 1 && N >= 256 && N <= g_max_npipe && K % 32 == 0 &&
            lda < (1 << 22) && ldb < (1 << 22) &&
            ((t256 >= 512 && K >= g_min_kpipe) || (t256 <= g_pipe_small_max && K >= 256 && M >= 256)))
            return launch256p(a, s);
        // short reduction, many tiles: two pipelined workgroups per CU, one's output write beside the other's k loop
        if (g_use_p2 && !transA && transB && a.vecA && a.vecB && a.vecC && splitk == 1 && N >= 256 && K % 32 == 0 && K >= 128 && K <= g_p2_maxk &&
            lda < (1 << 22) && ldb < (1 << 22) && ((long long)(M + P2_BM - 1) / P2_BM) * ((N + P2_BN - 1) / P2_BN) >= 1024)
            return launch_p2(a, s);
        if (g_use256 && !transA && transB && a.vecA && a.vecB && a.vecC && splitk == 1 && t256 >= 1024 && N >= 256 && K >= g_min_k256)
            return launch256(a, s);
        return launch<bf16_t, bf16_t>(a, transA, transB, s);
    }
    if (in_dtype == GTOS_BF16 && out_dtype == GTOS_F32)  return launch<bf16_t, float>(a, transA, transB, s);
    if (in_dtype == GTOS_F32 && out_dtype == GTOS_F32)   return launch<float, float>(a, transA, transB, s);
    return -1;
}

GTOS_SEED_EPOCH_SETTER(gemm)
