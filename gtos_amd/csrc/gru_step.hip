// Fused forward GRU time step for gfx950 (bf16): gate products on MFMA with the cell in the epilogue.
//
// Replaces, for one time step of nn.GRU as RelationEncoder uses it (/root/reference/generator/encoder.py:93-111),
//   xg = x W_ih^T + b_ih        (optional: HAS_X; otherwise xg is read)
//   hg = h W_hh^T + b_hh
//   r = s(xg_r + hg_r), z = s(xg_z + hg_z), n = tanh(xg_n + r * hg_n), h' = (1 - z) n + z h
// without the [rows, 3*hs] round trips through HBM: a 256-thread workgroup owns 128 rows x 64 hidden channels, i.e. the
// r, z and n columns of those channels (3 x 64 weight rows per operand tile), accumulates x- and h-products in four
// fp32 accumulator groups (r, z, n_x, n_h) and applies the gates in registers.
//
// Layout tricks:
//   * 4 waves stacked along the rows (32 rows x 192 columns each): a lane ends up with 16 CONSECUTIVE channels of a
//     row (32 bytes) and the 4 lanes of a row with a full 128-byte segment, because the weight rows are loaded into
//     the LDS tile in the order  lds_row = group*64 + nt*16 + q*4 + e  <->  channel = q*16 + nt*4 + e
//     (nt = 16-column MFMA block, lane quad q holds rows q*4+e of the swapped-operand 16x16 result);
//   * operands K-contiguous in LDS (128-byte rows, XOR chunk swizzle as in gemm.hip), filled by global_load_lds_dwordx4;
//   * one 40 KB LDS stage; the resident workgroups of a CU cover each other's load latency;
//   * XCD-aware tile order: the 4 channel tiles of a row panel run back to back on one XCD (h/x rows shared in its L2).
// The new state of row m goes to h_out[m] when m < n_out (the packed "previous state" slot of the next time step, which
// is the next launch's A operand) and to h_fin[m] otherwise (sequence finished) -- no separate h_prev copy.
#include <cstdlib>
#include "common.h"

__attribute__((visibility("hidden"))) const void* gtos_zero_block();      // gemm.hip: 256 zero bytes in global memory

#define GTOS_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
// vmcnt(n) AND lgkmcnt(0): in front of a barrier after which somebody's LDS-DMA overwrites a slot this wave has been reading.  hipcc sinks the
// MFMAs of a stage -- and with them the lgkmcnt wait of their fragment reads -- below the next s_barrier; the reads would then still be
// queued when another wave, past the barrier, starts the DMA into that slot.  (Found by test_gru_backward_step_input_gradient_role_vs_torch
// on the backward step's pipelined k loop, whose weight rows come back from L1 within the time a queued ds_read waits: 1,200 of 40,000 rows
// differed from run to run.  The forward kernels' operands take an L2 / HBM round trip and never showed it; they wait the same way now.)
#ifndef GTOS_RACE_DEMO
#define GTOS_VMCNT_LDS(n) __builtin_amdgcn_s_waitcnt(0x0070 | ((n) & 15) | (((n) >> 4) << 14))
#define GTOS_LGKM0() __builtin_amdgcn_s_waitcnt(0xc07f)
#else      // tools/race_demo.sh only: the waits as they were before that fix, to show tests/test_zz_race_soak.py catching the race
#define GTOS_VMCNT_LDS(n) GTOS_VMCNT(n)
#define GTOS_LGKM0()
#endif

namespace {

constexpr int ROWB = 128, BK = 64, TM = 128, TC = 64, WROWS = 3 * TC;
constexpr int A_BYTES = TM * ROWB, B_BYTES = WROWS * ROWB;

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 3); }
__device__ __forceinline__ int lds_off(int r, int c) { return r * ROWB + ((c ^ swz(r)) << 4); }

struct StepArgs {
    const bf16_t* x; int64_t ldx; int in_dim; const bf16_t* w_ih; const float* b_ih;
    const bf16_t* xg;
    const bf16_t* gf; const int* gf_idx; const bf16_t* gb; const int* gb_idx;     // MODE 2: xg = gf[gf_idx[m]] + gb[gb_idx[m]] + b_ih
    const bf16_t* h_in; const int* h_idx; const bf16_t* w_hh; const float* b_hh;   // h_idx: row m enters with h_in[h_idx[m]]
    bf16_t* h_out; int n_out; bf16_t* h_fin; bf16_t* gates; bf16_t* y; int64_t ldy;
    int64_t ld_fin; const int* fin_idx;            // a finished row m goes to h_fin[(fin_idx ? fin_idx[m] : m) * ld_fin + channel]
    float p_drop; uint64_t seed; int64_t drop_base;
    int rows, hs; const void* zeros;
};

// 128 activation rows x 64 k
__device__ __forceinline__ void dma_rows(const bf16_t* __restrict__ base, const U128* __restrict__ zeros, int64_t ld, int rows_total,
                                         int row0, int k0, int kend, char* tile, int wave, int lane,
                                         const int* __restrict__ gather = nullptr) {
#pragma unroll
    for (int it = 0; it < TM / 32; ++it) {
        const int blk = it * 4 + wave, rl = blk * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz(rl);
        const int r = row0 + rl, k = k0 + c * 8;
        const bool ok = r < rows_total && k + 8 <= kend;
        const int64_t sr = (gather && ok) ? gather[r] : r;             // optional row gather (trie parent / state row)
        const void* src = ok ? static_cast<const void*>(base + sr * ld + k) : static_cast<const void*>(zeros);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(tile + blk * 1024), 16, 0, 0);
    }
}

// 3 x 64 weight rows (gates r, z, n of channels c0..c0+63) x 64 k, in the permuted row order described above
__device__ __forceinline__ void dma_weights(const bf16_t* __restrict__ w, const U128* __restrict__ zeros, int64_t ld, int hs,
                                            int c0, int k0, int kend, char* tile, int wave, int lane) {
#pragma unroll
    for (int it = 0; it < WROWS / 32; ++it) {
        const int blk = it * 4 + wave, rl = blk * 8 + (lane >> 3);
        const int g = rl >> 6, nt = (rl >> 4) & 3, q = (rl >> 2) & 3, e = rl & 3;
        const int wrow = g * hs + c0 + q * 16 + nt * 4 + e;
        const int c = (lane & 7) ^ swz(rl);
        const int k = k0 + c * 8;
        const bool ok = k + 8 <= kend;
        const void* src = ok ? static_cast<const void*>(w + (int64_t)wrow * ld + k) : static_cast<const void*>(zeros);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(tile + blk * 1024), 16, 0, 0);
    }
}

// The same two loaders with the per-lane part of the source addresses computed ONCE per operand instead of once per k
// tile (row index -- through the gather table when there is one: that was a dependent global load in front of every
// DMA --, swizzled chunk, bounds): per k tile only the k bound of a partial last tile is left.
struct RowSrc { const char* p[TM / 32]; int c[TM / 32]; };
__device__ __forceinline__ RowSrc row_src(const bf16_t* __restrict__ base, int64_t ld, int rows_total, int row0, int wave, int lane,
                                          const int* __restrict__ gather = nullptr, int zero_row = -1) {
    RowSrc s;
#pragma unroll
    for (int it = 0; it < TM / 32; ++it) {
        const int rl = (it * 4 + wave) * 8 + (lane >> 3);
        const int r = row0 + rl;
        s.c[it] = (lane & 7) ^ swz(rl);
        const bool ok = r < rows_total;
        const int64_t sr = (gather && ok) ? gather[r] : r;
        // null: rows past the end, and rows whose source is the all-zero row (trie leaves: half of all nodes would otherwise
        // hammer ONE 2 KB row of HBM)
        s.p[it] = (ok && sr != zero_row) ? reinterpret_cast<const char*>(base + sr * ld + s.c[it] * 8) : nullptr;
    }
    return s;
}
__device__ __forceinline__ void dma_rows_at(const RowSrc& s, const U128* __restrict__ zeros, int k0, int kend, char* tile, int wave) {
#pragma unroll
    for (int it = 0; it < TM / 32; ++it) {
        const bool ok = s.p[it] != nullptr && k0 + s.c[it] * 8 + 8 <= kend;
        const void* src = ok ? static_cast<const void*>(s.p[it] + (int64_t)k0 * 2) : static_cast<const void*>(zeros);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(tile + (it * 4 + wave) * 1024), 16, 0, 0);
    }
}
// the same for operands with many zero rows: a lane whose row is null writes its 16 bytes of zeros into its LDS slot itself
// (the slot the DMA of an active lane would fill: tile + block * 1024 + lane * 16) instead of fetching them
__device__ __forceinline__ void dma_rows_at_z(const RowSrc& s, int k0, char* tile, int wave, int lane) {
#pragma unroll
    for (int it = 0; it < TM / 32; ++it) {
        char* blk = tile + (it * 4 + wave) * 1024;
        if (s.p[it] != nullptr)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s.p[it] + (int64_t)k0 * 2),
                                             (__attribute__((address_space(3))) void*)blk, 16, 0, 0);
        else
            *reinterpret_cast<uint4*>(blk + lane * 16) = make_uint4(0, 0, 0, 0);
    }
}
struct WSrc { uint32_t off[WROWS / 32]; int c[WROWS / 32]; };
__device__ __forceinline__ WSrc w_src(int64_t ld, int hs, int c0, int wave, int lane) {
    WSrc s;
#pragma unroll
    for (int it = 0; it < WROWS / 32; ++it) {
        const int rl = (it * 4 + wave) * 8 + (lane >> 3);
        const int g = rl >> 6, nt = (rl >> 4) & 3, q = (rl >> 2) & 3, e = rl & 3;
        const int wrow = g * hs + c0 + q * 16 + nt * 4 + e;
        s.c[it] = (lane & 7) ^ swz(rl);
        s.off[it] = (uint32_t)(wrow * (int)ld + s.c[it] * 8) * 2u;        // weights are a few hundred KB: 32-bit byte offsets
    }
    return s;
}
__device__ __forceinline__ void dma_weights_at(const WSrc& s, const bf16_t* __restrict__ w, const U128* __restrict__ zeros, int k0, int kend,
                                               char* tile, int wave) {
    const char* wk = reinterpret_cast<const char*>(w) + (int64_t)k0 * 2;
#pragma unroll
    for (int it = 0; it < WROWS / 32; ++it) {
        const bool ok = k0 + s.c[it] * 8 + 8 <= kend;
        const void* src = ok ? static_cast<const void*>(wk + s.off[it]) : static_cast<const void*>(zeros);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(tile + (it * 4 + wave) * 1024), 16, 0, 0);
    }
}

// one 64-deep k tile: weight group g (0 r, 1 z, 2 n) accumulates into accumulator group (g < 2 ? g : G2)
template <int NG, int G2>
__device__ __forceinline__ void mma_tile(const char* As, const char* Bs, int wrow0, int fr, int fq, f32x4_t (&acc)[2][NG * 4]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t fa[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) fa[mt] = *reinterpret_cast<const bf16x8_t*>(As + lds_off(wrow0 + mt * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int ag = g < 2 ? g : G2;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(Bs + lds_off(g * 64 + nt * 16 + fr, ks * 4 + fq));
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)   // swapped operands: result rows <-> channels, columns <-> activation rows
                    acc[mt][ag * 4 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa[mt], acc[mt][ag * 4 + nt], 0, 0, 0);
            }
        }
    }
}

__device__ __forceinline__ void ld16(const bf16_t* p, float (&v)[16]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 8);
    v[0] = lo_bf(a.x); v[1] = hi_bf(a.x); v[2] = lo_bf(a.y); v[3] = hi_bf(a.y);
    v[4] = lo_bf(a.z); v[5] = hi_bf(a.z); v[6] = lo_bf(a.w); v[7] = hi_bf(a.w);
    v[8] = lo_bf(b.x); v[9] = hi_bf(b.x); v[10] = lo_bf(b.y); v[11] = hi_bf(b.y);
    v[12] = lo_bf(b.z); v[13] = hi_bf(b.z); v[14] = lo_bf(b.w); v[15] = hi_bf(b.w);
}
__device__ __forceinline__ void st16(bf16_t* p, const float (&v)[16]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf(v[0], v[1]), pack_bf(v[2], v[3]), pack_bf(v[4], v[5]), pack_bf(v[6], v[7]));
    *reinterpret_cast<uint4*>(p + 8) = make_uint4(pack_bf(v[8], v[9]), pack_bf(v[10], v[11]), pack_bf(v[12], v[13]), pack_bf(v[14], v[15]));
}
__device__ __forceinline__ void ldf16(const float* p, float (&v)[16]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 a = *reinterpret_cast<const float4*>(p + i * 4);
        v[i * 4] = a.x; v[i * 4 + 1] = a.y; v[i * 4 + 2] = a.z; v[i * 4 + 3] = a.w;
    }
}

// tanh for the bf16 step kernels: 1 - 2 / (exp(2x) + 1) on v_exp_f32 / v_rcp_f32 (~6 instructions; exact limits -1 / +1, no NaN for finite
// x).  tanhf() is ~40 instructions with range checks per element and made the cell of the fused forward step VALU-bound: ~2,700 VALU
// instructions per wave after the last MFMA, half of them tanhf -- the "cell alone" time of the measuring switches, 437-524 us per 434 k-row
// launch, was arithmetic, not stores.  Absolute error ~1e-7, far below the bf16 rounding of the value it feeds (n is stored as bf16).
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __expf(2.f * x);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

// The cell of a 128-row x 64-channel tile on the accumulators of the gate products (shared by the step kernels below): lane (fr, fq)
// holds rows m0 + wave*32 + mt*16 + fr, channels cb .. cb+15 (index nt*4 + e) of accumulator groups r, z, (n_x,) n_h.
template <int MODE>
__device__ __forceinline__ void step_cell(const StepArgs& a, f32x4_t (&acc)[2][(MODE == 1 ? 4 : 3) * 4], int m0, int c0, int wave, int fr, int fq) {
    constexpr bool HAS_X = MODE == 1;
    constexpr int GH = HAS_X ? 3 : 2;
    // ---- cell: lane (fr, fq) holds rows m0 + wave*32 + mt*16 + fr, channels cb .. cb+15 (index nt*4 + e)
    const int cb = c0 + fq * 16, hs = a.hs;
    float bhr[16], bhz[16], bhn[16];
    ldf16(a.b_hh + cb, bhr); ldf16(a.b_hh + hs + cb, bhz); ldf16(a.b_hh + 2 * hs + cb, bhn);
    if constexpr (MODE != 0) {                   // b_ih of r and z joins b_hh; the n part is added to the input-side term below
        float t[16];
        ldf16(a.b_ih + cb, t);
#pragma unroll
        for (int i = 0; i < 16; ++i) bhr[i] += t[i];
        ldf16(a.b_ih + hs + cb, t);
#pragma unroll
        for (int i = 0; i < 16; ++i) bhz[i] += t[i];
    }
    const float ks = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + wave * 32 + mt * 16 + fr;
        if (m >= a.rows) continue;
        float xr[16], xz[16], xn[16], hp[16];
        if constexpr (HAS_X) {
            ldf16(a.b_ih + 2 * hs + cb, xn);
#pragma unroll
            for (int i = 0; i < 16; ++i) { xr[i] = 0.f; xz[i] = 0.f; xn[i] += acc[mt][8 + (i >> 2)][i & 3]; }
        } else if constexpr (MODE == 2) {
            const bf16_t* fp = a.gf + (int64_t)a.gf_idx[m] * 3 * hs + cb;
            const bf16_t* bp = a.gb + (int64_t)a.gb_idx[m] * 3 * hs + cb;
            float t[16];
            ld16(fp, xr); ld16(bp, t);
#pragma unroll
            for (int i = 0; i < 16; ++i) xr[i] += t[i];
            ld16(fp + hs, xz); ld16(bp + hs, t);
#pragma unroll
            for (int i = 0; i < 16; ++i) xz[i] += t[i];
            ld16(fp + 2 * hs, xn); ld16(bp + 2 * hs, t);
#pragma unroll
            for (int i = 0; i < 16; ++i) xn[i] += t[i];
            ldf16(a.b_ih + 2 * hs + cb, t);
#pragma unroll
            for (int i = 0; i < 16; ++i) xn[i] += t[i];
        } else {
            const bf16_t* xp = a.xg + (int64_t)m * 3 * hs + cb;
            ld16(xp, xr); ld16(xp + hs, xz); ld16(xp + 2 * hs, xn);
        }
        ld16(a.h_in + (int64_t)(a.h_idx ? a.h_idx[m] : m) * hs + cb, hp);
        float gr[16], gz[16], gn[16], hn[16], o[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            gr[i] = 1.f / (1.f + __expf(-(xr[i] + acc[mt][(i >> 2)][i & 3] + bhr[i])));
            gz[i] = 1.f / (1.f + __expf(-(xz[i] + acc[mt][4 + (i >> 2)][i & 3] + bhz[i])));
            hn[i] = acc[mt][GH * 4 + (i >> 2)][i & 3] + bhn[i];
            // the saved hn is what backward multiplies by: round it first so that forward and backward agree
            hn[i] = bf2f(f2bf(hn[i]));
            gn[i] = fast_tanh(xn[i] + gr[i] * hn[i]);
            o[i] = (1.f - gz[i]) * gn[i] + gz[i] * hp[i];
        }
        bf16_t* gp = a.gates + (int64_t)m * 4 * hs + cb;
        st16(gp, gr); st16(gp + hs, gz); st16(gp + 2 * hs, gn);
        st16(gp + 3 * hs, hn);
        bf16_t* hdst = m < a.n_out ? a.h_out + (int64_t)m * hs + cb
                                   : a.h_fin + (int64_t)(a.fin_idx ? a.fin_idx[m] : m) * a.ld_fin + cb;
        st16(hdst, o);
        if (a.y) {
            if (a.p_drop > 0.f) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    o[i] = drop_keep(a.seed, (uint64_t)(a.drop_base + (int64_t)m * a.ldy + cb + i), a.p_drop) ? o[i] * ks : 0.f;
            }
            st16(a.y + (int64_t)m * a.ldy + cb, o);
        }
    }
}

// MODE 0: input gates read from xg; 1: x W_ih^T computed here (HAS_X); 2: input gates gathered from two bf16 tables
template <int MODE>
__global__ __launch_bounds__(256, 2) void gru_step_fwd_kernel(StepArgs a) {
    if (a.p_drop > 0.f) a.seed = live_seed(a.seed);
    constexpr bool HAS_X = MODE == 1;
    constexpr int NG = HAS_X ? 4 : 3;           // accumulator groups: r, z, (n_x,) n_h
    constexpr int GH = HAS_X ? 3 : 2;           // group of the h-part of n
    __shared__ __attribute__((aligned(16))) char lds[A_BYTES + B_BYTES];
    char* As = lds;
    char* Bs = lds + A_BYTES;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int nC = a.hs / TC;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    const int m0 = ((sq / nC) * 8 + xcd) * TM, c0 = (sq % nC) * TC;
    if (m0 >= a.rows) return;
    const U128* Z = static_cast<const U128*>(a.zeros);

    f32x4_t acc[2][NG * 4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NG * 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if constexpr (HAS_X) {
        const RowSrc xs = row_src(a.x, a.ldx, a.rows, m0, wave, lane);
        const WSrc ws = w_src(a.in_dim, a.hs, c0, wave, lane);
        for (int k0 = 0; k0 < a.in_dim; k0 += BK) {
            dma_rows_at(xs, Z, k0, a.in_dim, As, wave);
            dma_weights_at(ws, a.w_ih, Z, k0, a.in_dim, Bs, wave);
            __syncthreads();
            mma_tile<NG, 2>(As, Bs, wave * 32, fr, fq, acc);
            __syncthreads();
        }
    }
    {
        const RowSrc hsrc = row_src(a.h_in, a.hs, a.rows, m0, wave, lane, a.h_idx);
        const WSrc ws = w_src(a.hs, a.hs, c0, wave, lane);
        for (int k0 = 0; k0 < a.hs; k0 += BK) {
            dma_rows_at(hsrc, Z, k0, a.hs, As, wave);
            dma_weights_at(ws, a.w_hh, Z, k0, a.hs, Bs, wave);
            __syncthreads();
            mma_tile<NG, GH>(As, Bs, wave * 32, fr, fq, acc);
            __syncthreads();
        }
    }

    step_cell<MODE>(a, acc, m0, c0, wave, fr, fq);
}


// ---------------------------------------------------------------------------------------------------------------------
// The fused forward step (input product inside) for production-sized launches: 256 rows x 64 channels on eight waves, the k axis [x | h]
// in 64-k stages of whole 128-byte lines per operand row and DMA instruction, THREE slots of activation rows (3 x 32 KB) and two of weight
// rows (2 x 24 KB) = 144 KB of LDS.  Per stage a wave issues its 3 pieces of W(s+1) and THEN its 4 of A(s+2); loads complete in order, so
// "all but the newest four" (vmcnt(4)) is exactly "W(s) and A(s) have landed, A(s+1) may still fly": while stage s is multiplied, A(s+1),
// W(s+1) and A(s+2) are under way -- 88 KB per CU in flight.  One wait + one barrier per stage; lgkmcnt(0) in front of the barrier: hipcc
// sinks a stage's last MFMAs, and the wait of their fragment reads, below it (GTOS_VMCNT_LDS).
// How it got here (round 5; each form bit-identical to the single-stage gru_step_fwd_kernel<1>, all but this one removed in round 6 -- their
// measurements are in DESIGN.md section 5): one 64-k stage ~895 us per 434 k-row launch of layer 1 -> a three-slot ring of 32-k stages
// 870-914 -> two slots of whole 64-k stages 792-826 -> this 730 (layer 0: 524).  The k loop is bound by the round trip of what a CU keeps
// in flight (its DMA alone: 434 us with one 56 KB stage in flight, 332 here), the cell by its stores, and the two add up.
// Slots are static arrays (hipcc's wait-count pass tracks LDS-DMA per LDS object): the body is unrolled over lcm(2, 3) = 6 stages, so the
// stage count (in_dim + hs) / 64 must be a multiple of 6 -- true for both layers of every BASELINE config (2 + 4, 8 + 4); the host sends
// everything else, and launches under 8192 rows, to gru_step_fwd_kernel<1>.  Same lane -> channel map, k order and cell: the same bits
// (tests/test_hip_parity.py::test_gru_forward_pipelined_kernel_bit_identical_to_single_stage).
__global__ __launch_bounds__(512, 2) void gru_step_fwd_a2w3_kernel(StepArgs a) {
    constexpr int TMW = 256, AW = TMW * ROWB;                                  // 32 KB of activation rows, B_BYTES = 24 KB of weight rows
    if (a.p_drop > 0.f) a.seed = live_seed(a.seed);
    __shared__ __attribute__((aligned(16))) char sa0[AW];
    __shared__ __attribute__((aligned(16))) char sa1[AW];
    __shared__ __attribute__((aligned(16))) char sw0[B_BYTES];
    __shared__ __attribute__((aligned(16))) char sw1[B_BYTES];
    __shared__ __attribute__((aligned(16))) char sa2[AW];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int hs = a.hs, nC = hs / TC;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    const int m0 = ((sq / nC) * 8 + xcd) * TMW, c0 = (sq % nC) * TC;
    if (m0 >= a.rows) return;
    const int nkx = a.in_dim / BK, nk = nkx + hs / BK;
    uint32_t axo[4], aho[4], bxo[3], bho[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rl = (i * 8 + wave) * 8 + (lane >> 3);
        const int r = min(m0 + rl, a.rows - 1) - m0;                          // rows past the end re-read the last valid row (never stored)
        const uint32_t c = (uint32_t)(((lane & 7) ^ swz(rl)) << 4);
        axo[i] = (uint32_t)r * (uint32_t)(a.ldx * 2) + c;
        aho[i] = (uint32_t)r * (uint32_t)(hs * 2) + c;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int rl = (i * 8 + wave) * 8 + (lane >> 3);
        const int g = rl >> 6, nt = (rl >> 4) & 3, q = (rl >> 2) & 3, e = rl & 3;
        const int wrow = g * hs + c0 + q * 16 + nt * 4 + e;
        const uint32_t c = (uint32_t)(((lane & 7) ^ swz(rl)) << 4);
        bxo[i] = (uint32_t)wrow * (uint32_t)(a.in_dim * 2) + c;
        bho[i] = (uint32_t)wrow * (uint32_t)(hs * 2) + c;
    }
    const char* Xb = reinterpret_cast<const char*>(a.x + (int64_t)m0 * a.ldx);
    const char* Hb = reinterpret_cast<const char*>(a.h_in + (int64_t)m0 * hs);
    const char* Wi = reinterpret_cast<const char*>(a.w_ih);
    const char* Wh = reinterpret_cast<const char*>(a.w_hh);

    f32x4_t acc[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#define GTOS_DMA1(src, dst)                                                                                                   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                                    \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
// this wave's 4 pieces of the activation rows / 3 pieces of the weight rows of stage s_ (past the end: the last stage again, never multiplied)
#define GTOS_A2W3_DMA_A(slot, s_)                                                                                             \
    {                                                                                                                         \
        const int st_ = min((s_), nk - 1);                                                                                    \
        const bool px_ = st_ < nkx;                                                                                           \
        const char* ab_ = px_ ? Xb + st_ * ROWB : Hb + (st_ - nkx) * ROWB;                                                    \
        GTOS_DMA1(ab_ + (px_ ? axo[0] : aho[0]), (slot) + (0 * 8 + wave) * 1024);                                             \
        GTOS_DMA1(ab_ + (px_ ? axo[1] : aho[1]), (slot) + (1 * 8 + wave) * 1024);                                             \
        GTOS_DMA1(ab_ + (px_ ? axo[2] : aho[2]), (slot) + (2 * 8 + wave) * 1024);                                             \
        GTOS_DMA1(ab_ + (px_ ? axo[3] : aho[3]), (slot) + (3 * 8 + wave) * 1024);                                             \
    }
#define GTOS_A2W3_DMA_W(slot, s_)                                                                                             \
    {                                                                                                                         \
        const int st_ = min((s_), nk - 1);                                                                                    \
        const bool px_ = st_ < nkx;                                                                                           \
        const char* bb_ = px_ ? Wi + st_ * ROWB : Wh + (st_ - nkx) * ROWB;                                                    \
        GTOS_DMA1(bb_ + (px_ ? bxo[0] : bho[0]), (slot) + (0 * 8 + wave) * 1024);                                             \
        GTOS_DMA1(bb_ + (px_ ? bxo[1] : bho[1]), (slot) + (1 * 8 + wave) * 1024);                                             \
        GTOS_DMA1(bb_ + (px_ ? bxo[2] : bho[2]), (slot) + (2 * 8 + wave) * 1024);                                             \
    }
// stage s_: A in sa_, W in sw_; W(s_+1) goes to sw_n (held W(s_-1)), A(s_+2) to sa_n (held A(s_-1))
#define GTOS_A2W3_STEP(sa_, sw_, sa_n, sw_n, s_)                                                                              \
    {                                                                                                                         \
        GTOS_VMCNT_LDS(4);                                 /* own pieces of A(s_), W(s_) landed (A(s_+1) may still fly); own reads of s_ - 1 done */ \
        __builtin_amdgcn_s_barrier();                      /* everybody's; and everybody is past its reads of stage s_ - 1 */ \
        GTOS_A2W3_DMA_W(sw_n, (s_) + 1);                                                                                      \
        GTOS_A2W3_DMA_A(sa_n, (s_) + 2);                                                                                      \
        {                                                                                                                     \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                \
                bf16x8_t fa[2], fb[12];                                                                                       \
                _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                              \
                    fa[mt] = *reinterpret_cast<const bf16x8_t*>((sa_) + lds_off(wave * 32 + mt * 16 + fr, ks * 4 + fq));      \
                _Pragma("unroll") for (int t = 0; t < 12; ++t)                                                                \
                    fb[t] = *reinterpret_cast<const bf16x8_t*>((sw_) + lds_off(t * 16 + fr, ks * 4 + fq));                    \
                _Pragma("unroll") for (int g = 0; g < 2; ++g)                                                                 \
                    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                          \
                        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                      \
                            acc[mt][g * 4 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[g * 4 + nt], fa[mt], acc[mt][g * 4 + nt], 0, 0, 0); \
                if ((s_) < nkx) {                                                                                             \
                    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                          \
                        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                      \
                            acc[mt][8 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[8 + nt], fa[mt], acc[mt][8 + nt], 0, 0, 0); \
                } else {                                                                                                      \
                    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                          \
                        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                      \
                            acc[mt][12 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[8 + nt], fa[mt], acc[mt][12 + nt], 0, 0, 0); \
                }                                                                                                             \
            }                                                                                                                 \
        }                                                                                                                     \
    }

    if (nk > 0) {                                          // the loop's invariant from the start: the newest pieces are one whole stage of one operand
        GTOS_A2W3_DMA_A(sa0, 0);
        GTOS_A2W3_DMA_W(sw0, 0);
        GTOS_A2W3_DMA_A(sa1, 1);
    }
    for (int s = 0; s + 6 <= nk; s += 6) {                 // (nk % 6 == 0: checked by the host)
        // A(s) in slot s % 3, W(s) in slot s % 2; A(s+2) -> (s+2) % 3, W(s+1) -> (s+1) % 2
        GTOS_A2W3_STEP(sa0, sw0, sa2, sw1, s);
        GTOS_A2W3_STEP(sa1, sw1, sa0, sw0, s + 1);
        GTOS_A2W3_STEP(sa2, sw0, sa1, sw1, s + 2);
        GTOS_A2W3_STEP(sa0, sw1, sa2, sw0, s + 3);
        GTOS_A2W3_STEP(sa1, sw0, sa0, sw1, s + 4);
        GTOS_A2W3_STEP(sa2, sw1, sa1, sw0, s + 5);
    }
    GTOS_VMCNT(0);                                         // the dummy prefetches of the last stages
#undef GTOS_A2W3_STEP
#undef GTOS_A2W3_DMA_W
#undef GTOS_A2W3_DMA_A
#undef GTOS_DMA1
    step_cell<1>(a, acc, m0, c0, wave, fr, fq);
}


// ---------------------------------------------------------------------------------------------------------------------
// PERSISTENT forward step for the second GRU layer on the path tries (input gates gathered from the two per-node tables,
// MODE 2 above), hs <= 256.  The step is memory-latency bound in the tile-per-workgroup kernel: every workgroup re-streams
// its 96 KB W_hh slice and walks load -> barrier -> MFMA four times before it even starts the gathers of its epilogue,
// with two workgroups per CU to hide all of that.  Here ONE 4-wave workgroup per CU keeps its channel tile's W_hh slice
// (3 x 64 rows x hs, 96 KB at hs = 256) in LDS for the whole launch and walks the row tiles; the waves are decoupled (each
// owns 32 rows and its own 16 KB state-row buffer, no workgroup barrier in the loop) and software-pipelined: while a wave
// runs the cell of tile i, the LDS-DMA of its state rows for tile i+1 is in flight, and the table / state gathers of
// tile i+1 are issued as soon as the registers of tile i are consumed, so they fly during the next MFMA phase's drain.
// Workgroup -> (XCD, channel tile, worker): the hs/64 channel tiles of a worker sit on ONE XCD and walk the same row
// tiles, so the state rows and node ids are fetched from HBM once and shared through that XCD's L2.
struct Raw16 { uint4 a, b; };
__device__ __forceinline__ void unpack16(const Raw16& r, float (&v)[16]) {
    v[0] = lo_bf(r.a.x); v[1] = hi_bf(r.a.x); v[2] = lo_bf(r.a.y); v[3] = hi_bf(r.a.y);
    v[4] = lo_bf(r.a.z); v[5] = hi_bf(r.a.z); v[6] = lo_bf(r.a.w); v[7] = hi_bf(r.a.w);
    v[8] = lo_bf(r.b.x); v[9] = hi_bf(r.b.x); v[10] = lo_bf(r.b.y); v[11] = hi_bf(r.b.y);
    v[12] = lo_bf(r.b.z); v[13] = hi_bf(r.b.z); v[14] = lo_bf(r.b.w); v[15] = hi_bf(r.b.w);
}
__device__ __forceinline__ Raw16 ldraw16(const bf16_t* p) {
    Raw16 r;
    r.a = *reinterpret_cast<const uint4*>(p);
    r.b = *reinterpret_cast<const uint4*>(p + 8);
    return r;
}

constexpr int P_ROWS = 32;                       // rows per wave tile (2 MFMA row blocks)
constexpr int P_TM = 4 * P_ROWS;                 // rows per workgroup tile

template <int KT>                                // k tiles of 64: hs = 64 * KT
__global__ __launch_bounds__(256, 1) void gru_l1_fwd_persistent_kernel(StepArgs a, int n_workers) {
    extern __shared__ __attribute__((aligned(16))) char plds[];
    constexpr int HS = 64 * KT;
    constexpr int A_TILE = P_ROWS * ROWB;        // one k tile of a wave's 32 rows: 4 KB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    char* Ws = plds;                                               // KT x [192 x 128 B]
    char* Aw = plds + KT * B_BYTES + wave * (KT * A_TILE);         // this wave's state rows: KT x [32 x 128 B]
    const int nC = KT;
    const int xcd = blockIdx.x & 7, sub = blockIdx.x >> 3;
    const int ct = sub % nC, worker = (sub / nC) * 8 + xcd, c0 = ct * TC;
    const U128* Z = static_cast<const U128*>(a.zeros);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) dma_weights(a.w_hh, Z, HS, HS, c0, kt * BK, HS, Ws + kt * B_BYTES, wave, lane);
    __syncthreads();

    const int cb = c0 + fq * 16;
    float bhr[16], bhz[16], bhn[16], bin[16];
    {
        float t[16];
        ldf16(a.b_hh + cb, bhr); ldf16(a.b_ih + cb, t);
#pragma unroll
        for (int i = 0; i < 16; ++i) bhr[i] += t[i];
        ldf16(a.b_hh + HS + cb, bhz); ldf16(a.b_ih + HS + cb, t);
#pragma unroll
        for (int i = 0; i < 16; ++i) bhz[i] += t[i];
        ldf16(a.b_hh + 2 * HS + cb, bhn); ldf16(a.b_ih + 2 * HS + cb, bin);
    }
    const int n_tiles = (a.rows + P_TM - 1) / P_TM;

    auto dma_A = [&](int tile) {
        const int r0 = tile * P_TM + wave * P_ROWS;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int it = 0; it < P_ROWS / 8; ++it) {
                const int rl = it * 8 + (lane >> 3);
                const int c = (lane & 7) ^ swz(rl);
                const int r = r0 + rl;
                const void* src = r < a.rows ? static_cast<const void*>(a.h_in + (int64_t)r * HS + kt * BK + c * 8)
                                             : static_cast<const void*>(Z);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(Aw + kt * A_TILE + it * 1024), 16, 0, 0);
            }
    };
    struct Pre { Raw16 f[3], b[3], h; };
    auto gather = [&](int tile, Pre (&P)[2]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            int m = tile * P_TM + wave * P_ROWS + mt * 16 + fr;
            m = m < a.rows ? m : a.rows - 1;
            const bf16_t* fp = a.gf + (int64_t)a.gf_idx[m] * 3 * HS + cb;
            const bf16_t* bp = a.gb + (int64_t)a.gb_idx[m] * 3 * HS + cb;
#pragma unroll
            for (int g = 0; g < 3; ++g) { P[mt].f[g] = ldraw16(fp + g * HS); P[mt].b[g] = ldraw16(bp + g * HS); }
            P[mt].h = ldraw16(a.h_in + (int64_t)m * HS + cb);
        }
    };

    int tile = worker;
    Pre P[2];
    if (tile < n_tiles) { dma_A(tile); gather(tile, P); }
    for (; tile < n_tiles; tile += n_workers) {
        f32x4_t acc[2][12];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 12; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const char* As = Aw + kt * A_TILE;
            const char* Bs = Ws + kt * B_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t fa[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) fa[mt] = *reinterpret_cast<const bf16x8_t*>(As + lds_off(mt * 16 + fr, ks * 4 + fq));
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(Bs + lds_off(g * 64 + nt * 16 + fr, ks * 4 + fq));
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
                            acc[mt][g * 4 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa[mt], acc[mt][g * 4 + nt], 0, 0, 0);
                    }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);        // every LDS read of this tile's state rows has returned
        const int next = tile + n_workers;
        if (next < n_tiles) dma_A(next);           // flies during the cell below
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = tile * P_TM + wave * P_ROWS + mt * 16 + fr;
            float xr[16], xz[16], xn[16], hp[16], t[16];
            unpack16(P[mt].f[0], xr); unpack16(P[mt].b[0], t);
#pragma unroll
            for (int i = 0; i < 16; ++i) xr[i] += t[i];
            unpack16(P[mt].f[1], xz); unpack16(P[mt].b[1], t);
#pragma unroll
            for (int i = 0; i < 16; ++i) xz[i] += t[i];
            unpack16(P[mt].f[2], xn); unpack16(P[mt].b[2], t);
#pragma unroll
            for (int i = 0; i < 16; ++i) xn[i] += t[i] + bin[i];
            unpack16(P[mt].h, hp);
            float gr[16], gz[16], gn[16], hn[16], o[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                gr[i] = 1.f / (1.f + __expf(-(xr[i] + acc[mt][(i >> 2)][i & 3] + bhr[i])));
                gz[i] = 1.f / (1.f + __expf(-(xz[i] + acc[mt][4 + (i >> 2)][i & 3] + bhz[i])));
                hn[i] = bf2f(f2bf(acc[mt][8 + (i >> 2)][i & 3] + bhn[i]));   // rounded like the saved value backward uses
                gn[i] = fast_tanh(xn[i] + gr[i] * hn[i]);
                o[i] = (1.f - gz[i]) * gn[i] + gz[i] * hp[i];
            }
            if (m < a.rows) {
                bf16_t* gp = a.gates + (int64_t)m * 4 * HS + cb;
                st16(gp, gr); st16(gp + HS, gz); st16(gp + 2 * HS, gn);
                st16(gp + 3 * HS, hn);
                bf16_t* hdst = m < a.n_out ? a.h_out + (int64_t)m * HS + cb
                                           : a.h_fin + (int64_t)(a.fin_idx ? a.fin_idx[m] : m) * a.ld_fin + cb;
                st16(hdst, o);
            }
        }
        if (next < n_tiles) gather(next, P);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Fused BACKWARD time step.  With d4 = [d r | d z | d n_x | d n_h] ([rows, 4*hs], ONE buffer: d(xg) = columns 0..3hs,
// d(hg) = columns {0..2hs, 3hs..4hs}) the gradient that reaches the state of step t through time is
//     dh_t = dh_direct (kept in the fp32 dh buffer) + d(hg)_{later step} W_hh
// so the kernel first multiplies the d4 rows of the step processed just before (A operand, K = 3*hs, skipping the n_x
// block) with W_hh (B operand: rows c0..c0+63 of W_hh^T, K-contiguous), adds dh and the gradient arriving through the
// layer output (dy, same dropout counters as forward), and then runs the cell backward in registers for its 128 rows x
// 64 channels: writes d4, leaves dh = dh_total * z, and accumulates the four bias-gradient column sums
// (wave DPP reduction over the 16 rows of a lane group -> LDS -> one fp32 atomic per (gate, channel) and workgroup).
struct StepBwdArgs {
    const bf16_t* d4_prev; int rows_prev; const bf16_t* wh_t;
    const bf16_t* gates; const bf16_t* hprev; const int* hprev_idx; const bf16_t* dy; int64_t ldy;
    void* dh; int dh_bf16; int64_t ld_dh; bf16_t* d4; float* bias_part; int n_partials;
    bf16_t* hp_out;                                // optional [rows,hs]: the (gathered) entering state of every row, written compactly
    const int* sum_idx;                            // optional [rows]: row m takes its operand row from d4_prev[sum_idx[m]] and its incoming
    const void* dh_src;                            //   state gradient from dh_src[sum_idx[m]] (row stride hs) instead of d4_prev[m] / dh[m]
    int zero_row;                                  //   sum_idx value that stands for "all zero" (a leaf): nothing is fetched for it
    float p_drop; uint64_t seed; int64_t drop_base;
    int rows, hs; const void* zeros;
    // Role B (round 5): workgroups of the SAME launch that turn the d4 rows of the step processed just before into that step's INPUT
    // gradient, dinp[rows_prev, n_in] = d4_prev[:, 0:3hs] (d r | d z | d n_x) x W_ih -- the product nn.GRU's autograd runs as one
    // [N,3hs] x [3hs,in] GEMM per direction after BPTT.  They read the rows the recurrent product of role A reads, at the same time and
    // on the same XCD (one fetch from HBM into its L2 instead of two, a step apart), and their MFMA work runs beside the HBM-bound cell
    // workgroups instead of in front of the next layer.  wi_t = W_ih^T [n_in, 3hs] (K-contiguous).  p_in > 0: dinp is masked with the
    // dropout the forward applied to this layer's INPUT (counter in_drop_base + m * n_in + column): the label-embedding rows.
    const bf16_t* wi_t; bf16_t* dinp; int64_t ld_dinp; int n_in; int dinp_acc;     // dinp_acc: dinp += (the other direction wrote it first)
    float p_in; uint64_t seed_in; int64_t in_drop_base;
};

#define GTOS_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ float row16_sum(float v) {   // sum over the 16 lanes of a DPP row (lane & 15)
    v += GTOS_DPP(v, 0xB1); v += GTOS_DPP(v, 0x4E); v += GTOS_DPP(v, 0x141); v += GTOS_DPP(v, 0x140);
    return v;
}

// 64 rows of W_hh^T (channels c0..c0+63, permuted like the forward weight tile) x 64 k
__device__ __forceinline__ void dma_wt(const bf16_t* __restrict__ w, int64_t ld, int c0, int k0, char* tile, int wave, int lane) {
#pragma unroll
    for (int it = 0; it < TC / 32; ++it) {
        const int blk = it * 4 + wave, rl = blk * 8 + (lane >> 3);
        const int nt = (rl >> 4) & 3, q = (rl >> 2) & 3, e = rl & 3;
        const int wrow = c0 + q * 16 + nt * 4 + e;
        const int c = (lane & 7) ^ swz(rl);
        const void* src = static_cast<const void*>(w + (int64_t)wrow * ld + k0 + c * 8);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(tile + blk * 1024), 16, 0, 0);
    }
}

// Role B of the backward step launch (StepBwdArgs.dinp): one 128-row x 128-column tile of the previous step's input gradient,
// dinp[m0.., n0..] = d4_prev[m0.., 0:3hs] x wi_t[n0.., 0:3hs]^T.  Same single-stage k loop as the recurrent product of role A (the two
// 64-row weight blocks in the permuted order of dma_wt, so a lane ends up with 16 consecutive columns of each half).
// The k loop of both roles of the backward step launch with the ACTIVATION operand one stage ahead (round 5, after the forward step's
// measurements: these loops wait for round trips, and the longer one is the row panel's): two slots for the 128-row panel of d4_prev (2 x 16
// KB), ONE for the weight rows (NB x 8 KB) -- 40 / 48 KB per workgroup, so three workgroups per CU stay (the fully double-buffered form
// needed 64 KB, ran two per CU and was slower: call 22).  Per stage: W(s) is issued, then everybody waits for A(s) (issued a stage ago) and
// W(s), A(s+1) goes into the other slot, the stage is multiplied, and a second barrier frees the weight slot.  Same stage and k order as the
// single-stage loop: the same bits.  SKIP: the A operand skips the d n_x block of d4 (the recurrent product).  nsub <= NB weight pieces are
// valid (uniform per workgroup); the pieces past them are neither fetched nor multiplied.
// Measured against single-stage loops on one box (round 5, call 36): 1065 vs 1080 us (layer 1) and 882 vs 874 us (layer 0) per
// 434 k-row launch alone, GRU backward 29.9 vs 30.3 ms and the step 78.6 / 78.8 vs 78.8 / 79.2 ms.
template <int NB, bool SKIP>
__device__ __forceinline__ void kloop_a2(const bf16_t* __restrict__ A, int64_t lda, int rows_total, int m0, const bf16_t* __restrict__ W,
                                         int64_t ldw, int n0, int nsub, int hs, const U128* __restrict__ Z, char* a0, char* a1, char* ws,
                                         f32x4_t (&acc)[2][NB * 4], int wave, int lane) {
    const int fr = lane & 15, fq = lane >> 4, nst = 3 * hs / BK;
#define GTOS_KA_A(slot, s_)                                                                                                   \
    {                                                                                                                         \
        const int kk_ = min((s_), nst - 1) * BK, ak_ = SKIP ? (kk_ < 2 * hs ? kk_ : kk_ + hs) : kk_;                            \
        dma_rows(A, Z, lda, rows_total, m0, ak_, ak_ + BK, (slot), wave, lane);                                               \
    }
#define GTOS_KA_STEP(cur, nxt, s_)                                                                                            \
    {                                                                                                                         \
        _Pragma("unroll") for (int h = 0; h < NB; ++h)                                                                        \
            if (h < nsub) dma_wt(W, ldw, n0 + h * TC, (s_) * BK, ws + h * TC * ROWB, wave, lane);                             \
        GTOS_VMCNT_LDS(0);                                                                                                    \
        __builtin_amdgcn_s_barrier();                      /* A(s_) and W(s_) of every wave have landed */                    \
        GTOS_KA_A(nxt, (s_) + 1);                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                    \
            bf16x8_t fa[2], fb[4];                                                                                            \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                                  \
                fa[mt] = *reinterpret_cast<const bf16x8_t*>((cur) + lds_off(wave * 32 + mt * 16 + fr, ks * 4 + fq));          \
            _Pragma("unroll") for (int h = 0; h < NB; ++h) {                                                                  \
                if (h < nsub) {                                                                                               \
                    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                          \
                        fb[nt] = *reinterpret_cast<const bf16x8_t*>(ws + h * TC * ROWB + lds_off(nt * 16 + fr, ks * 4 + fq)); \
                    _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                          \
                        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                      \
                            acc[mt][h * 4 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][h * 4 + nt], 0, 0, 0); \
                }                                                                                                             \
            }                                                                                                                 \
        }                                                                                                                     \
        GTOS_LGKM0();                                      /* lgkmcnt(0): this wave's fragment reads have COMPLETED, not just been issued */ \
        __builtin_amdgcn_s_barrier();                      /* the weight slot (and this panel slot) are free again */         \
    }
    GTOS_KA_A(a0, 0);
    int s = 0;
    for (; s + 2 <= nst; s += 2) {
        GTOS_KA_STEP(a0, a1, s);
        GTOS_KA_STEP(a1, a0, s + 1);
    }
    if (s < nst) GTOS_KA_STEP(a0, a1, s);
    GTOS_VMCNT(0);                                         // the dummy prefetch of the last stage
    __builtin_amdgcn_s_barrier();
#undef GTOS_KA_STEP
#undef GTOS_KA_A
}

template <int NS>
__device__ __forceinline__ void dinp_tile(const StepBwdArgs& a, int m0, int n0, char* As, char* Bs, char* A1, int wave, int lane) {
    if (m0 >= a.rows_prev) return;
    const int fr = lane & 15, fq = lane >> 4, hs = a.hs;
    const U128* Z = static_cast<const U128*>(a.zeros);
    const int nsub = min(NS, (a.n_in - n0) / TC);                    // n_in % 64 == 0: the last tile may be narrower
    f32x4_t acc[2][NS * 4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NS * 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    kloop_a2<NS, false>(a.d4_prev, 4 * (int64_t)hs, a.rows_prev, m0, a.wi_t, 3 * (int64_t)hs, n0, nsub, hs, Z, As, A1, Bs, acc, wave, lane);
    const float ks_in = a.p_in > 0.f ? 1.f / (1.f - a.p_in) : 1.f;
    const uint64_t seed_in = a.p_in > 0.f ? live_seed(a.seed_in) : 0;
    // direction 1 adds to what direction 0 wrote: all of a lane's pieces in flight before the first is used (round 6; one dependent
    // round trip per piece before: a launch that accumulates took 1,114 us where one that writes took 1,066)
    Raw16 oldv[2][NS];
    if (a.dinp_acc) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = min(m0 + wave * 32 + mt * 16 + fr, a.rows_prev - 1);
#pragma unroll
            for (int h = 0; h < NS; ++h)
                if (h < nsub) oldv[mt][h] = ldraw16(a.dinp + (int64_t)m * a.ld_dinp + n0 + h * TC + fq * 16);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + wave * 32 + mt * 16 + fr;
        if (m >= a.rows_prev) continue;
#pragma unroll
        for (int h = 0; h < NS; ++h) {
            if (h >= nsub) continue;
            const int nb = n0 + h * TC + fq * 16;
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = acc[mt][h * 4 + (i >> 2)][i & 3];
            if (a.p_in > 0.f) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    v[i] = drop_keep(seed_in, (uint64_t)(a.in_drop_base + (int64_t)m * a.n_in + nb + i), a.p_in) ? v[i] * ks_in : 0.f;
            }
            bf16_t* dp = a.dinp + (int64_t)m * a.ld_dinp + nb;
            if (a.dinp_acc) {
                float old[16];
                unpack16(oldv[mt][h], old);
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] += old[i];
            }
            st16(dp, v);
        }
    }
}

// NS: 64-column pieces per input-gradient tile (role B): 2 = 128 columns (32 KB of LDS with the row panel).  4 = 256 columns (the previous
// step's rows fetched half as often) was built and measured in round 5: role B alone 400 vs 409-432 us per 434 k-row launch of layer 1, the
// whole launch 1128 vs 1046-1059 us at two workgroups per CU (202 registers) and 2314 us at three (124 bytes of scratch per lane): fewer
// fetched bytes do not shorten these k loops.  Fully double-buffered k loops (two 32 KB slots per role: 64 KB of LDS = two workgroups per CU):
// 1098-1126 vs 1072-1082 us -- the third workgroup per CU is worth more than the second weight slot; what runs keeps it: kloop_a2 above.
// Round 6 built two more forms, both bit-identical, both slower, both removed again (profiles/r6_gru_bwd_forms.txt):
// 256-row panels on eight waves with BOTH roles fed from one walk over d4_prev (-42 % LDS-DMA bytes, 88 KB in flight; 1,098-1,107 vs
// 1,062-1,070 us) and persistent workgroups with a 96-column weight slice resident in LDS and decoupled waves (1,224-1,319 us).
// Measured split of this kernel at 434,624 rows of layer 1: 520 us without its k loops + 566 us of k loops alone = 1,070 together.
template <int NS = 2>
__global__ __launch_bounds__(256, 3) void gru_step_bwd_kernel(StepBwdArgs a) {
    if (a.p_drop > 0.f) a.seed = live_seed(a.seed);
    // separate LDS objects for the row-panel slots and the weight slot: hipcc's wait-count pass tracks LDS-DMA per object (with the weight rows
    // behind the panel in one array it drained vmcnt(0) in front of every fragment read of the weights while the next panel was in flight)
    __shared__ __attribute__((aligned(16))) char lds[A_BYTES];
    __shared__ __attribute__((aligned(16))) char lds_w[NS * TC * ROWB];
    __shared__ __attribute__((aligned(16))) char lds_a1[A_BYTES];                   // the second slot of the row panel
    __shared__ float btab[4 * TC];
    char* As = lds;
    char* Bs = lds_w;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int hs = a.hs, nC = hs / TC;
    const int nB = a.dinp ? (a.n_in + NS * TC - 1) / (NS * TC) : 0, per = nC + nB;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    // the workgroups of a 128-row panel -- nC cell tiles (role A), then nB input-gradient tiles (role B) -- run back to back on one XCD
    const int m0 = ((sq / per) * 8 + xcd) * TM, role = sq % per;
    if (role >= nC) { dinp_tile<NS>(a, m0, (role - nC) * NS * TC, As, Bs, lds_a1, wave, lane); return; }
    const int c0 = role * TC;
    if (m0 >= a.rows) return;
    const U128* Z = static_cast<const U128*>(a.zeros);
    if (a.bias_part) { btab[threadIdx.x] = 0.f; }

    f32x4_t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (a.d4_prev && a.sum_idx) {
        // trie: operand rows through the children-sum indirection; the per-lane source addresses are computed once
        const RowSrc src = row_src(a.d4_prev, 4 * (int64_t)hs, a.rows, m0, wave, lane, a.sum_idx, a.zero_row);
        for (int kk = 0; kk < 3 * hs; kk += BK) {
            const int ak = kk < 2 * hs ? kk : kk + hs;                 // skip the d n_x block of d4
            dma_rows_at_z(src, ak, As, wave, lane);
            dma_wt(a.wh_t, 3 * (int64_t)hs, c0, kk, Bs, wave, lane);
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t fa[2], fb[4];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) fa[mt] = *reinterpret_cast<const bf16x8_t*>(As + lds_off(wave * 32 + mt * 16 + fr, ks * 4 + fq));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) fb[nt] = *reinterpret_cast<const bf16x8_t*>(Bs + lds_off(nt * 16 + fr, ks * 4 + fq));
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
            }
            __syncthreads();
        }
    } else if (a.d4_prev && m0 < a.rows_prev) {            // the packed rows: the row panel one stage ahead
        kloop_a2<1, true>(a.d4_prev, 4 * (int64_t)hs, a.rows_prev, m0, a.wh_t, 3 * (int64_t)hs, c0, 1, hs, Z, As, lds_a1, Bs, acc, wave, lane);
    } else if (a.bias_part) {
        __syncthreads();                                               // btab zeroed before anyone adds to it
    }

    const int cb = c0 + fq * 16;
    const float ks = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        // rows past the end read the last valid row and contribute zeros: every lane stays active, because the bias sums
        // below are cross-lane (DPP) reductions
        const int m_raw = m0 + wave * 32 + mt * 16 + fr;
        const bool valid = m_raw < a.rows;
        const int m = valid ? m_raw : a.rows - 1;
        float gr[16], gz[16], gn[16], hn[16], hp[16], g[16];
        const bf16_t* gp = a.gates + (int64_t)m * 4 * hs + cb;
        ld16(gp, gr); ld16(gp + hs, gz); ld16(gp + 2 * hs, gn);
        ld16(gp + 3 * hs, hn);
        ld16(a.hprev + (int64_t)(a.hprev_idx ? a.hprev_idx[m] : m) * hs + cb, hp);
        float* dhp = static_cast<float*>(a.dh) + (int64_t)m * a.ld_dh + cb;
        bf16_t* dhb = static_cast<bf16_t*>(a.dh) + (int64_t)m * a.ld_dh + cb;
        if (a.sum_idx) {                                               // trie: the children's summed gradient lives in another row
            const int64_t sr = a.sum_idx[m];
            if (sr == a.zero_row) {
#pragma unroll
                for (int i = 0; i < 16; ++i) g[i] = 0.f;
            } else if (a.dh_bf16) ld16(static_cast<const bf16_t*>(a.dh_src) + sr * hs + cb, g);
            else ldf16(static_cast<const float*>(a.dh_src) + sr * hs + cb, g);
        } else if (a.dh_bf16) ld16(dhb, g); else ldf16(dhp, g);
#pragma unroll
        for (int i = 0; i < 16; ++i) g[i] += acc[mt][i >> 2][i & 3];
        if (a.dy) {
            float dyv[16];
            ld16(a.dy + (int64_t)m * a.ldy + cb, dyv);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float t2 = dyv[i];
                if (a.p_drop > 0.f) t2 = drop_keep(a.seed, (uint64_t)(a.drop_base + (int64_t)m * a.ldy + cb + i), a.p_drop) ? t2 * ks : 0.f;
                g[i] += t2;
            }
        }
        float dr_[16], dz_[16], dn_[16], dhn[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float dn = g[i] * (1.f - gz[i]);
            const float dz = g[i] * (hp[i] - gn[i]);
            dn_[i] = dn * (1.f - gn[i] * gn[i]);
            dhn[i] = dn_[i] * gr[i];
            dr_[i] = dn_[i] * hn[i] * gr[i] * (1.f - gr[i]);
            dz_[i] = dz * gz[i] * (1.f - gz[i]);
            g[i] *= gz[i];                                             // the direct path h_prev -> h
            if (!valid) { dn_[i] = 0.f; dhn[i] = 0.f; dr_[i] = 0.f; dz_[i] = 0.f; }
        }
        if (valid) {
            if (a.dh_bf16) {
                st16(dhb, g);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<float4*>(dhp + i * 4) = make_float4(g[i * 4], g[i * 4 + 1], g[i * 4 + 2], g[i * 4 + 3]);
            }
            bf16_t* dp = a.d4 + (int64_t)m * 4 * hs + cb;
            st16(dp, dr_); st16(dp + hs, dz_); st16(dp + 2 * hs, dn_); st16(dp + 3 * hs, dhn);
            if (a.hp_out) st16(a.hp_out + (int64_t)m * hs + cb, hp);     // operand rows of the recurrent weight gradient
        }
        if (a.bias_part) {
            // bias gradients: column sums of the values as stored (rounded), reduced over the 16 rows of this lane group
            // right away (keeping 64 running sums per lane across both row blocks costs a third workgroup per CU)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float t0 = row16_sum(bf2f(f2bf(dr_[i]))), t1 = row16_sum(bf2f(f2bf(dz_[i])));
                const float t2 = row16_sum(bf2f(f2bf(dn_[i]))), t3 = row16_sum(bf2f(f2bf(dhn[i])));
                if (fr == 0) {
                    atomicAdd(&btab[0 * TC + fq * 16 + i], t0); atomicAdd(&btab[1 * TC + fq * 16 + i], t1);
                    atomicAdd(&btab[2 * TC + fq * 16 + i], t2); atomicAdd(&btab[3 * TC + fq * 16 + i], t3);
                }
            }
        }
    }
    if (a.bias_part) {
        __syncthreads();
        const int q = threadIdx.x >> 6, ch = threadIdx.x & 63;
        float* dst = a.bias_part + (int64_t)(blockIdx.x % a.n_partials) * 4 * hs + q * hs + c0 + ch;
        atomicAdd(dst, btab[threadIdx.x]);
    }
}


}  // namespace

extern "C" int gtos_gru_step_fwd(int rows, int hs, const void* x, int64_t ldx, int in_dim, const void* w_ih, const float* b_ih,
                                 const void* xg, const void* gf, const int* gf_idx, const void* gb, const int* gb_idx,
                                 const void* h_in, const int* h_idx, const void* w_hh, const float* b_hh,
                                 void* h_out, int n_out, void* h_fin, int64_t ld_fin, const int* fin_idx, void* gates, void* y, int64_t ldy,
                                 float p_drop, uint64_t seed, int64_t drop_base, void* stream) {
    if (rows <= 0) return 0;
    if (hs <= 0 || hs % TC) return -22;
    if (!h_in || !w_hh || !b_hh || !gates) return -23;
    if ((n_out > 0 && !h_out) || (n_out < rows && !h_fin)) return -23;
    if (h_fin && (ld_fin < hs || ld_fin % 8)) return -25;
    const int mode = x ? 1 : (gf ? 2 : 0);
    if (mode == 1) {
        if (!w_ih || !b_ih || in_dim <= 0 || in_dim % 8 || ldx % 8 || (uintptr_t)x % 16 || (uintptr_t)w_ih % 16) return -24;
    } else if (mode == 2) {
        if (!gb || !gf_idx || !gb_idx || !b_ih || (uintptr_t)gf % 16 || (uintptr_t)gb % 16) return -24;
    } else if (!xg || (uintptr_t)xg % 16) return -24;
    if ((uintptr_t)h_in % 16 || (uintptr_t)w_hh % 16 || (uintptr_t)gates % 16 || (uintptr_t)h_out % 16 || (uintptr_t)h_fin % 16 ||
        (uintptr_t)b_hh % 16 || (uintptr_t)b_ih % 16) return -25;
    if (y && ((uintptr_t)y % 16 || ldy % 8)) return -25;
    StepArgs a;
    a.x = (const bf16_t*)x; a.ldx = ldx; a.in_dim = in_dim; a.w_ih = (const bf16_t*)w_ih; a.b_ih = b_ih; a.xg = (const bf16_t*)xg;
    a.gf = (const bf16_t*)gf; a.gf_idx = gf_idx; a.gb = (const bf16_t*)gb; a.gb_idx = gb_idx;
    a.h_in = (const bf16_t*)h_in; a.h_idx = h_idx; a.w_hh = (const bf16_t*)w_hh; a.b_hh = b_hh;
    a.h_out = (bf16_t*)h_out; a.n_out = n_out; a.h_fin = (bf16_t*)h_fin; a.gates = (bf16_t*)gates; a.y = (bf16_t*)y; a.ldy = ldy;
    a.ld_fin = ld_fin; a.fin_idx = fin_idx;
    a.p_drop = p_drop; a.seed = seed; a.drop_base = drop_base; a.rows = rows; a.hs = hs;
    a.zeros = gtos_zero_block();
    if (!a.zeros) return -5;
    const long long nM = (rows + TM - 1) / TM, nC = hs / TC;
    const long long nblk = ((nM + 7) / 8) * 8 * nC;
    if (nblk > 0x7fffffffLL) return -6;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == 2 && hs == 256 && !h_idx && !y && rows >= 16384) {                      // trie evaluation, layer 1: W_hh slice resident in LDS
        constexpr int KT = 4;
        const size_t lds_bytes = (size_t)KT * B_BYTES + 4 * KT * P_ROWS * ROWB;        // 96 KB + 64 KB
        static bool configured = false;
        if (!configured) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(gru_l1_fwd_persistent_kernel<KT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return -7;
            configured = true;
        }
        const int n_cu = 256, n_workers = n_cu / KT;                                   // one workgroup per CU
        hipLaunchKernelGGL(gru_l1_fwd_persistent_kernel<KT>, dim3(n_cu), dim3(256), lds_bytes, s, a, n_workers);
        GTOS_CHECK_LAUNCH();
        return 0;
    }
    // GTOS_GRU_FWD_A2W3=0: never the pipelined kernel (the tests compare it with the single-stage one bit for bit)
    static const bool use_a2w3 = !(getenv("GTOS_GRU_FWD_A2W3") && getenv("GTOS_GRU_FWD_A2W3")[0] == '0');
    if (mode == 1 && use_a2w3 && rows >= 8192 && in_dim % 64 == 0 && (in_dim / 64 + hs / 64) % 6 == 0 && !h_idx && ldx < (1 << 20) &&
        (int64_t)3 * hs * (in_dim > hs ? in_dim : hs) * 2 < (1LL << 31)) {
        const long long nM8 = (rows + 255) / 256, nblk8 = ((nM8 + 7) / 8) * 8 * nC;
        hipLaunchKernelGGL(gru_step_fwd_a2w3_kernel, dim3((unsigned)nblk8), dim3(512), 0, s, a);
    }
    else if (mode == 1) hipLaunchKernelGGL(gru_step_fwd_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    else if (mode == 2) hipLaunchKernelGGL(gru_step_fwd_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(gru_step_fwd_kernel<0>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_gru_step_bwd_fused(int rows, int hs, const void* d4_prev, int rows_prev, const void* w_hh_t,
                                       const void* gates, const void* hprev, const int* hprev_idx, const void* dy, int64_t ldy,
                                       void* dh, int dh_dtype, int64_t ld_dh, void* d4, float p_drop, uint64_t seed, int64_t drop_base,
                                       float* bias_partials, int n_partials, void* hprev_out, const int* sum_idx, const void* dh_src,
                                       int zero_row,
                                       const void* w_ih_t, void* dinp, int64_t ld_dinp, int n_in, int dinp_accumulate, float p_in,
                                       uint64_t seed_in, int64_t in_drop_base, void* stream) {
    const bool role_b = dinp != nullptr && d4_prev != nullptr && rows_prev > 0;
    if (rows <= 0 && !role_b) return 0;
    if (hs <= 0 || hs % TC) return -22;
    if (rows > 0) {
        if (!gates || !hprev || !dh || !d4 || (d4_prev && !w_hh_t)) return -23;
        if (sum_idx && (!dh_src || (uintptr_t)dh_src % 16)) return -23;
    }
    if (dinp && (!d4_prev || !w_ih_t || sum_idx)) return -23;                 // role B reads the previous step's rows in place
    if (dinp && (n_in <= 0 || n_in % TC || ld_dinp < n_in || ld_dinp % 8 || (uintptr_t)dinp % 16 || (uintptr_t)w_ih_t % 16)) return -27;
    if (bias_partials && n_partials < 1) return -26;
    if ((uintptr_t)d4_prev % 16 || (uintptr_t)w_hh_t % 16 || (uintptr_t)gates % 16 || (uintptr_t)hprev % 16 || (uintptr_t)dh % 16 ||
        (uintptr_t)d4 % 16 || (dy && ((uintptr_t)dy % 16 || ldy % 8)) || (rows > 0 && (ld_dh < hs || ld_dh % 8)) ||
        (uintptr_t)hprev_out % 16) return -25;
    StepBwdArgs a;
    a.d4_prev = (const bf16_t*)d4_prev; a.rows_prev = d4_prev ? rows_prev : 0; a.wh_t = (const bf16_t*)w_hh_t;
    a.gates = (const bf16_t*)gates; a.hprev = (const bf16_t*)hprev; a.hprev_idx = hprev_idx; a.dy = (const bf16_t*)dy; a.ldy = ldy;
    a.dh = dh; a.dh_bf16 = dh_dtype == GTOS_BF16; a.ld_dh = ld_dh; a.d4 = (bf16_t*)d4; a.bias_part = bias_partials; a.n_partials = n_partials;
    a.hp_out = (bf16_t*)hprev_out; a.sum_idx = sum_idx; a.dh_src = dh_src; a.zero_row = sum_idx ? zero_row : -1;
    a.wi_t = (const bf16_t*)w_ih_t; a.dinp = role_b ? (bf16_t*)dinp : nullptr; a.ld_dinp = ld_dinp; a.n_in = n_in; a.dinp_acc = dinp_accumulate;
    a.p_in = p_in; a.seed_in = seed_in; a.in_drop_base = in_drop_base;
    a.p_drop = p_drop; a.seed = seed; a.drop_base = drop_base; a.rows = rows > 0 ? rows : 0; a.hs = hs;
    a.zeros = gtos_zero_block();
    if (!a.zeros) return -5;
    const long long cover = role_b && rows_prev > a.rows ? rows_prev : a.rows;
    const long long nM = (cover + TM - 1) / TM, per = hs / TC + (role_b ? (n_in + 2 * TC - 1) / (2 * TC) : 0);
    const long long nblk = ((nM + 7) / 8) * 8 * per;
    if (nblk > 0x7fffffffLL) return -6;
    hipLaunchKernelGGL(gru_step_bwd_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_gru_step_bwd(int rows, int hs, const void* d4_prev, int rows_prev, const void* w_hh_t,
                                 const void* gates, const void* hprev, const int* hprev_idx, const void* dy, int64_t ldy, void* dh, int dh_dtype,
                                 int64_t ld_dh, void* d4, float p_drop, uint64_t seed, int64_t drop_base, float* bias_partials,
                                 int n_partials, void* hprev_out, const int* sum_idx, const void* dh_src, int zero_row, void* stream) {
    return gtos_gru_step_bwd_fused(rows, hs, d4_prev, rows_prev, w_hh_t, gates, hprev, hprev_idx, dy, ldy, dh, dh_dtype, ld_dh, d4,
                                   p_drop, seed, drop_base, bias_partials, n_partials, hprev_out, sum_idx, dh_src, zero_row,
                                   nullptr, nullptr, 0, 0, 0, 0.f, 0, 0, stream);
}

GTOS_SEED_EPOCH_SETTER(gru_step)
