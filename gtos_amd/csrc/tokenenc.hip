// Elementwise pieces of TokenEncoder / CNNEncoder / Highway (/root/reference/generator/encoder.py:123-201) for gfx950.
// The GEMMs of these modules already run on gtos_gemm; what the reference does around them with a dozen small ATen kernels
// per call and direction is three fused HBM-streaming kernels here (each with its backward):
//   * highway gate        new_x, gate = layer(x).chunk(2, -1); x = sigmoid(gate) * x + (1 - sigmoid(gate)) * relu(new_x)   (:141-149)
//   * max over time       conv(x).max(time)[0] -> relu                                                                 (:169-172)
//   * token row assembly  dropout(cat([char_repr, token_embed(token)], -1)), zero-padded to a multiple of 8 columns      (:196-199)
// One thread owns 8 consecutive channels (16 bytes of bf16), rows are contiguous, grids are grid-stride.
#include "common.h"

namespace {

inline int grid_for(int64_t work, int block) { int64_t g = (work + block - 1) / block; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

// y [N, 2D] = layer(x): columns [0, D) = new_x, [D, 2D) = gate; x, out [N, D]
template <typename T>
__global__ void highway_fwd_kernel(int64_t n8, int D, const T* __restrict__ y, const T* __restrict__ x, T* __restrict__ out) {
    const int d8 = D / 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / d8; const int c = (int)(i % d8) * 8;
        float nx[8], g[8], xv[8], o[8];
        Vec8<T>::load(y + row * 2 * D + c, nx);
        Vec8<T>::load(y + row * 2 * D + D + c, g);
        Vec8<T>::load(x + row * D + c, xv);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float s = sigmoidf_(g[e]); o[e] = s * xv[e] + (1.f - s) * fmaxf(nx[e], 0.f); }
        Vec8<T>::store(out + row * D + c, o);
    }
}

// dy [N, 2D] (gradient of layer(x)), dx [N, D] (the direct path through gate * x; the path through layer(x) is the GEMM's)
template <typename T>
__global__ void highway_bwd_kernel(int64_t n8, int D, const T* __restrict__ y, const T* __restrict__ x, const T* __restrict__ dout,
                                   T* __restrict__ dy, T* __restrict__ dx) {
    const int d8 = D / 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / d8; const int c = (int)(i % d8) * 8;
        float nx[8], g[8], xv[8], go[8], dnx[8], dg[8], dxv[8];
        Vec8<T>::load(y + row * 2 * D + c, nx);
        Vec8<T>::load(y + row * 2 * D + D + c, g);
        Vec8<T>::load(x + row * D + c, xv);
        Vec8<T>::load(dout + row * D + c, go);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float s = sigmoidf_(g[e]), r = fmaxf(nx[e], 0.f);
            dxv[e] = go[e] * s;
            dnx[e] = nx[e] > 0.f ? go[e] * (1.f - s) : 0.f;
            dg[e] = go[e] * (xv[e] - r) * s * (1.f - s);
        }
        Vec8<T>::store(dy + row * 2 * D + c, dnx);
        Vec8<T>::store(dy + row * 2 * D + D + c, dg);
        Vec8<T>::store(dx + row * D + c, dxv);
    }
}

// y [N, L, F] -> out [N, F] = relu(max_t y[n, t, :]); arg [N, F] = the first maximising t (torch.max's tie rule), as one byte
template <typename T>
__global__ void max_relu_fwd_kernel(int64_t n8, int L, int F, const T* __restrict__ y, T* __restrict__ out, uint8_t* __restrict__ arg) {
    const int f8 = F / 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / f8; const int c = (int)(i % f8) * 8;
        float best[8]; int at[8];
        Vec8<T>::load(y + (n * L) * F + c, best);
#pragma unroll
        for (int e = 0; e < 8; ++e) at[e] = 0;
        for (int t = 1; t < L; ++t) {
            float v[8];
            Vec8<T>::load(y + (n * L + t) * F + c, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) if (v[e] > best[e]) { best[e] = v[e]; at[e] = t; }
        }
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            best[e] = fmaxf(best[e], 0.f);
            if (e < 4) lo |= (uint32_t)at[e] << (8 * e); else hi |= (uint32_t)at[e] << (8 * (e - 4));
        }
        Vec8<T>::store(out + n * F + c, best);
        *reinterpret_cast<uint2*>(arg + n * F + c) = make_uint2(lo, hi);
    }
}

// dy [N, L, F]: dout at the maximising position where the output is positive, zero elsewhere (every element written)
template <typename T>
__global__ void max_relu_bwd_kernel(int64_t n8, int L, int F, const T* __restrict__ out, const uint8_t* __restrict__ arg,
                                    const T* __restrict__ dout, T* __restrict__ dy) {
    const int f8 = F / 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / f8; const int c = (int)(i % f8) * 8;
        float o[8], go[8];
        Vec8<T>::load(out + n * F + c, o);
        Vec8<T>::load(dout + n * F + c, go);
        const uint2 a = *reinterpret_cast<const uint2*>(arg + n * F + c);
        for (int t = 0; t < L; ++t) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int at = (int)(((e < 4 ? a.x : a.y) >> (8 * (e & 3))) & 0xff);
                v[e] = (at == t && o[e] > 0.f) ? go[e] : 0.f;
            }
            Vec8<T>::store(dy + (n * L + t) * F + c, v);
        }
    }
}

// out [N, Cp] = dropout([feat[n, 0:Cc] | table[tok[n], 0:Ct] | 0...]); Cc % 8 == 0, Cp = roundup(Cc + Ct, 8); table fp32
template <typename T>
__global__ void token_row_fwd_kernel(int64_t n8, int Cc, int Ct, int Cp, const T* __restrict__ feat, const int64_t* __restrict__ tok,
                                     const float* __restrict__ table, T* __restrict__ out, float p_drop, uint64_t seed) {
    if (p_drop > 0.f) seed = live_seed(seed);
    const int p8 = Cp / 8;
    const float ks = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / p8; const int c = (int)(i % p8) * 8;
        float v[8];
        if (c < Cc) {
            Vec8<T>::load(feat + n * Cc + c, v);
        } else {
            const float* trow = table + tok[n] * Ct;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (c - Cc + e < Ct) ? trow[c - Cc + e] : 0.f;
        }
        if (p_drop > 0.f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = drop_keep(seed, (uint64_t)n * Cp + c + e, p_drop) ? v[e] * ks : 0.f;
        }
        Vec8<T>::store(out + n * Cp + c, v);
    }
}

// d_feat [N, Cc] = mask * dout[:, 0:Cc]; dtable[tok[n], :] += mask * dout[n, Cc:Cc+Ct] (fp32 atomics: a word table has thousands of rows)
template <typename T>
__global__ void token_row_bwd_kernel(int64_t n8, int Cc, int Ct, int Cp, const T* __restrict__ dout, const int64_t* __restrict__ tok,
                                     T* __restrict__ dfeat, float* __restrict__ dtable, float p_drop, uint64_t seed) {
    if (p_drop > 0.f) seed = live_seed(seed);
    const int p8 = Cp / 8;
    const float ks = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / p8; const int c = (int)(i % p8) * 8;
        float v[8];
        Vec8<T>::load(dout + n * Cp + c, v);
        if (p_drop > 0.f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = drop_keep(seed, (uint64_t)n * Cp + c + e, p_drop) ? v[e] * ks : 0.f;
        }
        if (c < Cc) {
            if (dfeat) Vec8<T>::store(dfeat + n * Cc + c, v);
        } else if (dtable) {
            float* trow = dtable + tok[n] * Ct;
#pragma unroll
            for (int e = 0; e < 8; ++e) if (c - Cc + e < Ct && v[e] != 0.f) atomicAdd(trow + c - Cc + e, v[e]);
        }
    }
}

}  // namespace

#define GTOS_TE_LAUNCH(KERNEL, T, N8, ...)                                                                              \
    do {                                                                                                                \
        hipLaunchKernelGGL(KERNEL<T>, dim3(grid_for((N8), 256)), dim3(256), 0, static_cast<hipStream_t>(stream), (N8), __VA_ARGS__); \
        GTOS_CHECK_LAUNCH();                                                                                            \
    } while (0)

extern "C" int gtos_highway_fwd(int dtype, int64_t N, int D, const void* y, const void* x, void* out, void* stream) {
    if (N <= 0) return 0;
    if (D <= 0 || D % 8 || !y || !x || !out) return -24;
    const int64_t n8 = N * (D / 8);
    if (dtype == GTOS_BF16) GTOS_TE_LAUNCH(highway_fwd_kernel, bf16_t, n8, D, (const bf16_t*)y, (const bf16_t*)x, (bf16_t*)out);
    else GTOS_TE_LAUNCH(highway_fwd_kernel, float, n8, D, (const float*)y, (const float*)x, (float*)out);
    return 0;
}

extern "C" int gtos_highway_bwd(int dtype, int64_t N, int D, const void* y, const void* x, const void* dout, void* dy, void* dx,
                                void* stream) {
    if (N <= 0) return 0;
    if (D <= 0 || D % 8 || !y || !x || !dout || !dy || !dx) return -24;
    const int64_t n8 = N * (D / 8);
    if (dtype == GTOS_BF16)
        GTOS_TE_LAUNCH(highway_bwd_kernel, bf16_t, n8, D, (const bf16_t*)y, (const bf16_t*)x, (const bf16_t*)dout, (bf16_t*)dy, (bf16_t*)dx);
    else GTOS_TE_LAUNCH(highway_bwd_kernel, float, n8, D, (const float*)y, (const float*)x, (const float*)dout, (float*)dy, (float*)dx);
    return 0;
}

extern "C" int gtos_max_relu_fwd(int dtype, int64_t N, int L, int F, const void* y, void* out, uint8_t* arg, void* stream) {
    if (N <= 0) return 0;
    if (L <= 0 || L > 255 || F <= 0 || F % 8 || !y || !out || !arg) return -24;
    const int64_t n8 = N * (F / 8);
    if (dtype == GTOS_BF16) GTOS_TE_LAUNCH(max_relu_fwd_kernel, bf16_t, n8, L, F, (const bf16_t*)y, (bf16_t*)out, arg);
    else GTOS_TE_LAUNCH(max_relu_fwd_kernel, float, n8, L, F, (const float*)y, (float*)out, arg);
    return 0;
}

extern "C" int gtos_max_relu_bwd(int dtype, int64_t N, int L, int F, const void* out, const uint8_t* arg, const void* dout, void* dy,
                                 void* stream) {
    if (N <= 0) return 0;
    if (L <= 0 || L > 255 || F <= 0 || F % 8 || !out || !arg || !dout || !dy) return -24;
    const int64_t n8 = N * (F / 8);
    if (dtype == GTOS_BF16) GTOS_TE_LAUNCH(max_relu_bwd_kernel, bf16_t, n8, L, F, (const bf16_t*)out, arg, (const bf16_t*)dout, (bf16_t*)dy);
    else GTOS_TE_LAUNCH(max_relu_bwd_kernel, float, n8, L, F, (const float*)out, arg, (const float*)dout, (float*)dy);
    return 0;
}

static int token_row_shape_ok(int Cc, int Ct, int Cp) { return !(Cc < 0 || Cc % 8 || Ct <= 0 || Cp % 8 || Cp < Cc + Ct || Cp >= Cc + Ct + 8); }

extern "C" int gtos_token_row_fwd(int dtype, int64_t N, int Cc, int Ct, int Cp, const void* feat, const int64_t* tok, const float* table,
                                  void* out, float p_drop, uint64_t seed, void* stream) {
    if (N <= 0) return 0;
    if (!token_row_shape_ok(Cc, Ct, Cp) || (Cc && !feat) || !tok || !table || !out) return -24;
    const int64_t n8 = N * (Cp / 8);
    if (dtype == GTOS_BF16) GTOS_TE_LAUNCH(token_row_fwd_kernel, bf16_t, n8, Cc, Ct, Cp, (const bf16_t*)feat, tok, table, (bf16_t*)out, p_drop, seed);
    else GTOS_TE_LAUNCH(token_row_fwd_kernel, float, n8, Cc, Ct, Cp, (const float*)feat, tok, table, (float*)out, p_drop, seed);
    return 0;
}

extern "C" int gtos_token_row_bwd(int dtype, int64_t N, int Cc, int Ct, int Cp, const void* dout, const int64_t* tok, void* dfeat,
                                  float* dtable, float p_drop, uint64_t seed, void* stream) {
    if (N <= 0) return 0;
    if (!token_row_shape_ok(Cc, Ct, Cp) || !dout || !tok) return -24;
    const int64_t n8 = N * (Cp / 8);
    if (dtype == GTOS_BF16) GTOS_TE_LAUNCH(token_row_bwd_kernel, bf16_t, n8, Cc, Ct, Cp, (const bf16_t*)dout, tok, (bf16_t*)dfeat, dtable, p_drop, seed);
    else GTOS_TE_LAUNCH(token_row_bwd_kernel, float, n8, Cc, Ct, Cp, (const float*)dout, tok, (float*)dfeat, dtable, p_drop, seed);
    return 0;
}

GTOS_SEED_EPOCH_SETTER(tokenenc)
