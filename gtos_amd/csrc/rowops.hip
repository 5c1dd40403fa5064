// Row-wise / element-wise HBM-bound kernels for gfx950: fused dropout+residual+LayerNorm (fwd/bwd),
// ReLU+dropout backward, bias-gradient column sums, the GRU cell (fwd/bwd), relation gather-mean,
// flat Adam + gradient square-norm, fp32->bf16 weight mirror.  All use 16/32-byte per-lane vector accesses.
#include "common.h"

namespace {

// =========================================================================== LayerNorm(x + dropout(r))
// Replaces  x = F.dropout(x); x = layer_norm(residual + x)
// (/root/reference/generator/graph_transformer.py:57-58,64-65, generator/transformer.py:57-58,63-64,70-71).
// One wave per row; d % 8 == 0, d <= 1024 (each lane holds up to 2 chunks of 8 channels).
constexpr int LN_MAXC = 2;

// Mixed storage types (round 4): TX = residual stream (x, y, dx), TR = sub-layer output (r, dr), so that bf16 activations can
// ride on an fp32 residual stream (x [rows,d] is a few MB per layer: free) -- y2 is an optional bf16 copy of y for the next GEMM
// operand, dy2 its gradient (added to dy in registers).  All arithmetic in fp32 as before.
template <typename TX, typename TR, typename TY>
__global__ __launch_bounds__(256) void ln_fwd_kernel(int rows, int d, const TX* __restrict__ x, const TR* __restrict__ r,
                                                     float p_drop, uint64_t seed, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, TY* __restrict__ y, bf16_t* __restrict__ y2,
                                                     float* __restrict__ mean, float* __restrict__ rstd) {
    if (p_drop > 0.f) seed = live_seed(seed);
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float ks = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    float z[LN_MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int cI = 0; cI < LN_MAXC; ++cI) {
        const int c = (cI * 64 + lane) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[cI][e] = 0.f;
        if (c < d) {
            Vec8<TX>::load(x + (int64_t)row * d + c, z[cI]);
            if (r) {
                float rv[8];
                Vec8<TR>::load(r + (int64_t)row * d + c, rv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = rv[e];
                    if (p_drop > 0.f) t = drop_keep(seed, (uint64_t)row * d + c + e, p_drop) ? t * ks : 0.f;
                    z[cI][e] += t;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += z[cI][e];
        }
    }
    const float mu = wave_sum(s) / d;
    float v = 0.f;
#pragma unroll
    for (int cI = 0; cI < LN_MAXC; ++cI) {
        const int c = (cI * 64 + lane) * 8;
        if (c < d) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float t = z[cI][e] - mu; v = fmaf(t, t, v); }
        }
    }
    const float rs = rsqrtf(wave_sum(v) / d + eps);
#pragma unroll
    for (int cI = 0; cI < LN_MAXC; ++cI) {
        const int c = (cI * 64 + lane) * 8;
        if (c < d) {
            float gm[8], bt[8], o[8];
            Vec8<float>::load(gamma + c, gm);
            Vec8<float>::load(beta + c, bt);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf((z[cI][e] - mu) * rs, gm[e], bt[e]);
            Vec8<TY>::store(y + (int64_t)row * d + c, o);
            if (y2) Vec8<bf16_t>::store(y2 + (int64_t)row * d + c, o);
        }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// backward: dz = rstd * (g - mean(g) - xhat * mean(g*xhat)), g = (dy + dy2)*gamma; dx = dz; dr = dz * dropmask.
// Each wave walks rows grid-stride and keeps dgamma/dbeta partials in registers -> one atomicAdd per lane-channel.
template <typename TX, typename TR, typename TY>
__global__ __launch_bounds__(256) void ln_bwd_kernel(int rows, int d, const TY* __restrict__ dy, const bf16_t* __restrict__ dy2,
                                                     const TX* __restrict__ x,
                                                     const TR* __restrict__ r, float p_drop, uint64_t seed,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, TX* __restrict__ dx, TR* __restrict__ dr,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
    if (p_drop > 0.f) seed = live_seed(seed);
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const float ks = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    float dg[LN_MAXC][8], db[LN_MAXC][8], gm[LN_MAXC][8];
#pragma unroll
    for (int cI = 0; cI < LN_MAXC; ++cI) {
        const int c = (cI * 64 + lane) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { dg[cI][e] = 0.f; db[cI][e] = 0.f; gm[cI][e] = 0.f; }
        if (c < d) Vec8<float>::load(gamma + c, gm[cI]);
    }
    for (int row = wid; row < rows; row += nw) {
        const float mu = mean[row], rs = rstd[row];
        float g[LN_MAXC][8], xh[LN_MAXC][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int cI = 0; cI < LN_MAXC; ++cI) {
            const int c = (cI * 64 + lane) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) { g[cI][e] = 0.f; xh[cI][e] = 0.f; }
            if (c < d) {
                float z[8], dyv[8];
                Vec8<TX>::load(x + (int64_t)row * d + c, z);
                if (dy) Vec8<TY>::load(dy + (int64_t)row * d + c, dyv);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) dyv[e] = 0.f;
                }
                if (dy2) {
                    float t2[8];
                    Vec8<bf16_t>::load(dy2 + (int64_t)row * d + c, t2);
#pragma unroll
                    for (int e = 0; e < 8; ++e) dyv[e] += t2[e];
                }
                if (r) {
                    float rv[8];
                    Vec8<TR>::load(r + (int64_t)row * d + c, rv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float t = rv[e];
                        if (p_drop > 0.f) t = drop_keep(seed, (uint64_t)row * d + c + e, p_drop) ? t * ks : 0.f;
                        z[e] += t;
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[cI][e] = (z[e] - mu) * rs;
                    g[cI][e] = dyv[e] * gm[cI][e];
                    s1 += g[cI][e]; s2 = fmaf(g[cI][e], xh[cI][e], s2);
                    dg[cI][e] = fmaf(dyv[e], xh[cI][e], dg[cI][e]); db[cI][e] += dyv[e];
                }
            }
        }
        s1 = wave_sum(s1) / d; s2 = wave_sum(s2) / d;
#pragma unroll
        for (int cI = 0; cI < LN_MAXC; ++cI) {
            const int c = (cI * 64 + lane) * 8;
            if (c < d) {
                float dz[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) dz[e] = rs * (g[cI][e] - s1 - xh[cI][e] * s2);
                if (dx) Vec8<TX>::store(dx + (int64_t)row * d + c, dz);
                if (dr) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (p_drop > 0.f) dz[e] = drop_keep(seed, (uint64_t)row * d + c + e, p_drop) ? dz[e] * ks : 0.f;
                    Vec8<TR>::store(dr + (int64_t)row * d + c, dz);
                }
            }
        }
    }
    // 4 waves -> 1 through LDS, then one atomicAdd per channel per block
    __shared__ float red[4][2][LN_MAXC * 512];
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int cI = 0; cI < LN_MAXC; ++cI) {
        const int c = (cI * 64 + lane) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[w][0][c + e] = dg[cI][e]; red[w][1][c + e] = db[cI][e]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) {
        atomicAdd(dgamma + c, red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c]);
        atomicAdd(dbeta + c, red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c]);
    }
}

// =========================================================================== relu+dropout backward (in place)
// dh *= (h > 0) * 1/(1-p): h is the saved post-ReLU post-dropout activation, zero wherever either killed it.
template <typename T>
__global__ void relu_drop_bwd_kernel(int64_t n8, T* __restrict__ dh, const T* __restrict__ h, float ks) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float a[8], b[8];
        Vec8<T>::load(dh + i * 8, a);
        Vec8<T>::load(h + i * 8, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = b[e] > 0.f ? a[e] * ks : 0.f;
        Vec8<T>::store(dh + i * 8, a);
    }
}

// =========================================================================== bias gradient: out[n] += sum_rows dy[row, n]
// Block = 4 waves; lane owns 8 consecutive columns of a 512-column chunk (one wave reads 1 KB of a row per load),
// each wave walks its own rows of the block's slice 4 at a time; LDS reduce over the 4 waves, one atomic per column.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(int rows, int N, int64_t ld, const T* __restrict__ dy, float* __restrict__ out,
                                                     int rows_per_block) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 8;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool full = (c + 8 <= N) && (ld % 8 == 0) && (((uintptr_t)dy) % 16 == 0);
    if (c < N) {
        if (full) {
            int r = r0 + w;
            for (; r + 12 < r1; r += 16) {
                float v0[8], v1[8], v2[8], v3[8];
                Vec8<T>::load(dy + (int64_t)r * ld + c, v0);
                Vec8<T>::load(dy + (int64_t)(r + 4) * ld + c, v1);
                Vec8<T>::load(dy + (int64_t)(r + 8) * ld + c, v2);
                Vec8<T>::load(dy + (int64_t)(r + 12) * ld + c, v3);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (v0[e] + v1[e]) + (v2[e] + v3[e]);
            }
            for (; r < r1; r += 4) {
                float v0[8];
                Vec8<T>::load(dy + (int64_t)r * ld + c, v0);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v0[e];
            }
        } else {
            for (int r = r0 + w; r < r1; r += 4)
                for (int e = 0; e < 8 && c + e < N; ++e) acc[e] += to_f<T>(dy[(int64_t)r * ld + c + e]);
        }
    }
    __shared__ float red[4][512];
#pragma unroll
    for (int e = 0; e < 8; ++e) red[w][lane * 8 + e] = acc[e];
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int col = blockIdx.x * 512 + i;
        if (col < N) atomicAdd(out + col, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
    }
}

// =========================================================================== GRU cell (PyTorch gate order r,z,n)
// Restates nn.GRU's cell as used by RelationEncoder (/root/reference/generator/encoder.py:76-82,101-105):
//   r = sig(xg_r + hg_r), z = sig(xg_z + hg_z), n = tanh(xg_n + r*hg_n), h' = (1-z)*n + z*h
// xg = x W_ih^T + b_ih and hg = h W_hh^T + b_hh come from the MFMA GEMM.  Rows are the active prefix of the
// length-sorted sequences.  Writes h' in place, the layer output y (optionally with inter-layer dropout),
// the previous state (for dW_hh) and the gate activations (for backward).
template <typename T>
__global__ void gru_fwd_kernel(int rows, int hs, const T* __restrict__ xg, const T* __restrict__ hg, T* __restrict__ h,
                               T* __restrict__ y, int64_t ldy, T* __restrict__ hprev_save, T* __restrict__ gates,
                               float p_drop, uint64_t seed, int64_t drop_base) {
    if (p_drop > 0.f) seed = live_seed(seed);
    const int hv = hs / 8;
    const float ks = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < (int64_t)rows * hv; t += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(t / hv), c = (int)(t % hv) * 8;
        float xr[8], xz[8], xn[8], hr[8], hz[8], hn[8], hp[8], o[8], gr[8], gz[8], gn[8];
        const T* xp = xg + (int64_t)row * 3 * hs + c;
        const T* hp_ = hg + (int64_t)row * 3 * hs + c;
        Vec8<T>::load(xp, xr); Vec8<T>::load(xp + hs, xz); Vec8<T>::load(xp + 2 * hs, xn);
        Vec8<T>::load(hp_, hr); Vec8<T>::load(hp_ + hs, hz); Vec8<T>::load(hp_ + 2 * hs, hn);
        Vec8<T>::load(h + (int64_t)row * hs + c, hp);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            gr[e] = 1.f / (1.f + __expf(-(xr[e] + hr[e])));
            gz[e] = 1.f / (1.f + __expf(-(xz[e] + hz[e])));
            gn[e] = tanhf(xn[e] + gr[e] * hn[e]);
            o[e] = (1.f - gz[e]) * gn[e] + gz[e] * hp[e];
        }
        Vec8<T>::store(h + (int64_t)row * hs + c, o);
        Vec8<T>::store(hprev_save + (int64_t)row * hs + c, hp);
        T* gp = gates + (int64_t)row * 4 * hs + c;
        Vec8<T>::store(gp, gr); Vec8<T>::store(gp + hs, gz); Vec8<T>::store(gp + 2 * hs, gn); Vec8<T>::store(gp + 3 * hs, hn);
        if (y) {
            if (p_drop > 0.f) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    o[e] = drop_keep(seed, (uint64_t)(drop_base + (int64_t)row * ldy + c + e), p_drop) ? o[e] * ks : 0.f;
            }
            Vec8<T>::store(y + (int64_t)row * ldy + c, o);
        }
    }
}

// backward of one step.  dh (in/out, [rows,hs]): on entry the gradient flowing back through time; the kernel adds
// the gradient arriving through the layer output (dy, masked by the same dropout), writes d(xg) [rows,3hs],
// d(hg) [rows,3hs] and leaves dh = dh_total * z (the direct path); the caller then adds d(hg) W_hh with the GEMM.
template <typename T>
__global__ void gru_bwd_kernel(int rows, int hs, const T* __restrict__ gates, const T* __restrict__ hprev,
                               const T* __restrict__ dy, int64_t ldy, float* __restrict__ dh,
                               T* __restrict__ dxg, T* __restrict__ dhg, float p_drop, uint64_t seed, int64_t drop_base,
                               float* __restrict__ bias_part) {
    if (p_drop > 0.f) seed = live_seed(seed);
    // bias_part (optional, [gridDim.x, 4*hs] fp32, owned block-row-wise): running column sums of d(r), d(z), d(n_x),
    // d(n_h) -- the bias gradients of the GRU -- accumulated here instead of re-reading dxg/dhg in 8 colsum passes.
    // Needs 256 % (hs/8) == 0 so that a thread keeps the same 8 channels over its grid-stride rows.
    extern __shared__ float btab[];
    if (bias_part) { for (int i = threadIdx.x; i < 4 * hs; i += blockDim.x) btab[i] = 0.f; __syncthreads(); }
    float bs[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[q][e] = 0.f;
    int my_c = -1;
    const int hv = hs / 8;
    const float ks = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < (int64_t)rows * hv; t += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(t / hv), c = (int)(t % hv) * 8;
        float gr[8], gz[8], gn[8], hn[8], hp[8], g[8], dyv[8];
        const T* gp = gates + (int64_t)row * 4 * hs + c;
        Vec8<T>::load(gp, gr); Vec8<T>::load(gp + hs, gz); Vec8<T>::load(gp + 2 * hs, gn); Vec8<T>::load(gp + 3 * hs, hn);
        Vec8<T>::load(hprev + (int64_t)row * hs + c, hp);
        Vec8<float>::load(dh + (int64_t)row * hs + c, g);
        if (dy) {
            Vec8<T>::load(dy + (int64_t)row * ldy + c, dyv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t2 = dyv[e];
                if (p_drop > 0.f) t2 = drop_keep(seed, (uint64_t)(drop_base + (int64_t)row * ldy + c + e), p_drop) ? t2 * ks : 0.f;
                g[e] += t2;
            }
        }
        float dr_[8], dz_[8], dn_[8], dhn[8], dhp[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float dn = g[e] * (1.f - gz[e]);
            const float dz = g[e] * (hp[e] - gn[e]);
            dhp[e] = g[e] * gz[e];
            dn_[e] = dn * (1.f - gn[e] * gn[e]);
            dhn[e] = dn_[e] * gr[e];
            dr_[e] = dn_[e] * hn[e] * gr[e] * (1.f - gr[e]);
            dz_[e] = dz * gz[e] * (1.f - gz[e]);
        }
        Vec8<float>::store(dh + (int64_t)row * hs + c, dhp);
        T* xp = dxg + (int64_t)row * 3 * hs + c;
        T* hp2 = dhg + (int64_t)row * 3 * hs + c;
        Vec8<T>::store(xp, dr_); Vec8<T>::store(xp + hs, dz_); Vec8<T>::store(xp + 2 * hs, dn_);
        Vec8<T>::store(hp2, dr_); Vec8<T>::store(hp2 + hs, dz_); Vec8<T>::store(hp2 + 2 * hs, dhn);
        if (bias_part) {
            my_c = c;
#pragma unroll
            for (int e = 0; e < 8; ++e) {   // sums of the values as stored (rounded to T), like a colsum over dxg/dhg would see
                bs[0][e] += to_f<T>(from_f<T>(dr_[e])); bs[1][e] += to_f<T>(from_f<T>(dz_[e]));
                bs[2][e] += to_f<T>(from_f<T>(dn_[e])); bs[3][e] += to_f<T>(from_f<T>(dhn[e]));
            }
        }
    }
    if (bias_part) {
        if (my_c >= 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) atomicAdd(&btab[q * hs + my_c + e], bs[q][e]);
        }
        __syncthreads();
        float* dst = bias_part + (int64_t)blockIdx.x * 4 * hs;
        for (int i = threadIdx.x; i < 4 * hs; i += blockDim.x) dst[i] += btab[i];
    }
}

// =========================================================================== relation gather-mean (eval lookup)
// out[p,:] = sum_k bank[idx[p,k]] (row 0 treated as zero) / max(1, #{k: idx[p,k] != 0})
// (/root/reference/generator/generator.py:83-88).  K == 1 without the row-0 rule is the train lookup (:79).
template <typename T>
__global__ __launch_bounds__(256) void gather_mean_kernel(int64_t P, int K, int d, const T* __restrict__ bank,
                                                          const int64_t* __restrict__ idx, int zero_row0, T* __restrict__ out) {
    const int dv = d / 8;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < P * dv; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = t / dv; const int c = (int)(t % dv) * 8;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int cnt = 0;
        for (int k = 0; k < K; ++k) {
            const int64_t id = idx[p * K + k];
            if (zero_row0 && id == 0) continue;
            ++cnt;
            float v[8];
            Vec8<T>::load(bank + id * d + c, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
        if (zero_row0) {
            const float s = 1.f / (float)max(cnt, 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] *= s;
        }
        Vec8<T>::store(out + p * d + c, acc);
    }
}


// =========================================================================== relation-label embedding rows
// out[n, 0:dim_pad] = dropout(table[tok[n], 0:dim]) zero-padded to dim_pad (a multiple of 8), in the compute dtype:
// nn.Embedding + F.dropout of RelationEncoder (/root/reference/generator/encoder.py:99-100) plus the layout the GEMM wants.
template <typename T>
__global__ void embed_rows_fwd_kernel(int64_t n, int dim, int dim_pad, const int64_t* __restrict__ tok, const float* __restrict__ table,
                                      T* __restrict__ out, float p_drop, uint64_t seed) {
    if (p_drop > 0.f) seed = live_seed(seed);
    const int vpr = dim_pad / 8;
    const float ks = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * vpr; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / vpr; const int c = (int)(t % vpr) * 8;
        const float* src = table + tok[row] * dim;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = (c + e < dim) ? src[c + e] : 0.f;
            if (p_drop > 0.f) x = drop_keep(seed, (uint64_t)row * dim_pad + c + e, p_drop) ? x * ks : 0.f;
            v[e] = x;
        }
        Vec8<T>::store(out + row * dim_pad + c, v);
    }
}

// The RelationEncoder's packed input in ONE pass, with no host read (round 5).  The reference sorts the paths by length, packs them
// time-major and embeds the packed tokens (generator/encoder.py:93-100); here the sorted order and the step sizes come WITH the batch
// (the loader knows the lengths: gtos_amd.pathtrie.PathTrie.seq_order32 / batch_sizes), and packed row n = (step t, sorted path m) with
// offs[t] <= n < offs[t+1], m = n - offs[t] takes token bank[t, order[m]]:
//   x[n, 0:dim_pad]  = dropout(table[token]) zero-padded   (counter n * dim_pad + column, as gtos_embed_rows_fwd)
//   onehot[n, 0:vp]  = e_token (optional, bf16): the left operand that turns the embedding's index_add backward into an MFMA product
//   tokens[n]        = token (optional)
template <typename T>
__global__ __launch_bounds__(256) void embed_packed_paths_kernel(int L, int R, int64_t n_rows, const int64_t* __restrict__ bank,
                                                                 const int* __restrict__ order, const int* __restrict__ offs,
                                                                 const float* __restrict__ table, int dim, int dim_pad, T* __restrict__ x,
                                                                 float p_drop, uint64_t seed, bf16_t* __restrict__ onehot, int vp,
                                                                 int64_t* __restrict__ tokens) {
    if (p_drop > 0.f) seed = live_seed(seed);
    __shared__ int so[66];
    for (int i = threadIdx.x; i <= L; i += 256) so[i] = offs[i];
    __syncthreads();
    const int vpr = dim_pad / 8, ovr = vp / 8;
    const float ks = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_rows * vpr; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / vpr; const int cv = (int)(t % vpr), c = cv * 8;
        int lo = 0, hi = L;                                  // the step whose range holds `row`
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (so[mid] <= row) lo = mid; else hi = mid; }
        const int64_t tk = bank[(int64_t)lo * R + order[row - so[lo]]];
        const float* src = table + tk * dim;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float xv = (c + e < dim) ? src[c + e] : 0.f;
            if (p_drop > 0.f) xv = drop_keep(seed, (uint64_t)row * dim_pad + c + e, p_drop) ? xv * ks : 0.f;
            v[e] = xv;
        }
        Vec8<T>::store(x + row * dim_pad + c, v);
        if (onehot) {
            for (int oc = cv; oc < ovr; oc += vpr) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (oc * 8 + e == tk) ? 1.f : 0.f;
                Vec8<bf16_t>::store(onehot + row * vp + oc * 8, o);
            }
        }
        if (tokens && cv == 0) tokens[row] = tk;
    }
}

// dtable[tok[n], :] += dropmask * dout[n, 0:dim].  The table is small (relation vocabulary ~ 90 x 100): each block
// accumulates a private copy in LDS (ds_add_f32) over its slice of rows and flushes it with global atomics.
// LDS layout [e][token][chunk] (channel = chunk * 8 + e): a lane owns the 8 channels of one 16-byte chunk and issues one
// LDS atomic per e, so in every wave instruction the lanes of a row hit CONSECUTIVE words -- with the natural [token][channel]
// layout they were 8 words apart, i.e. 4 of the 32 banks, and the relation table's two launches cost 0.6 + 0.3 ms per step.
template <typename T, bool use_lds>
__global__ __launch_bounds__(256) void embed_rows_bwd_kernel(int64_t n, int V, int dim, int dim_pad, const int64_t* __restrict__ tok,
                                                             const T* __restrict__ dout, float* __restrict__ dtable, float p_drop,
                                                             uint64_t seed, int64_t rows_per_block, float* __restrict__ partials) {
    if (p_drop > 0.f) seed = live_seed(seed);
    // small tables (the relation / character vocabularies) are accumulated in a private LDS copy first; large ones take
    // fp32 atomics in global memory directly (many rows: little contention)
    extern __shared__ float tab[];
    const int vpr = dim_pad / 8;
    const int plane = V * vpr;                     // words per e
    if (use_lds) {
        for (int i = threadIdx.x; i < 8 * plane; i += 256) tab[i] = 0.f;
        __syncthreads();
    }
    const float ks = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    // four (row, chunk) items per thread and iteration: their loads are issued together, then the atomics -- one item per
    // iteration was a dependent chain (token id -> row load -> 8 atomics) with at most one wave per SIMD to hide it
    constexpr int UN = 4;
    for (int64_t t0 = r0 * vpr + threadIdx.x; t0 < r1 * vpr; t0 += 256 * UN) {
        float v[UN][8];
        int64_t tk[UN], row[UN];
        int ch[UN];
        bool ok[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t t = t0 + (int64_t)u * 256;
            ok[u] = t < r1 * vpr;
            row[u] = ok[u] ? t / vpr : r0;
            ch[u] = ok[u] ? (int)(t % vpr) : 0;
            tk[u] = tok[row[u]];
            Vec8<T>::load(dout + row[u] * dim_pad + ch[u] * 8, v[u]);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (!ok[u]) continue;
            const int c = ch[u] * 8;
            float* gdst = dtable + tk[u] * dim;
            float* ldst = tab + tk[u] * vpr + ch[u];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (c + e < dim) {
                    float x = v[u][e];
                    if (p_drop > 0.f) x = drop_keep(seed, (uint64_t)row[u] * dim_pad + c + e, p_drop) ? x * ks : 0.f;
                    if (use_lds) atomicAdd(ldst + e * plane, x);
                    else if (x != 0.f) atomicAdd(gdst + c + e, x);
                }
            }
        }
    }
    if (use_lds) {
        __syncthreads();
        if (partials) {          // two-level: the block's table goes to its own row of the workspace, embed_reduce_kernel adds the rows up
            float* mine = partials + (int64_t)blockIdx.x * 8 * plane;
            for (int i = threadIdx.x; i < 8 * plane; i += 256) mine[i] = tab[i];
            return;
        }
        for (int i = threadIdx.x; i < 8 * plane; i += 256) {
            const float x = tab[i];
            const int e = i / plane, rem = i % plane, tk = rem / vpr, chn = (rem % vpr) * 8 + e;
            if (x != 0.f && chn < dim) atomicAdd(dtable + (int64_t)tk * dim + chn, x);
        }
    }
}

// dtable += sum over the blocks' private tables (LDS layout [e][token][chunk]).  Every block flushing its V*dim words with global
// atomics onto the SAME V*dim addresses serialises per address: 850 blocks at the relation table took 0.4 ms for 95 MB of input.
__global__ void embed_reduce_kernel(int nb, int V, int dim, int dim_pad, const float* __restrict__ partials, float* __restrict__ dtable) {
    const int vpr = dim_pad / 8, plane = V * vpr, total = 8 * plane;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b0 = (int)((int64_t)nb * blockIdx.y / gridDim.y), b1 = (int)((int64_t)nb * (blockIdx.y + 1) / gridDim.y);
    float acc = 0.f;
    for (int b = b0; b < b1; ++b) acc += partials[(int64_t)b * total + i];
    const int e = i / plane, rem = i % plane, tk = rem / vpr, chn = (rem % vpr) * 8 + e;
    if (chn < dim && acc != 0.f) atomicAdd(dtable + (int64_t)tk * dim + chn, acc);   // (slices of the block range; two streams may share dtable)
}

// =========================================================================== segmented row sums (trie GRU backward)
// dst[node, 0:W] = sum over the node's rows of src[row, 0:W]  (bf16 rows, fp32 accumulation, one rounding).  The rows of a
// node are entries [start, start+cnt) of `rows` (or the consecutive rows start.. when rows == NULL); a node with several
// chunks ("heavy": a trie node near the root is the prefix of thousands of paths) accumulates its chunks with fp32 atomics
// in heavy[slot, 0:W] and is written by seg_finish_kernel.  One WAVE per chunk: a lane owns 8 channels of each 512-channel
// slab, so a row is read with 1 KB coalesced wave loads; four row loads are in flight per lane.
template <int SLABS, int NSRC>
__global__ __launch_bounds__(256) void seg_sum_kernel(int n_chunks, const int* __restrict__ rows, const int* __restrict__ chunk_node,
                                                      const int* __restrict__ chunk_start, const int* __restrict__ chunk_cnt,
                                                      const int* __restrict__ chunk_slot, const bf16_t* __restrict__ src0,
                                                      const bf16_t* __restrict__ src1, int64_t ld_src,
                                                      int W, bf16_t* __restrict__ dst0, bf16_t* __restrict__ dst1, int64_t ld_dst,
                                                      float* __restrict__ heavy0, float* __restrict__ heavy1) {
    // NSRC = 2: the same row lists reduce two source matrices in one pass (one index fetch, twice the bytes per gathered row).
    // Most chunks hold a handful of rows, so a chunk is a chain of three dependent loads (chunk record -> row ids -> rows).
    // The waves walk the chunk list grid-stride and software-pipeline the chain: while chunk i is summed, the row ids of
    // chunk i+1 (one coalesced load, a lane per row, broadcast with shuffles) and the record of chunk i+2 are in flight.
    const int lane = threadIdx.x & 63;
    const int stride = gridDim.x * 4;
    int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ch >= n_chunks) return;
    struct Meta { int node, start, cnt, slot; };
    const int last = n_chunks - 1;
    auto load_meta = [&](int c) {
        const int cc = c < last ? c : last;
        Meta m;
        m.node = chunk_node[cc]; m.start = chunk_start[cc]; m.cnt = chunk_cnt[cc]; m.slot = chunk_slot ? chunk_slot[cc] : -1;
        return m;
    };
    auto load_ids = [&](const Meta& m) { return rows ? (lane < m.cnt ? rows[m.start + lane] : 0) : m.start + lane; };
    const bf16_t* srcs[2] = {src0, src1};
    bf16_t* dsts[2] = {dst0, dst1};
    float* heavies[2] = {heavy0, heavy1};
    Meta m1 = load_meta(ch), m2 = load_meta(ch + stride);
    int ids1 = load_ids(m1);
    for (; ch < n_chunks; ch += stride) {
        const int ids2 = load_ids(m2);
        const Meta m3 = load_meta(ch + 2 * stride);
        const int cnt = m1.cnt;
        float acc[NSRC][SLABS][8];
#pragma unroll
        for (int q = 0; q < NSRC; ++q)
#pragma unroll
            for (int sl = 0; sl < SLABS; ++sl)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[q][sl][e] = 0.f;
        constexpr int UN = NSRC == 2 ? 2 : 4;
        for (int i0 = 0; i0 < cnt; i0 += UN) {
            uint4 raw[UN][NSRC][SLABS];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t r = __shfl(ids1, i0 + u < cnt ? i0 + u : 0);
#pragma unroll
                for (int q = 0; q < NSRC; ++q)
#pragma unroll
                    for (int sl = 0; sl < SLABS; ++sl) {
                        const int chn = sl * 512 + lane * 8;
                        raw[u][q][sl] = (chn < W && i0 + u < cnt) ? *reinterpret_cast<const uint4*>(srcs[q] + r * ld_src + chn)
                                                                  : make_uint4(0, 0, 0, 0);
                    }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int q = 0; q < NSRC; ++q)
#pragma unroll
                    for (int sl = 0; sl < SLABS; ++sl) {
                        const uint4 v = raw[u][q][sl];
                        acc[q][sl][0] += lo_bf(v.x); acc[q][sl][1] += hi_bf(v.x); acc[q][sl][2] += lo_bf(v.y); acc[q][sl][3] += hi_bf(v.y);
                        acc[q][sl][4] += lo_bf(v.z); acc[q][sl][5] += hi_bf(v.z); acc[q][sl][6] += lo_bf(v.w); acc[q][sl][7] += hi_bf(v.w);
                    }
        }
#pragma unroll
        for (int q = 0; q < NSRC; ++q)
#pragma unroll
            for (int sl = 0; sl < SLABS; ++sl) {
                const int chn = sl * 512 + lane * 8;
                if (chn < W) {
                    if (m1.slot < 0) {
                        Vec8<bf16_t>::store(dsts[q] + (int64_t)m1.node * ld_dst + chn, acc[q][sl]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) atomicAdd(heavies[q] + (int64_t)m1.slot * W + chn + e, acc[q][sl][e]);
                    }
                }
            }
        m1 = m2; ids1 = ids2; m2 = m3;
    }
}

// Streaming variant (round 3).  The chunk list is a CSR: chunks in node order, their rows consecutive in `rows`.  seg_sum_kernel
// gives a wave one CHUNK at a time, so a wave of one- and two-row chunks (most trie nodes) has 1.5-3 KB in flight and the kernel
// sits at 4.4 TB/s.  Here a wave owns a contiguous RANGE of chunks holding about the same number of rows as every other wave's
// (wave_off, built with the trie) and streams through its rows UN at a time regardless of the chunk boundaries: the loads of
// a group never wait for a chunk record, and every wave keeps UN rows in flight.  The sums are flushed where a chunk ends (a
// wave-uniform test on the rows left in the chunk; chunk records and row ids arrive 64 at a time, one block ahead).  A row of
// W = SLABS * 512 + (HALF ? 256 : 0) channels: a lane owns 8 channels of every full slab and 4 of the half slab, so W = 768 keeps
// all 64 lanes loading (the slab version idles half of them on its second slab).
// (separate const __restrict__ parameters: what lets hipcc read the wave-uniform chunk records with s_load)
struct SegStreamArgs {
    int n_chunks, total_rows, n_waves;
    const int *rows, *chunk_node, *chunk_start, *chunk_cnt, *chunk_slot, *wave_off;
    const bf16_t* src;
    int64_t ld_src;
    bf16_t* dst;
    int64_t ld_dst;
    float* heavy;
};

template <int SLABS, bool HALF>
__global__ __launch_bounds__(256) void seg_sum_stream_kernel(int n_chunks, int total_rows, int n_waves, const int* __restrict__ rows_,
                                                             const int* __restrict__ chunk_node_, const int* __restrict__ chunk_start_,
                                                             const int* __restrict__ chunk_cnt_, const int* __restrict__ chunk_slot_,
                                                             const int* __restrict__ wave_off_, const bf16_t* __restrict__ src_,
                                                             int64_t ld_src, bf16_t* __restrict__ dst_, int64_t ld_dst,
                                                             float* __restrict__ heavy_) {
    struct { int n_chunks, total_rows, n_waves; const int* __restrict__ rows; const int* __restrict__ chunk_node;
             const int* __restrict__ chunk_start; const int* __restrict__ chunk_cnt; const int* __restrict__ chunk_slot;
             const int* __restrict__ wave_off; const bf16_t* __restrict__ src; int64_t ld_src; bf16_t* __restrict__ dst; int64_t ld_dst;
             float* __restrict__ heavy; } a = {n_chunks, total_rows, n_waves, rows_, chunk_node_, chunk_start_, chunk_cnt_, chunk_slot_,
                                               wave_off_, src_, ld_src, dst_, ld_dst, heavy_};
    constexpr int W = SLABS * 512 + (HALF ? 256 : 0);
    constexpr int UN = W <= 768 ? 8 : 4;
    constexpr int NF = SLABS > 0 ? SLABS : 1;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (w >= a.n_waves) return;
    int c = a.wave_off[w];
    const int c1 = a.wave_off[w + 1];
    if (c >= c1) return;
    int p = a.chunk_start[c];
    const int p1 = c1 < a.n_chunks ? a.chunk_start[c1] : a.total_rows;
    auto ids_at = [&](int base) {                    // (clamped, unconditional: a load under a lane mask makes hipcc drain vmcnt at the join)
        const int q = base + lane;
        return a.rows[q < a.total_rows ? q : a.total_rows - 1];
    };
    int pb = p;
    int idC = ids_at(pb);
    int idN = ids_at(pb + 64);
    float accF[NF][8], accH[4];
#pragma unroll
    for (int sl = 0; sl < NF; ++sl)
#pragma unroll
        for (int e = 0; e < 8; ++e) accF[sl][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) accH[e] = 0.f;
    // chunk records through the SCALAR cache (c is wave-uniform), one chunk ahead: their waits are lgkmcnt waits.  Read as vector
    // loads a block ahead they shared vmcnt with the flushes' stores, and every wait for a record became a wait for those.
    int rem = 0, node = 0, slot = -1, node_n, cnt_n, slot_n;
    auto load_next = [&](int cn) {
        cn = cn < a.n_chunks ? cn : a.n_chunks - 1;
        node_n = a.chunk_node[cn]; cnt_n = a.chunk_cnt[cn]; slot_n = a.chunk_slot ? a.chunk_slot[cn] : -1;
    };
    auto fetch_chunk = [&]() {                       // make chunk c (c < c1) current, start the loads of the record of chunk c + 1
        node = node_n; rem = cnt_n; slot = slot_n;
        load_next(c + 1);
    };
    load_next(c);
    auto flush = [&]() {
        if (slot < 0) {
            bf16_t* out = a.dst + (int64_t)node * a.ld_dst;
#pragma unroll
            for (int sl = 0; sl < SLABS; ++sl) Vec8<bf16_t>::store(out + sl * 512 + lane * 8, accF[sl]);
            if (HALF) *reinterpret_cast<uint2*>(out + SLABS * 512 + lane * 4) = make_uint2(pack_bf(accH[0], accH[1]), pack_bf(accH[2], accH[3]));
        } else {
            float* out = a.heavy + (int64_t)slot * W;
#pragma unroll
            for (int sl = 0; sl < SLABS; ++sl)
#pragma unroll
                for (int e = 0; e < 8; ++e) atomicAdd(out + sl * 512 + lane * 8 + e, accF[sl][e]);
            if (HALF) {
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(out + SLABS * 512 + lane * 4 + e, accH[e]);
            }
        }
#pragma unroll
        for (int sl = 0; sl < NF; ++sl)
#pragma unroll
            for (int e = 0; e < 8; ++e) accF[sl][e] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) accH[e] = 0.f;
    };
    fetch_chunk();
    while (rem == 0) {                               // a node without rows still gets its (zero) result
        flush();
        if (++c >= c1) return;
        fetch_chunk();
    }
    for (;; pb += 64) {                              // one block of 64 row ids per outer iteration, the next one in flight
      const int pe = pb + 64 < p1 ? pb + 64 : p1;
      for (; p < pe; p += UN) {
        uint4 rawF[UN][NF];
        uint2 rawH[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool ok = p + u < p1;                                   // wave-uniform
            const int64_t r = __builtin_amdgcn_readlane(idC, (p - pb + u) & 63);
            const bf16_t* rp = a.src + r * a.ld_src;
#pragma unroll
            for (int sl = 0; sl < SLABS; ++sl)
                rawF[u][sl] = ok ? *reinterpret_cast<const uint4*>(rp + sl * 512 + lane * 8) : make_uint4(0, 0, 0, 0);
            if (HALF) rawH[u] = ok ? *reinterpret_cast<const uint2*>(rp + SLABS * 512 + lane * 4) : make_uint2(0, 0);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (p + u < p1) {
#pragma unroll
                for (int sl = 0; sl < SLABS; ++sl) {
                    const uint4 v = rawF[u][sl];
                    accF[sl][0] += lo_bf(v.x); accF[sl][1] += hi_bf(v.x); accF[sl][2] += lo_bf(v.y); accF[sl][3] += hi_bf(v.y);
                    accF[sl][4] += lo_bf(v.z); accF[sl][5] += hi_bf(v.z); accF[sl][6] += lo_bf(v.w); accF[sl][7] += hi_bf(v.w);
                }
                if (HALF) {
                    const uint2 v = rawH[u];
                    accH[0] += lo_bf(v.x); accH[1] += hi_bf(v.x); accH[2] += lo_bf(v.y); accH[3] += hi_bf(v.y);
                }
                if (--rem == 0) {
                    do {
                        flush();
                        if (++c >= c1) break;
                        fetch_chunk();
                    } while (rem == 0);
                }
            }
        }
      }
      if (p >= p1) break;
      idC = idN;
      idN = ids_at(pb + 128);
    }
}

// contiguous variant: segment s sums the consecutive rows ranges[2s] .. ranges[2s+1]-1 (the children of a trie node) into
// dst row s; an empty range writes zeros
template <int SLABS>
__global__ __launch_bounds__(256) void seg_sum_range_kernel(int n_seg, const int* __restrict__ ranges, const bf16_t* __restrict__ src,
                                                            int64_t ld_src, int W, bf16_t* __restrict__ dst, int64_t ld_dst) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_seg) return;
    const int lo = ranges[2 * s], hi = ranges[2 * s + 1];
    float acc[SLABS][8];
#pragma unroll
    for (int sl = 0; sl < SLABS; ++sl)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[sl][e] = 0.f;
    for (int64_t r = lo; r < hi; ++r) {
#pragma unroll
        for (int sl = 0; sl < SLABS; ++sl) {
            const int ch = sl * 512 + lane * 8;
            if (ch < W) {
                float v[8];
                Vec8<bf16_t>::load(src + r * ld_src + ch, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[sl][e] += v[e];
            }
        }
    }
#pragma unroll
    for (int sl = 0; sl < SLABS; ++sl) {
        const int ch = sl * 512 + lane * 8;
        if (ch < W) Vec8<bf16_t>::store(dst + (int64_t)s * ld_dst + ch, acc[sl]);
    }
}

__global__ void seg_finish_kernel(int n_heavy, const int* __restrict__ heavy_node, const float* __restrict__ heavy, int W,
                                  bf16_t* __restrict__ dst, int64_t ld_dst) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int vpr = W / 8;
    if (t >= (int64_t)n_heavy * vpr) return;
    const int s = (int)(t / vpr), ch = (int)(t % vpr) * 8;
    float v[8];
    Vec8<float>::load(heavy + (int64_t)s * W + ch, v);
    Vec8<bf16_t>::store(dst + (int64_t)heavy_node[s] * ld_dst + ch, v);
}

// =========================================================================== optimizer (flat buffers)
// sum of squares of g (for clip_grad_norm_, /root/reference/generator/train.py:151)
__global__ __launch_bounds__(256) void sqnorm_kernel(int64_t n, const float* __restrict__ g, float* __restrict__ out) {
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s = fmaf(g[i], g[i], s);
    s = wave_sum(s);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// AdamWeightDecayOptimizer.step (/root/reference/generator/adam.py:66-87): no bias correction, decoupled decay added
// to the update before the lr multiply.  g is first scaled by gscale (1/world_size) and by the clip coefficient
// min(1, max_norm / (gscale*sqrt(sqnorm) + 1e-6)) read from device memory (no host sync).
__global__ void adam_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, float lr, float b1, float b2, float eps, float wd, float gscale,
                            const float* __restrict__ sqnorm, float max_norm, bf16_t* __restrict__ mirror,
                            const float* __restrict__ ctl) {
    if (ctl) {                                   // device-side step control: {lr of this step, skip flag} (step_control_kernel)
        if (ctl[1] != 0.f) return;               // batch discarded: parameters and moments stay as they are
        lr = ctl[0];
    }
    float coef = gscale;
    if (sqnorm) { const float nrm = gscale * sqrtf(*sqnorm); coef *= fminf(1.f, max_norm / (nrm + 1e-6f)); }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        const float pi = p[i] - lr * (mi / (sqrtf(vi) + eps) + wd * p[i]);
        m[i] = mi; v[i] = vi; p[i] = pi;
        if (mirror) mirror[i] = f2bf(pi);
    }
}

// The abnormal-loss rule and the learning-rate schedule of the training loop (/root/reference/generator/train.py:81-83,
// 142-148) on the device, so the step needs no host read of the loss between forward and backward:
//   phase 0: flag = batches_acm > warmup && loss > 5 * loss_acm / batches_acm        (this rank's decision)
//   (data parallel: the caller all-reduces `flag` with MAX -- a rank-local skip would desynchronise the collectives)
//   phase 1: flag != 0 -> discarded += 1, ctl = {*, 1};  else loss_acm += loss, batches_acm += 1,
//            ctl = {embed^-0.5 * min(s^-0.5, s * warmup^-1.5) with s = batches_acm, 0}
// state = {loss_acm, batches_acm, discarded} in double (the reference keeps Python floats).
__global__ void step_control_kernel(int phase, const float* __restrict__ loss, double* __restrict__ state, float* __restrict__ flag,
                                    double warmup, double embed_dim, float* __restrict__ ctl) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double l = (double)*loss;
    if (phase == 0) {
        *flag = (state[1] > warmup && l > 5.0 * (state[0] / state[1])) ? 1.f : 0.f;
        return;
    }
    if (*flag != 0.f) {
        state[2] += 1.0;
        ctl[1] = 1.f;
        return;
    }
    state[0] += l;
    state[1] += 1.0;
    const double s = state[1];
    const double a = 1.0 / sqrt(s), b = s / (warmup * sqrt(warmup));
    ctl[0] = (float)((1.0 / sqrt(embed_dim)) * (a < b ? a : b));
    ctl[1] = 0.f;
}

// All 2-D weights' transposes in ONE launch: the bf16 mirror of the flat parameter buffer holds every weight [out, in]; dX = dY W
// runs as an NT product on W^T [in, out], which used to cost one ATen transpose-copy launch per weight and step (~60).  desc[m] =
// {offset of matrix m in the flat buffers (elements), rows, cols}; tile_start[m] = number of 32x32 tiles before matrix m.  The
// transposed copy sits at the same offset of a second flat buffer.
__global__ __launch_bounds__(256) void transpose_batch_kernel(int n_mat, const int64_t* __restrict__ desc, const int* __restrict__ tile_start,
                                                              const bf16_t* __restrict__ src, bf16_t* __restrict__ dst) {
    __shared__ bf16_t tile[32][33];
    const int t = blockIdx.x;
    int lo = 0, hi = n_mat - 1;                      // last m with tile_start[m] <= t
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tile_start[mid] <= t) lo = mid; else hi = mid - 1; }
    const int64_t off = desc[3 * lo];
    const int rows = (int)desc[3 * lo + 1], cols = (int)desc[3 * lo + 2];
    const int tc_n = (cols + 31) / 32, lt = t - tile_start[lo];
    const int r0 = (lt / tc_n) * 32, c0 = (lt % tc_n) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8 threads
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 8 * k][tx] = src[off + (int64_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;             // dst [cols, rows]
        if (r < rows && c < cols) dst[off + (int64_t)c * rows + r] = tile[tx][ty + 8 * k];
    }
}

__global__ void cast_bf16_kernel(int64_t n, const float* __restrict__ src, bf16_t* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = f2bf(src[i]);
}

inline int grid_for(int64_t work, int block) { int64_t g = (work + block - 1) / block; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

}  // namespace

// dtype triple -> kernel instance.  Supported: all three equal (fp32 parity mode / plain bf16), and the fp32 residual stream of bf16
// activations: x fp32 or bf16, r bf16 (or none), y fp32.
#define GTOS_LN_COMBOS(X)              \
    X(GTOS_F32, GTOS_F32, GTOS_F32, float, float, float)        \
    X(GTOS_BF16, GTOS_BF16, GTOS_BF16, bf16_t, bf16_t, bf16_t)  \
    X(GTOS_F32, GTOS_BF16, GTOS_F32, float, bf16_t, float)      \
    X(GTOS_BF16, GTOS_BF16, GTOS_F32, bf16_t, bf16_t, float)

extern "C" int gtos_ln_residual_fwd2(int x_dtype, int r_dtype, int y_dtype, int rows, int d, const void* x, const void* r, float p_drop,
                                     uint64_t seed, const float* gamma, const float* beta, float eps, void* y, void* y2_bf16,
                                     float* mean, float* rstd, void* stream) {
    if (d % 8 || d > 512 * LN_MAXC) return -20;
    if (rows <= 0) return 0;
    if ((uintptr_t)x % 16 || (uintptr_t)r % 16 || (uintptr_t)y % 16 || (uintptr_t)y2_bf16 % 16) return -25;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((rows + 3) / 4), block(256);
#define X(dx_, dr_, dy_, TX, TR, TY)                                                                                                   \
    if (x_dtype == dx_ && r_dtype == dr_ && y_dtype == dy_) {                                                                          \
        hipLaunchKernelGGL((ln_fwd_kernel<TX, TR, TY>), grid, block, 0, s, rows, d, (const TX*)x, (const TR*)r, p_drop, seed, gamma,   \
                           beta, eps, (TY*)y, (bf16_t*)y2_bf16, mean, rstd);                                                           \
        GTOS_CHECK_LAUNCH();                                                                                                           \
        return 0;                                                                                                                      \
    }
    GTOS_LN_COMBOS(X)
#undef X
    return -21;
}

extern "C" int gtos_ln_residual_fwd(int dtype, int rows, int d, const void* x, const void* r, float p_drop, uint64_t seed,
                                    const float* gamma, const float* beta, float eps, void* y, float* mean, float* rstd,
                                    void* stream) {
    return gtos_ln_residual_fwd2(dtype, dtype, dtype, rows, d, x, r, p_drop, seed, gamma, beta, eps, y, nullptr, mean, rstd, stream);
}

extern "C" int gtos_ln_residual_bwd2(int x_dtype, int r_dtype, int y_dtype, int rows, int d, const void* dy, const void* dy2_bf16,
                                     const void* x, const void* r, float p_drop, uint64_t seed, const float* gamma, const float* mean,
                                     const float* rstd, void* dx, void* dr, float* dgamma, float* dbeta, void* stream) {
    if (d % 8 || d > 512 * LN_MAXC) return -20;
    if (rows <= 0) return 0;
    if (!dy && !dy2_bf16) return -23;
    if ((uintptr_t)x % 16 || (uintptr_t)r % 16 || (uintptr_t)dy % 16 || (uintptr_t)dy2_bf16 % 16 || (uintptr_t)dx % 16 || (uintptr_t)dr % 16) return -25;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int nb = (rows + 31) / 32; if (nb > 256) nb = 256; if (nb < 1) nb = 1;   // >= 8 rows per wave: few atomics
    dim3 grid(nb), block(256);
#define X(dx_, dr_, dy_, TX, TR, TY)                                                                                                   \
    if (x_dtype == dx_ && r_dtype == dr_ && y_dtype == dy_) {                                                                          \
        hipLaunchKernelGGL((ln_bwd_kernel<TX, TR, TY>), grid, block, 0, s, rows, d, (const TY*)dy, (const bf16_t*)dy2_bf16,            \
                           (const TX*)x, (const TR*)r, p_drop, seed, gamma, mean, rstd, (TX*)dx, (TR*)dr, dgamma, dbeta);              \
        GTOS_CHECK_LAUNCH();                                                                                                           \
        return 0;                                                                                                                      \
    }
    GTOS_LN_COMBOS(X)
#undef X
    return -21;
}

extern "C" int gtos_ln_residual_bwd(int dtype, int rows, int d, const void* dy, const void* x, const void* r, float p_drop,
                                    uint64_t seed, const float* gamma, const float* mean, const float* rstd,
                                    void* dx, void* dr, float* dgamma, float* dbeta, void* stream) {
    return gtos_ln_residual_bwd2(dtype, dtype, dtype, rows, d, dy, nullptr, x, r, p_drop, seed, gamma, mean, rstd, dx, dr, dgamma, dbeta,
                                 stream);
}

extern "C" int gtos_relu_dropout_bwd(int dtype, int64_t n, void* dh, const void* h, float p_drop, void* stream) {
    if (n % 8) return -21;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float ks = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    dim3 grid(grid_for(n / 8, 256)), block(256);
    if (dtype == GTOS_BF16) hipLaunchKernelGGL(relu_drop_bwd_kernel<bf16_t>, grid, block, 0, s, n / 8, (bf16_t*)dh, (const bf16_t*)h, ks);
    else hipLaunchKernelGGL(relu_drop_bwd_kernel<float>, grid, block, 0, s, n / 8, (float*)dh, (const float*)h, ks);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_colsum(int dtype, int rows, int N, int64_t ld, const void* dy, float* out, void* stream) {
    if (rows <= 0 || N <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int gx = (N + 511) / 512;
    int rpb = (rows + 1023) / 1024; if (rpb < 64) rpb = 64;
    dim3 grid(gx, (rows + rpb - 1) / rpb), block(256);
    if (dtype == GTOS_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, block, 0, s, rows, N, ld, (const bf16_t*)dy, out, rpb);
    else hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, s, rows, N, ld, (const float*)dy, out, rpb);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_gru_cell_fwd(int dtype, int rows, int hs, const void* xg, const void* hg, void* h, void* y, int64_t ldy,
                                 void* hprev_save, void* gates, float p_drop, uint64_t seed, int64_t drop_base, void* stream) {
    if (hs % 8) return -22;
    if (rows <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid(grid_for((int64_t)rows * hs / 8, 256)), block(256);
    if (dtype == GTOS_BF16) hipLaunchKernelGGL(gru_fwd_kernel<bf16_t>, grid, block, 0, s, rows, hs, (const bf16_t*)xg, (const bf16_t*)hg, (bf16_t*)h, (bf16_t*)y, ldy, (bf16_t*)hprev_save, (bf16_t*)gates, p_drop, seed, drop_base);
    else hipLaunchKernelGGL(gru_fwd_kernel<float>, grid, block, 0, s, rows, hs, (const float*)xg, (const float*)hg, (float*)h, (float*)y, ldy, (float*)hprev_save, (float*)gates, p_drop, seed, drop_base);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_gru_cell_bwd(int dtype, int rows, int hs, const void* gates, const void* hprev, const void* dy, int64_t ldy,
                                 float* dh, void* dxg, void* dhg, float p_drop, uint64_t seed, int64_t drop_base,
                                 float* bias_partials, int n_partials, void* stream) {
    if (hs % 8) return -22;
    if (rows <= 0) return 0;
    if (bias_partials && (256 % (hs / 8) != 0 || n_partials < 1)) return -26;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int nb = grid_for((int64_t)rows * hs / 8, 256);
    if (bias_partials && nb > n_partials) nb = n_partials;
    dim3 grid(nb), block(256);
    const size_t sh = bias_partials ? (size_t)4 * hs * sizeof(float) : 0;
    if (dtype == GTOS_BF16) hipLaunchKernelGGL(gru_bwd_kernel<bf16_t>, grid, block, sh, s, rows, hs, (const bf16_t*)gates, (const bf16_t*)hprev, (const bf16_t*)dy, ldy, dh, (bf16_t*)dxg, (bf16_t*)dhg, p_drop, seed, drop_base, bias_partials);
    else hipLaunchKernelGGL(gru_bwd_kernel<float>, grid, block, sh, s, rows, hs, (const float*)gates, (const float*)hprev, (const float*)dy, ldy, dh, (float*)dxg, (float*)dhg, p_drop, seed, drop_base, bias_partials);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_relation_gather_mean(int dtype, int64_t P, int K, int d, const void* bank, const int64_t* idx, int zero_row0,
                                         void* out, void* stream) {
    if (d % 8) return -23;
    if (P <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid(grid_for(P * (d / 8), 256)), block(256);
    if (dtype == GTOS_BF16) hipLaunchKernelGGL(gather_mean_kernel<bf16_t>, grid, block, 0, s, P, K, d, (const bf16_t*)bank, idx, zero_row0, (bf16_t*)out);
    else hipLaunchKernelGGL(gather_mean_kernel<float>, grid, block, 0, s, P, K, d, (const float*)bank, idx, zero_row0, (float*)out);
    GTOS_CHECK_LAUNCH();
    return 0;
}


extern "C" int gtos_embed_rows_fwd(int dtype, int64_t n, int dim, int dim_pad, const int64_t* tok, const float* table, void* out,
                                   float p_drop, uint64_t seed, void* stream) {
    if (dim_pad % 8 || dim_pad < dim) return -24;
    if (n <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid(grid_for(n * (dim_pad / 8), 256)), block(256);
    if (dtype == GTOS_BF16) hipLaunchKernelGGL(embed_rows_fwd_kernel<bf16_t>, grid, block, 0, s, n, dim, dim_pad, tok, table, (bf16_t*)out, p_drop, seed);
    else hipLaunchKernelGGL(embed_rows_fwd_kernel<float>, grid, block, 0, s, n, dim, dim_pad, tok, table, (float*)out, p_drop, seed);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_embed_packed_paths(int dtype, int L, int R, int64_t n_rows, const int64_t* bank, const int* order, const int* offs,
                                       const float* table, int dim, int dim_pad, void* x, float p_drop, uint64_t seed, void* onehot, int vp,
                                       int64_t* tokens, void* stream) {
    if (dim_pad % 8 || dim_pad < dim || L < 1 || L > 64 || R < 1) return -24;
    if (onehot && (vp < 8 || vp % 8 || (uintptr_t)onehot % 16)) return -24;
    if (!bank || !order || !offs || !table || !x || (uintptr_t)x % 16) return -23;
    if (n_rows <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid(grid_for(n_rows * (dim_pad / 8), 256)), block(256);
    if (dtype == GTOS_BF16)
        hipLaunchKernelGGL(embed_packed_paths_kernel<bf16_t>, grid, block, 0, s, L, R, n_rows, bank, order, offs, table, dim, dim_pad, (bf16_t*)x,
                           p_drop, seed, (bf16_t*)onehot, vp, tokens);
    else
        hipLaunchKernelGGL(embed_packed_paths_kernel<float>, grid, block, 0, s, L, R, n_rows, bank, order, offs, table, dim, dim_pad, (float*)x,
                           p_drop, seed, (bf16_t*)onehot, vp, tokens);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_embed_rows_bwd(int dtype, int64_t n, int V, int dim, int dim_pad, const int64_t* tok, const void* dout,
                                   float* dtable, float p_drop, uint64_t seed, float* workspace, int64_t workspace_bytes, void* stream) {
    if (dim_pad % 8 || dim_pad < dim) return -24;
    const int use_lds = (size_t)V * dim_pad * 4 <= 60 * 1024;     // private LDS table, else global atomics
    if (n <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // rows per block: 512 -- at 2048 the relation table's 457 k rows made 224 blocks, under one wave per SIMD
    int64_t nb = (n + 511) / 512; if (nb > 2048) nb = 2048; if (nb < 1) nb = 1;
    const int64_t table_bytes = (int64_t)V * dim_pad * 4;
    float* partials = nullptr;
    if (use_lds && workspace && nb > 8) {                          // two-level flush when the caller lends a workspace
        if (workspace_bytes < table_bytes) return -24;
        const int64_t fit = workspace_bytes / table_bytes;
        if (nb > fit) nb = fit;
        partials = workspace;
    }
    const int64_t rpb = (n + nb - 1) / nb;
    dim3 grid((unsigned)((n + rpb - 1) / rpb)), block(256);
    const size_t sh = use_lds ? (size_t)table_bytes : 0;
#define GTOS_EMB_BWD(T, L) hipLaunchKernelGGL((embed_rows_bwd_kernel<T, L>), grid, block, sh, s, n, V, dim, dim_pad, tok, (const T*)dout, dtable, p_drop, seed, rpb, partials)
    if (dtype == GTOS_BF16) { if (use_lds) GTOS_EMB_BWD(bf16_t, true); else GTOS_EMB_BWD(bf16_t, false); }
    else { if (use_lds) GTOS_EMB_BWD(float, true); else GTOS_EMB_BWD(float, false); }
#undef GTOS_EMB_BWD
    GTOS_CHECK_LAUNCH();
    if (partials) {
        const int total = V * dim_pad;
        const int slices = grid.x >= 64 ? 16 : 1;       // 16 slices of the block range per word: ~9 k x 16 short loops instead of 9 k long ones
        hipLaunchKernelGGL(embed_reduce_kernel, dim3((total + 255) / 256, slices), dim3(256), 0, s, (int)grid.x, V, dim, dim_pad, partials, dtable);
        GTOS_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int gtos_segment_sum_rows(int n_chunks, const int* rows, const int* chunk_node, const int* chunk_start, const int* chunk_cnt,
                                     const int* chunk_slot, const void* src, const void* src2, int64_t ld_src, int width,
                                     void* dst, void* dst2, int64_t ld_dst, float* heavy, float* heavy2, void* stream) {
    if (n_chunks <= 0) return 0;
    if (width <= 0 || width % 8 || width > 1536 || ld_src % 8 || ld_dst % 8 || (uintptr_t)src % 16 || (uintptr_t)dst % 16 ||
        (uintptr_t)src2 % 16 || (uintptr_t)dst2 % 16) return -24;
    if (!chunk_node || !chunk_start || !chunk_cnt || !src || !dst || (chunk_slot && !heavy)) return -23;
    if (src2 && (!dst2 || (chunk_slot && !heavy2))) return -23;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nblk = (n_chunks + 3) / 4;
    const dim3 grid((unsigned)(nblk < 4096 ? nblk : 4096)), block(256);      // grid-stride: 16 resident blocks per CU
#define GTOS_SEG(S, Q) hipLaunchKernelGGL((seg_sum_kernel<S, Q>), grid, block, 0, s, n_chunks, rows, chunk_node, chunk_start, chunk_cnt, \
                                          chunk_slot, (const bf16_t*)src, (const bf16_t*)src2, ld_src, width, (bf16_t*)dst, (bf16_t*)dst2, \
                                          ld_dst, heavy, heavy2)
    if (src2) { if (width <= 512) GTOS_SEG(1, 2); else if (width <= 1024) GTOS_SEG(2, 2); else GTOS_SEG(3, 2); }
    else { if (width <= 512) GTOS_SEG(1, 1); else if (width <= 1024) GTOS_SEG(2, 1); else GTOS_SEG(3, 1); }
#undef GTOS_SEG
    GTOS_CHECK_LAUNCH();
    return 0;
}

// Streaming form of gtos_segment_sum_rows for a CSR chunk list (chunks in node order, rows consecutive): wave_off[n_waves + 1] are
// chunk indices cutting the list into ranges of about equal row counts (pathtrie.TrieSide.wave_off).  width in {256, 512, ..., 1536}.
extern "C" int gtos_segment_sum_stream(int n_chunks, int total_rows, const int* rows, const int* chunk_node, const int* chunk_start,
                                       const int* chunk_cnt, const int* chunk_slot, const int* wave_off, int n_waves, const void* src,
                                       int64_t ld_src, int width, void* dst, int64_t ld_dst, float* heavy, void* stream) {
    if (n_chunks <= 0 || n_waves <= 0) return 0;
    if (width <= 0 || width % 256 || width > 1536 || ld_src % 8 || ld_dst % 8 || (uintptr_t)src % 16 || (uintptr_t)dst % 16) return -24;
    if (!rows || !chunk_node || !chunk_start || !chunk_cnt || !wave_off || !src || !dst || (chunk_slot && !heavy) || total_rows <= 0) return -23;
    SegStreamArgs a;
    a.n_chunks = n_chunks; a.total_rows = total_rows; a.n_waves = n_waves;
    a.rows = rows; a.chunk_node = chunk_node; a.chunk_start = chunk_start; a.chunk_cnt = chunk_cnt; a.chunk_slot = chunk_slot;
    a.wave_off = wave_off; a.src = (const bf16_t*)src; a.ld_src = ld_src; a.dst = (bf16_t*)dst; a.ld_dst = ld_dst; a.heavy = heavy;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((n_waves + 3) / 4)), block(256);
#define GTOS_SEGS(S, H) hipLaunchKernelGGL((seg_sum_stream_kernel<S, H>), grid, block, 0, s, a.n_chunks, a.total_rows, a.n_waves, a.rows, \
        a.chunk_node, a.chunk_start, a.chunk_cnt, a.chunk_slot, a.wave_off, a.src, a.ld_src, a.dst, a.ld_dst, a.heavy)
    switch (width / 256) {
        case 1: GTOS_SEGS(0, true); break;
        case 2: GTOS_SEGS(1, false); break;
        case 3: GTOS_SEGS(1, true); break;
        case 4: GTOS_SEGS(2, false); break;
        case 5: GTOS_SEGS(2, true); break;
        default: GTOS_SEGS(3, false); break;
    }
#undef GTOS_SEGS
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_segment_sum_finish(int n_heavy, const int* heavy_node, const float* heavy, int width, void* dst, int64_t ld_dst,
                                       void* stream) {
    if (n_heavy <= 0) return 0;
    if (width <= 0 || width % 8 || ld_dst % 8 || !heavy_node || !heavy || !dst) return -24;
    const int64_t n = (int64_t)n_heavy * (width / 8);
    hipLaunchKernelGGL(seg_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n_heavy,
                       heavy_node, heavy, width, (bf16_t*)dst, ld_dst);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_segment_sum_ranges(int n_seg, const int* ranges, const void* src, int64_t ld_src, int width, void* dst, int64_t ld_dst,
                                       void* stream) {
    if (n_seg <= 0) return 0;
    if (width <= 0 || width % 8 || width > 1536 || ld_src % 8 || ld_dst % 8 || (uintptr_t)src % 16 || (uintptr_t)dst % 16) return -24;
    if (!ranges || !src || !dst) return -23;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((n_seg + 3) / 4)), block(256);
#define GTOS_SEG(S) hipLaunchKernelGGL(seg_sum_range_kernel<S>, grid, block, 0, s, n_seg, ranges, (const bf16_t*)src, ld_src, width, \
                                       (bf16_t*)dst, ld_dst)
    if (width <= 512) GTOS_SEG(1); else if (width <= 1024) GTOS_SEG(2); else GTOS_SEG(3);
#undef GTOS_SEG
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_sqnorm(int64_t n, const float* g, float* out, void* stream) {
    if (n <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(sqnorm_kernel, dim3(grid_for(n, 256 * 8)), dim3(256), 0, s, n, g, out);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_adam_step(int64_t n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                              float eps, float weight_decay, float gscale, const float* sqnorm, float max_norm,
                              void* bf16_mirror, void* stream) {
    if (n <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, s, n, p, g, m, v, lr, beta1, beta2, eps, weight_decay,
                       gscale, sqnorm, max_norm, (bf16_t*)bf16_mirror, (const float*)nullptr);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_adam_step_ctl(int64_t n, float* p, const float* g, float* m, float* v, const float* ctl, float beta1, float beta2,
                                  float eps, float weight_decay, float gscale, const float* sqnorm, float max_norm,
                                  void* bf16_mirror, void* stream) {
    if (n <= 0) return 0;
    if (!ctl) return -23;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, s, n, p, g, m, v, 0.f, beta1, beta2, eps, weight_decay,
                       gscale, sqnorm, max_norm, (bf16_t*)bf16_mirror, ctl);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_step_control(int phase, const float* loss, double* state, float* flag, int warmup_steps, int embed_dim,
                                 float* ctl, void* stream) {
    if ((phase != 0 && phase != 1) || !loss || !state || !flag || !ctl || warmup_steps < 1 || embed_dim < 1) return -23;
    hipLaunchKernelGGL(step_control_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), phase, loss, state, flag,
                       (double)warmup_steps, (double)embed_dim, ctl);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_cast_f32_to_bf16(int64_t n, const float* src, void* dst, void* stream) {
    if (n <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, s, n, src, (bf16_t*)dst);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_transpose_batch_bf16(int n_mat, const int64_t* desc, const int* tile_start, int total_tiles, const void* src,
                                         void* dst, void* stream) {
    if (n_mat <= 0 || total_tiles <= 0) return 0;
    if (!desc || !tile_start || !src || !dst) return -23;
    hipLaunchKernelGGL(transpose_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0, static_cast<hipStream_t>(stream), n_mat, desc,
                       tile_start, (const bf16_t*)src, (bf16_t*)dst);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_set_seed_epoch(const void* epoch) {
    int rc = gtosi_gemm_set_seed_epoch(epoch);
    if (!rc) rc = gtosi_rel_attn_set_seed_epoch(epoch);
    if (!rc) rc = gtosi_rowops_set_seed_epoch(epoch);
    if (!rc) rc = gtosi_gru_step_set_seed_epoch(epoch);
    if (!rc) rc = gtosi_tokenenc_set_seed_epoch(epoch);
    return rc;
}

extern "C" int gtos_abi_version(void) { return 21; }

GTOS_SEED_EPOCH_SETTER(rowops)
