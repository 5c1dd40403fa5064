// Fused generate/copy mixture of TokenGenerator (/root/reference/generator/decoder.py:40-63) for gfx950.
//
// The reference materialises, per target position (t, b):  softmax over the V-word vocabulary, the 2-way diverter softmax,
// gen_gate * vocabulary probabilities padded to V + (copy ids), a scatter_add_ of copy_gate * alignment weights at the
// concepts' copy ids, log(p + 1e-12) of the whole row, and a gather at the target: five passes over a [T,B,~10k] fp32
// tensor plus their autograd duals.  Training only needs p(target), so here ONE workgroup per (t, b) row
//   * reads the logits row once (max / sum-exp in fp32, DPP wave reductions + an LDS cross-wave step),
//   * sums the alignment mass of the source positions whose copy id equals the target,
//   * writes nll = -log(g * softmax[target] + c * mass + 1e-12), lse and p for the backward,
// and the backward writes d(logits) (a scaled softmax row with the target entry shifted), d(diverter logits) and
// d(alignment) in one more pass.  The inference form (work=True: the full log-likelihood row incl. copy ids) is
// copy_ll_kernel: probabilities into the output row, copy mass added with atomics (several concepts may share a copy
// id), then the log in place -- the row stays in L2.
#include "common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
    v = is_max ? wave_max(v) : wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                               // red may still be read from a previous call
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
    return r;
}

template <typename T>
__device__ __forceinline__ float row_lse(const T* __restrict__ lp, int V, bool vec, float* red) {
    float m = -INFINITY;
    if (vec) {
        for (int v = threadIdx.x * 8; v < V; v += NT * 8) {
            float x[8];
            Vec8<T>::load(lp + v, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) m = fmaxf(m, x[e]);
        }
    } else {
        for (int v = threadIdx.x; v < V; v += NT) m = fmaxf(m, to_f<T>(lp[v]));
    }
    m = block_reduce(m, true, red);
    float s = 0.f;
    if (vec) {
        for (int v = threadIdx.x * 8; v < V; v += NT * 8) {
            float x[8];
            Vec8<T>::load(lp + v, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += __expf(x[e] - m);
        }
    } else {
        for (int v = threadIdx.x; v < V; v += NT) s += __expf(to_f<T>(lp[v]) - m);
    }
    s = block_reduce(s, false, red);
    return m + __logf(s);
}

struct NllArgs {
    int T, B, V, S;
    const void* logits; int64_t ld; const void* div; const float* align; const int64_t* cp_seq; const int64_t* target;
    int64_t pad_idx;
    float* nll; float* lse; float* p_tgt;              // forward outputs
    const float* d_nll; void* d_logits; void* d_div; float* d_align;   // backward
};

template <typename T>
__global__ __launch_bounds__(NT) void copy_nll_fwd_kernel(NllArgs a) {
    __shared__ float red[NT / 64];
    const int row = blockIdx.x, b = row % a.B;
    const T* lp = static_cast<const T*>(a.logits) + (int64_t)row * a.ld;
    const bool vec = (a.V % 8 == 0) && (a.ld % 8 == 0) && ((uintptr_t)a.logits % 16 == 0);
    const float lse = row_lse<T>(lp, a.V, vec, red);
    const int64_t tgt = a.target[row];
    float mass = 0.f;
    for (int s = threadIdx.x; s < a.S; s += NT)
        if (a.cp_seq[(int64_t)s * a.B + b] == tgt) mass += a.align[(int64_t)row * a.S + s];
    mass = block_reduce(mass, false, red);
    if (threadIdx.x == 0) {
        const T* dp = static_cast<const T*>(a.div) + (int64_t)row * 2;
        const float d0 = to_f<T>(dp[0]), d1 = to_f<T>(dp[1]);
        const float g = 1.f / (1.f + __expf(d1 - d0)), c = 1.f - g;
        const float sig = (tgt >= 0 && tgt < a.V) ? __expf(to_f<T>(lp[tgt]) - lse) : 0.f;
        const float p = g * sig + c * mass;
        a.nll[row] = tgt == a.pad_idx ? 0.f : -__logf(p + 1e-12f);
        a.lse[row] = lse;
        a.p_tgt[row] = p;
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void copy_nll_bwd_kernel(NllArgs a) {
    const int row = blockIdx.x, b = row % a.B;
    const T* lp = static_cast<const T*>(a.logits) + (int64_t)row * a.ld;
    T* dl = static_cast<T*>(a.d_logits) + (int64_t)row * a.V;
    const bool vec = (a.V % 8 == 0) && (a.ld % 8 == 0) && ((uintptr_t)a.logits % 16 == 0) && ((uintptr_t)a.d_logits % 16 == 0);
    const int64_t tgt = a.target[row];
    const float lse = a.lse[row], p = a.p_tgt[row];
    const float u = tgt == a.pad_idx ? 0.f : a.d_nll[row] / (p + 1e-12f);        // dL/dp = -u
    const T* dp = static_cast<const T*>(a.div) + (int64_t)row * 2;
    const float d0 = to_f<T>(dp[0]), d1 = to_f<T>(dp[1]);
    const float g = 1.f / (1.f + __expf(d1 - d0)), c = 1.f - g;
    const bool in_v = tgt >= 0 && tgt < a.V;
    const float sig = in_v ? __expf(to_f<T>(lp[tgt]) - lse) : 0.f;
    const float coef = u * g * sig;                  // d logit_v = coef * (softmax_v - [v == tgt])
    if (vec) {
        for (int v = threadIdx.x * 8; v < a.V; v += NT * 8) {
            float x[8];
            Vec8<T>::load(lp + v, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = coef * (__expf(x[e] - lse) - ((int64_t)(v + e) == tgt ? 1.f : 0.f));
            Vec8<T>::store(dl + v, x);
        }
    } else {
        for (int v = threadIdx.x; v < a.V; v += NT)
            dl[v] = from_f<T>(coef * (__expf(to_f<T>(lp[v]) - lse) - ((int64_t)v == tgt ? 1.f : 0.f)));
    }
    float mass = 0.f;                                // recomputed: the alignment mass on the target's copy positions
    for (int s = threadIdx.x; s < a.S; s += NT) {
        const bool hit = a.cp_seq[(int64_t)s * a.B + b] == tgt;
        const float w = a.align[(int64_t)row * a.S + s];
        if (hit) mass += w;
        a.d_align[(int64_t)row * a.S + s] = hit ? -u * c : 0.f;
    }
    __shared__ float red[NT / 64];
    mass = block_reduce(mass, false, red);
    if (threadIdx.x == 0) {
        const float dg = -u * sig, dc = -u * mass, mix = g * dg + c * dc;
        T* dd = static_cast<T*>(a.d_div) + (int64_t)row * 2;
        dd[0] = from_f<T>(g * (dg - mix));
        dd[1] = from_f<T>(c * (dc - mix));
    }
}

struct LlArgs {
    int T, B, V, S, tot;
    const void* logits; int64_t ld; const void* div; const float* align; const int64_t* cp_seq; float* ll;
};

template <typename T>
__global__ __launch_bounds__(NT) void copy_ll_kernel(LlArgs a) {
    __shared__ float red[NT / 64];
    const int row = blockIdx.x, b = row % a.B;
    const T* lp = static_cast<const T*>(a.logits) + (int64_t)row * a.ld;
    float* out = a.ll + (int64_t)row * a.tot;
    const bool vec = (a.V % 8 == 0) && (a.ld % 8 == 0) && ((uintptr_t)a.logits % 16 == 0);
    const float lse = row_lse<T>(lp, a.V, vec, red);
    const T* dp = static_cast<const T*>(a.div) + (int64_t)row * 2;
    const float d0 = to_f<T>(dp[0]), d1 = to_f<T>(dp[1]);
    const float g = 1.f / (1.f + __expf(d1 - d0)), c = 1.f - g;
    for (int v = threadIdx.x; v < a.tot; v += NT) out[v] = v < a.V ? g * __expf(to_f<T>(lp[v]) - lse) : 0.f;
    __threadfence_block();
    __syncthreads();
    for (int s = threadIdx.x; s < a.S; s += NT) {
        const int64_t id = a.cp_seq[(int64_t)s * a.B + b];
        if (id >= 0 && id < a.tot) atomicAdd(out + id, c * a.align[(int64_t)row * a.S + s]);
    }
    __threadfence_block();
    __syncthreads();
    for (int v = threadIdx.x; v < a.tot; v += NT) out[v] = __logf(out[v] + 1e-12f);
}

}  // namespace

extern "C" int gtos_copy_nll_fwd(int dtype, int T, int B, int V, int S, const void* logits, int64_t ld_logits, const void* div,
                                 const float* align, const int64_t* cp_seq, const int64_t* target, int64_t pad_idx,
                                 float* nll, float* lse, float* p_tgt, void* stream) {
    if (T <= 0 || B <= 0) return 0;
    if (V <= 0 || S < 0 || ld_logits < V) return -24;
    if (!logits || !div || !target || !nll || !lse || !p_tgt || (S > 0 && (!align || !cp_seq))) return -23;
    NllArgs a{};
    a.T = T; a.B = B; a.V = V; a.S = S; a.logits = logits; a.ld = ld_logits; a.div = div; a.align = align; a.cp_seq = cp_seq;
    a.target = target; a.pad_idx = pad_idx; a.nll = nll; a.lse = lse; a.p_tgt = p_tgt;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == GTOS_BF16) hipLaunchKernelGGL(copy_nll_fwd_kernel<bf16_t>, dim3((unsigned)(T * B)), dim3(NT), 0, s, a);
    else hipLaunchKernelGGL(copy_nll_fwd_kernel<float>, dim3((unsigned)(T * B)), dim3(NT), 0, s, a);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_copy_nll_bwd(int dtype, int T, int B, int V, int S, const void* logits, int64_t ld_logits, const void* div,
                                 const float* align, const int64_t* cp_seq, const int64_t* target, int64_t pad_idx,
                                 const float* lse, const float* p_tgt, const float* d_nll, void* d_logits, void* d_div,
                                 float* d_align, void* stream) {
    if (T <= 0 || B <= 0) return 0;
    if (V <= 0 || S < 0 || ld_logits < V) return -24;
    if (!logits || !div || !target || !lse || !p_tgt || !d_nll || !d_logits || !d_div || (S > 0 && (!align || !cp_seq || !d_align)))
        return -23;
    NllArgs a{};
    a.T = T; a.B = B; a.V = V; a.S = S; a.logits = logits; a.ld = ld_logits; a.div = div; a.align = align; a.cp_seq = cp_seq;
    a.target = target; a.pad_idx = pad_idx; a.lse = const_cast<float*>(lse); a.p_tgt = const_cast<float*>(p_tgt);
    a.d_nll = d_nll; a.d_logits = d_logits; a.d_div = d_div; a.d_align = d_align;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == GTOS_BF16) hipLaunchKernelGGL(copy_nll_bwd_kernel<bf16_t>, dim3((unsigned)(T * B)), dim3(NT), 0, s, a);
    else hipLaunchKernelGGL(copy_nll_bwd_kernel<float>, dim3((unsigned)(T * B)), dim3(NT), 0, s, a);
    GTOS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gtos_copy_ll_fwd(int dtype, int T, int B, int V, int S, int tot_ext, const void* logits, int64_t ld_logits,
                                const void* div, const float* align, const int64_t* cp_seq, float* ll, void* stream) {
    if (T <= 0 || B <= 0) return 0;
    if (V <= 0 || S < 0 || ld_logits < V || tot_ext < V) return -24;
    if (!logits || !div || !ll || (S > 0 && (!align || !cp_seq))) return -23;
    LlArgs a{};
    a.T = T; a.B = B; a.V = V; a.S = S; a.tot = tot_ext; a.logits = logits; a.ld = ld_logits; a.div = div; a.align = align;
    a.cp_seq = cp_seq; a.ll = ll;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == GTOS_BF16) hipLaunchKernelGGL(copy_ll_kernel<bf16_t>, dim3((unsigned)(T * B)), dim3(NT), 0, s, a);
    else hipLaunchKernelGGL(copy_ll_kernel<float>, dim3((unsigned)(T * B)), dim3(NT), 0, s, a);
    GTOS_CHECK_LAUNCH();
    return 0;
}
