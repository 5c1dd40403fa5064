// Fused relation-aware graph attention for gfx950 (forward + backward): streaming kernels, one 4-wave workgroup per
// (query, graph) row whose waves split the keys.
//
// Replaces the reference op sequence of RelationMultiheadAttention.forward
// (/root/reference/generator/graph_transformer.py:122-165: q.unsqueeze(1)+ra, k.unsqueeze(0)+rb, scaling,
// einsum scores, two masked_fill_, softmax over keys, weight dropout, einsum with v) and, with mode 0,
// MultiheadAttention.forward (/root/reference/generator/transformer.py:120-162).  Nothing of size
// [n,n,B*H,hd] is materialised:
//
//   s[i,j,b,h] = scale * sum_e (q[i,b,h,e] + ra[j,i,b,h,e]) * (k[j,b,h,e] + rb[j,i,b,h,e])
//   p = softmax_j(mask(s)),  o[i,b,h,:] = sum_j drop(p)[i,j,b,h] * v[j,b,h,:]
//
// relation operand (mode):
//   0  none            plain multi-head attention (decoder self / cross attention)
//   1  dense           rarb[S,T,B,2d]  = relation_in_proj(relation), memory order [key j][query i]
//   2  factored        bank[R,2d] = relation_in_proj(relation_encoder output) + int32 type ids,
//                      idx_q[T,B,S] (query-major) / idx_k[S,B,T] (key-major); the per-pair row is gathered
//                      inside the kernel, which fuses away the reference's index_select
//                      (/root/reference/generator/generator.py:79).
//
// Mapping: one 256-thread workgroup owns one (query i, graph b); its 4 waves take the keys interleaved (wave w, key
// group g: j = jb + w*G + g) and merge their online-softmax states (running max / sum / output) through LDS at the
// end.  Inside a wave a row of d channels is spread over LR = d/8 lanes, 8 channels (16 B bf16 / 32 B fp32) per lane,
// so one wave-instruction streams a whole 2d-wide relation row with fully coalesced 16-byte loads; the per-head dot
// product is a DPP reduction (v_add with quad_perm / row_mirror modifiers, no LDS round trip) over the LH = hd/8
// lanes of a head.  If d < 512 a wave processes G = 64/LR keys at once.  The key loop is double buffered: two named
// register sets of U keys, the loads of one in flight while the other is consumed.  Masks are staged in LDS once per
// workgroup.  K/V rows come through L2: the block->(i,b) map keeps every graph's K/V on one XCD (8 XCDs, private L2s).
//
// Shapes.  The fast path above needs d and hd = d/H to be powers of two, hd >= 8, d <= 512 (every shipped configuration).
// Everything else the reference accepts with hd % 8 == 0, hd <= 512 (graph_transformer.py:75 only asks d % H == 0) runs on
// the GENERIC lane map: a head takes LH = pow2ceil(hd/8) lanes (the lanes past hd/8 idle), a key row takes nhs heads =
// LR = nhs*LH <= 64 lanes, and the heads are processed in ceil(H/nhs) independent SLICES (blockIdx.y) -- attention heads
// never mix, so a slice is the same kernel on its own channels.  Same arithmetic, same dropout counters.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
    uint4 a;
    __device__ __forceinline__ void load(const bf16_t* p) { a = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void zero() { a = make_uint4(0, 0, 0, 0); }
    __device__ __forceinline__ void get(float (&v)[8]) const {
        v[0] = lo_bf(a.x); v[1] = hi_bf(a.x); v[2] = lo_bf(a.y); v[3] = hi_bf(a.y);
        v[4] = lo_bf(a.z); v[5] = hi_bf(a.z); v[6] = lo_bf(a.w); v[7] = hi_bf(a.w);
    }
};
template <> struct Raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4);
    }
    __device__ __forceinline__ void zero() { a = make_float4(0, 0, 0, 0); b = a; }
    __device__ __forceinline__ void get(float (&v)[8]) const {
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
};

// keys per register set; the streaming loops keep TWO sets in flight (software pipeline).  One key per set: 90 VGPRs in
// the forward kernel (5 waves per SIMD), 106 in the query-major backward (4); two keys per set measured 4-9 % slower on
// the same box (fewer resident waves outweigh the deeper per-wave pipeline)
#ifndef GTOS_ATTN_U
#define GTOS_ATTN_U 1
#endif
template <typename T> struct Unroll { static constexpr int U = GTOS_ATTN_U; };
template <> struct Unroll<float> { static constexpr int U = 1; };

struct AttnArgs {
    const void *q, *k, *v; int64_t ldq, ldk, ldv;          // rows (t*B+b)*ld, element units
    const void* rel;            // mode 1: rarb [S,T,B,2d]; mode 2: bank [R,2d]
    const int* idx_q;           // mode 2: [T,B,S]
    const int* idx_k;           // mode 2: [S,B,T]   (backward, key-major pass)
    const uint8_t* key_pad;     // [S,B] or null
    const uint8_t* attn_mask;   // [T,S] or null
    void* o; int64_t ldo;       // fwd out / bwd in [T,B,d]
    float* lse;                 // [T,B,H]
    float* w;                   // optional [T,S,B,H] post-dropout weights (fwd out; bwd in when dw given)
    // backward
    const void* d_o; int64_t lddo;
    const float* dw;            // optional upstream grad on w, [T,S,B,H]
    void *dq, *dk, *dv; int64_t lddq, lddk, lddv;
    int64_t ld_drel;            // mode 2: row stride of d_rel = d(bank) for the singleton types' direct rows
    void* d_rel;                // mode 1: d_rarb [S,T,B,2d] (type T)
    float* pd;                  // scratch [T,S,B,H]: post-dropout probabilities
    float* gs;                  // scratch [T,S,B,H]: scale * dS
    int T, S, B, H, d, mode;
    int hd, nhs, lr;            // generic lane map: head width, heads per slice, lanes per key row (= nhs * LH)
    float scale, p_drop; uint64_t seed;
};

// lane -> (key group g, lane in row cl, first channel c, head h); act: the lane owns 8 real channels; lead: first lane of a head
struct LaneMap { int LR, g, cl, c, h; bool act, lead; };
template <int LH, bool GEN>
__device__ __forceinline__ LaneMap lane_map(int d, int H, int hd, int nhs, int lr, int lane) {
    LaneMap m;
    if constexpr (!GEN) {
        m.LR = d >> 3; m.g = lane / m.LR; m.cl = lane % m.LR; m.c = m.cl * 8; m.h = m.cl / LH;
        m.act = true; m.lead = (m.cl % LH) == 0;
    } else {
        m.LR = lr; m.g = lane / lr; m.cl = lane % lr;
        const int hl = m.cl / LH, p = m.cl % LH;
        m.h = (int)blockIdx.y * nhs + hl;
        m.c = m.h * hd + p * 8;
        m.act = m.h < H && p * 8 < hd;
        m.lead = m.act && p == 0;
    }
    return m;
}

// Sum over the LH lanes of a head.  DPP lane permutes (VALU speed) for the steps inside a 16-lane row: hipcc lowers
// __shfl_xor to ds_bpermute (an LDS-pipe round trip of ~100 cycles) even for constant offsets, and three of those per key
// sat on the critical path of the streaming loop.
#define GTOS_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
template <int LH> __device__ __forceinline__ float head_sum(float v) {
    if (LH >= 2) v += GTOS_DPP(v, 0xB1);       // quad_perm [1,0,3,2]  (xor 1)
    if (LH >= 4) v += GTOS_DPP(v, 0x4E);       // quad_perm [2,3,0,1]  (xor 2)
    if (LH >= 8) v += GTOS_DPP(v, 0x141);      // row_half_mirror: the other quad of the 8-lane half row
    if (LH >= 16) v += GTOS_DPP(v, 0x140);     // row_mirror: the other half of the 16-lane row
    if (LH >= 32) v += __shfl_xor(v, 16);
    if (LH >= 64) v += __shfl_xor(v, 32);
    return v;
}

// block -> (row index, graph).  One 4-wave workgroup owns one (row, graph); its waves split the other axis (keys in the
// forward / query-major kernels, queries in the key-major one) 4 ways, interleaved, and merge through LDS.  Work items
// of 1/4 the size quadruple the number of workgroups (6464 at C2 instead of 1616 for ~768 resident), which removes the
// 30 % tail loss of "2.1 rounds rounded up to 3".  Blocks are dealt to XCDs round-robin by the dispatcher (id % 8), so
// each XCD gets a fixed subset of graphs and their K/V rows stay in its private L2.
__device__ __forceinline__ bool map_block(int rows, int B, int& row, int& b) {
    const int blk = blockIdx.x;
    if ((B & 7) == 0) { const int x = blk & 7, w = blk >> 3, gpx = B >> 3; b = x * gpx + (w % gpx); row = w / gpx; }
    else { b = blk % B; row = blk / B; }
    return row < rows;
}

__device__ __forceinline__ bool is_masked(const AttnArgs& a, int i, int j, int b) {
    bool m = false;
    if (a.key_pad) m = a.key_pad[(int64_t)j * a.B + b] != 0;
    if (a.attn_mask) m = m || (a.attn_mask[(int64_t)i * a.S + j] != 0);
    return m;
}

// Combined key-padding / attention mask of this block's (i,b) row staged once in LDS (one byte per key): the per-key
// global byte loads in the streaming loop cost a VMEM issue each and showed up as ~25 % of the kernel time.
constexpr int MAXS_LDS = 1024;
__device__ __forceinline__ const unsigned char* stage_mask(const AttnArgs& a, int i, int b, unsigned char* sm) {
    const bool use = (a.key_pad || a.attn_mask) && a.S <= MAXS_LDS;
    if (use) for (int j = threadIdx.x; j < a.S; j += 256) sm[j] = is_masked(a, i, j, b) ? 1 : 0;
    __syncthreads();
    return use ? sm : nullptr;
}
// Factored operand: the type ids of a workgroup's row (S of them, 404 bytes at C2) staged in LDS by one coalesced read (round 5).  A bank
// row's address needs its id first; read from global memory, that id load shared the wave's in-order vmcnt with the row loads of the
// set in flight, so waiting for the ids of the NEXT set meant waiting for the rows of THIS one: one set of rows in flight instead of two,
// and the factored forward took the dense kernel's time on two thirds of its bytes (0.46 of HBM peak on its algorithmic
// bytes against 0.68).  From LDS the ids arrive on lgkmcnt, independent of the row loads.  No barrier here: the callers' next barrier
// (stage_mask's, or their own) publishes the table.
__device__ __forceinline__ const int* stage_ids(const int* __restrict__ src, int n, int* sid) {
    if (!src || n > MAXS_LDS) return nullptr;
    for (int j = threadIdx.x; j < n; j += 256) sid[j] = src[j];
    return sid;
}
__device__ __forceinline__ bool key_dead(const AttnArgs& a, const unsigned char* sm, int i, int j, int b) {
    if (sm) return sm[j] != 0;
    return (a.key_pad || a.attn_mask) ? is_masked(a, i, j, b) : false;
}

// ------------------------------------------------------------------------------------------- forward
template <typename T, int LH, bool GEN>
__global__ __launch_bounds__(256, 4) void rel_attn_fwd_kernel(AttnArgs a) {
    if (a.p_drop > 0.f) a.seed = live_seed(a.seed);
    constexpr int U = Unroll<T>::U;
    __shared__ unsigned char smask[MAXS_LDS];
    __shared__ int sid[MAXS_LDS];                // factored operand: the type ids of this (query, graph) row (see stage_ids)
    __shared__ float red[4][64][10];             // per wave, per lane: m, l, o[8]
    __shared__ float fin[64][2];                 // merged m and 1/l per lane (for the weights pass)
    int i, b;
    if (!map_block(a.T, a.B, i, b)) return;      // whole block
    const int* sidp = stage_ids(a.mode == 2 ? a.idx_q + ((int64_t)i * a.B + b) * a.S : nullptr, a.S, sid);
    const unsigned char* sm = stage_mask(a, i, b, smask);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const LaneMap lm = lane_map<LH, GEN>(a.d, a.H, a.hd, a.nhs, a.lr, lane);
    const int d = a.d, LR = lm.LR, G = 64 / LR, g = lm.g, c = lm.c, h = lm.h;
    const bool act = lm.act, lead = lm.lead;
    const int KS = 4 * G;                        // keys taken per unroll slot by the whole block
    const T* qp = static_cast<const T*>(a.q) + ((int64_t)i * a.B + b) * a.ldq + c;
    const T* kb = static_cast<const T*>(a.k) + (int64_t)b * a.ldk + c;
    const T* vb = static_cast<const T*>(a.v) + (int64_t)b * a.ldv + c;
    const T* rel = static_cast<const T*>(a.rel);
    const int* iq = a.mode == 2 ? (sidp ? sidp : a.idx_q + ((int64_t)i * a.B + b) * a.S) : nullptr;
    const float keep_scale = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
    const int joff = wv * G + g;                 // this lane group's key inside a slot

    float qf[8];
    { Raw8<T> r; r.zero(); if (act) r.load(qp); r.get(qf); }
    float m = -INFINITY, l = 0.f, o[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    // Software pipeline over two register sets: the loads of the NEXT U keys are issued before the current U keys are
    // reduced, so every wave always has a set of loads in flight (the un-pipelined loop had none while it computed).
    struct KeySet { Raw8<T> ra[U], rb[U], k[U], v[U]; int tn[U]; };
    auto load_idx = [&](int jb, KeySet& ks) {
#pragma unroll
        for (int u = 0; u < U; ++u) { const int j = jb + u * KS + joff; ks.tn[u] = (iq && j < a.S) ? iq[j] : 0; }   // bit 31: singleton type
    };
    auto issue = [&](int jb, KeySet& ks) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb + u * KS + joff;
            ks.ra[u].zero(); ks.rb[u].zero(); ks.k[u].zero(); ks.v[u].zero();
            if (j < a.S && act) {
                ks.k[u].load(kb + (int64_t)j * a.B * a.ldk);
                ks.v[u].load(vb + (int64_t)j * a.B * a.ldv);
                if (a.mode == 1) {
                    const T* p = rel + (((int64_t)j * a.T + i) * a.B + b) * (2 * d) + c;
                    ks.ra[u].load(p); ks.rb[u].load(p + d);
                } else if (a.mode == 2) {
                    const T* p = rel + (int64_t)(ks.tn[u] & 0x7fffffff) * (2 * d) + c;
                    ks.ra[u].load(p); ks.rb[u].load(p + d);
                }
            }
        }
    };
    auto consume = [&](int jb, const KeySet& ks) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb + u * KS + joff;
            float ra[8], rb[8], kf[8], vf[8];
            ks.ra[u].get(ra); ks.rb[u].get(rb); ks.k[u].get(kf); ks.v[u].get(vf);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(qf[e] + ra[e], kf[e] + rb[e], s);
            s = head_sum<LH>(s) * a.scale;
            const bool dead = (j >= a.S) || key_dead(a, sm, i, j, b);
            if (dead) s = -INFINITY;
            if (a.w && j < a.S && lead) a.w[(((int64_t)i * a.S + j) * a.B + b) * a.H + h] = s;
            const float mn = fmaxf(m, s);
            float alpha = 1.f, pe = 0.f;
            if (mn != -INFINITY) { alpha = __expf(m - mn); pe = __expf(s - mn); }
            l = l * alpha + pe;
            float pv = pe;
            if (a.p_drop > 0.f && !dead)
                pv = drop_keep(a.seed, (((uint64_t)i * a.S + j) * a.B + b) * a.H + h, a.p_drop) ? pe * keep_scale : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(o[e], alpha, pv * vf[e]);
            m = mn;
        }
    };
    const int STEP = KS * U;
    KeySet A, Bs;
    load_idx(0, A); load_idx(STEP, Bs);
    issue(0, A);
    for (int jb = 0; jb < a.S; jb += 2 * STEP) {
        issue(jb + STEP, Bs); load_idx(jb + 2 * STEP, A);
        consume(jb, A);
        if (jb + STEP >= a.S) break;
        issue(jb + 2 * STEP, A); load_idx(jb + 3 * STEP, Bs);
        consume(jb + STEP, Bs);
    }
    // merge the G key-groups of this wave (lanes with equal cl, different g)
    for (int off = LR; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(m, off), l2 = __shfl_xor(l, off);
        const float mn = fmaxf(m, m2);
        float a1 = 1.f, a2 = 1.f;
        if (mn != -INFINITY) { a1 = __expf(m - mn); a2 = __expf(m2 - mn); }
        l = l * a1 + l2 * a2;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float o2 = __shfl_xor(o[e], off); o[e] = o[e] * a1 + o2 * a2; }
        m = mn;
    }
    // merge the 4 waves through LDS
    red[wv][lane][0] = m; red[wv][lane][1] = l;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wv][lane][2 + e] = o[e];
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) {
            const float m2 = red[w2][lane][0], l2 = red[w2][lane][1];
            const float mn = fmaxf(m, m2);
            float a1 = 1.f, a2 = 1.f;
            if (mn != -INFINITY) { a1 = __expf(m - mn); a2 = __expf(m2 - mn); }
            l = l * a1 + l2 * a2;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = o[e] * a1 + red[w2][lane][2 + e] * a2;
            m = mn;
        }
        const float inv = l > 0.f ? 1.f / l : 0.f;
        fin[lane][0] = m; fin[lane][1] = inv;
        if (g == 0 && act) {
            float r[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = o[e] * inv;
            Vec8<T>::store(static_cast<T*>(a.o) + ((int64_t)i * a.B + b) * a.ldo + c, r);
            if (lead) a.lse[((int64_t)i * a.B + b) * a.H + h] = (l > 0.f) ? m + __logf(l) : -INFINITY;
        }
    }
    if (a.w) {                                   // normalise the raw scores this same lane wrote above
        __syncthreads();
        if (lead) {
            const float mf = fin[lane][0], inv = fin[lane][1];
            for (int j = joff; j < a.S; j += KS) {
                const int64_t off = (((int64_t)i * a.S + j) * a.B + b) * a.H + h;
                const float s = a.w[off];
                float p = (s == -INFINITY || inv <= 0.f) ? 0.f : __expf(s - mf) * inv;
                if (a.p_drop > 0.f && p > 0.f) p = drop_keep(a.seed, (uint64_t)off, a.p_drop) ? p * keep_scale : 0.f;
                a.w[off] = p;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------- backward, query-major
// per (i,b): recompute p from lse; dS; dq_i = sum_j scale*dS*(k_j+rb); dense mode writes d_rarb rows;
// stores pd (post-dropout p) and gs (= scale*dS) for the key-major and bank passes.
template <typename T, int LH, bool GEN>
__global__ __launch_bounds__(256) void rel_attn_bwd_q_kernel(AttnArgs a) {
    if (a.p_drop > 0.f) a.seed = live_seed(a.seed);
    constexpr int U = Unroll<T>::U;
    __shared__ unsigned char smask[MAXS_LDS];
    __shared__ int sid[MAXS_LDS];
    __shared__ float red[4][64][8];
    __shared__ float redw[4][64];
    int i, b;
    if (!map_block(a.T, a.B, i, b)) return;
    const int* sidp = stage_ids(a.mode == 2 ? a.idx_q + ((int64_t)i * a.B + b) * a.S : nullptr, a.S, sid);
    const unsigned char* sm = stage_mask(a, i, b, smask);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const LaneMap lm = lane_map<LH, GEN>(a.d, a.H, a.hd, a.nhs, a.lr, lane);
    const int d = a.d, LR = lm.LR, G = 64 / LR, g = lm.g, c = lm.c, h = lm.h;
    const bool act = lm.act, lead = lm.lead;
    const int KS = 4 * G, joff = wv * G + g;
    const int64_t row = (int64_t)i * a.B + b;
    const T* kb = static_cast<const T*>(a.k) + (int64_t)b * a.ldk + c;
    const T* vb = static_cast<const T*>(a.v) + (int64_t)b * a.ldv + c;
    const T* rel = static_cast<const T*>(a.rel);
    const int* iq = a.mode == 2 ? (sidp ? sidp : a.idx_q + row * a.S) : nullptr;
    const float keep_scale = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;

    float qf[8], dof[8], of[8];
    { Raw8<T> r; r.zero(); if (act) r.load(static_cast<const T*>(a.q) + row * a.ldq + c); r.get(qf); }
    { Raw8<T> r; r.zero(); if (act) r.load(static_cast<const T*>(a.d_o) + row * a.lddo + c); r.get(dof); }
    { Raw8<T> r; r.zero(); if (act) r.load(static_cast<const T*>(a.o) + row * a.ldo + c); r.get(of); }
    float D = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) D = fmaf(dof[e], of[e], D);
    D = head_sum<LH>(D);
    if (a.dw) {     // upstream gradient on the returned weights: D += sum_j w_ij * dw_ij (over ALL keys: 4 waves x G groups)
        float acc = 0.f;
        for (int j = joff; j < a.S && act; j += KS) {
            const int64_t off = (((int64_t)i * a.S + j) * a.B + b) * a.H + h;
            acc = fmaf(a.w[off], a.dw[off], acc);
        }
        for (int off = LR; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
        redw[wv][lane] = acc;
        __syncthreads();
        D += redw[0][lane] + redw[1][lane] + redw[2][lane] + redw[3][lane];
    }
    const float lse = act ? a.lse[row * a.H + h] : -INFINITY;
    float dq[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    struct KeySet { Raw8<T> ra[U], rb[U], k[U], v[U]; int tn[U], tw[U]; };   // tn: ids prefetched for the NEXT issue; tw: ids of the rows in flight
    auto load_idx = [&](int jb, KeySet& ks) {
#pragma unroll
        for (int u = 0; u < U; ++u) { const int j = jb + u * KS + joff; ks.tn[u] = (iq && j < a.S) ? iq[j] : 0; }   // bit 31: singleton type
    };
    auto issue = [&](int jb, KeySet& ks) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb + u * KS + joff;
            ks.ra[u].zero(); ks.rb[u].zero(); ks.k[u].zero(); ks.v[u].zero();
            if (j < a.S && act) {
                ks.k[u].load(kb + (int64_t)j * a.B * a.ldk);
                ks.v[u].load(vb + (int64_t)j * a.B * a.ldv);
                if (a.mode == 1) {
                    const T* p = rel + (((int64_t)j * a.T + i) * a.B + b) * (2 * d) + c;
                    ks.ra[u].load(p); ks.rb[u].load(p + d);
                } else if (a.mode == 2) {
                    const T* p = rel + (int64_t)(ks.tn[u] & 0x7fffffff) * (2 * d) + c;
                    ks.ra[u].load(p); ks.rb[u].load(p + d);
                    ks.tw[u] = ks.tn[u];
                }
            }
        }
    };
    auto consume = [&](int jb, const KeySet& ks) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb + u * KS + joff;
            if (j >= a.S) continue;
            float ra[8], rb[8], kf[8], vf[8];
            ks.ra[u].get(ra); ks.rb[u].get(rb); ks.k[u].get(kf); ks.v[u].get(vf);
            float s = 0.f, dpv = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ra[e] += qf[e]; rb[e] += kf[e];                 // ra := q+ra, rb := k+rb
                s = fmaf(ra[e], rb[e], s);
                dpv = fmaf(dof[e], vf[e], dpv);
            }
            s = head_sum<LH>(s) * a.scale;
            dpv = head_sum<LH>(dpv);
            const int64_t off = (((int64_t)i * a.S + j) * a.B + b) * a.H + h;
            const bool dead = key_dead(a, sm, i, j, b) || lse == -INFINITY;
            const float p = dead ? 0.f : __expf(s - lse);
            float keep = 1.f;
            if (a.p_drop > 0.f) keep = drop_keep(a.seed, (uint64_t)off, a.p_drop) ? keep_scale : 0.f;
            float dp = dpv;
            if (a.dw && act) dp += a.dw[off];
            const float gsc = a.scale * p * (keep * dp - D);
            if (lead) { a.pd[off] = p * keep; a.gs[off] = gsc; }
            float dra[8], drb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { dra[e] = gsc * rb[e]; drb[e] = gsc * ra[e]; dq[e] += dra[e]; }
            if (!act) {
            } else if (a.mode == 1) {
                T* p2 = static_cast<T*>(a.d_rel) + (((int64_t)j * a.T + i) * a.B + b) * (2 * d) + c;
                Vec8<T>::store(p2, dra);
                Vec8<T>::store(p2 + d, drb);
            } else if (a.mode == 2 && ks.tw[u] < 0 && a.d_rel) {
                // a type that occurs ONCE in the batch (bit 31 of its id, set by the host index): this pair's term IS the
                // type's bank gradient row -- written here, where both halves sit in registers; the type-major pass skips it
                T* p2 = static_cast<T*>(a.d_rel) + (int64_t)(ks.tw[u] & 0x7fffffff) * a.ld_drel + c;
                Vec8<T>::store(p2, dra);
                Vec8<T>::store(p2 + d, drb);
            }
        }
    };
    const int STEP = KS * U;
    KeySet A, Bs;
    load_idx(0, A); load_idx(STEP, Bs);
    issue(0, A);
    for (int jb = 0; jb < a.S; jb += 2 * STEP) {
        issue(jb + STEP, Bs); load_idx(jb + 2 * STEP, A);
        consume(jb, A);
        if (jb + STEP >= a.S) break;
        issue(jb + 2 * STEP, A); load_idx(jb + 3 * STEP, Bs);
        consume(jb + STEP, Bs);
    }
    for (int off = LR; off < 64; off <<= 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) dq[e] += __shfl_xor(dq[e], off);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wv][lane][e] = dq[e];
    __syncthreads();
    if (wv == 0 && g == 0 && act) {
#pragma unroll
        for (int e = 0; e < 8; ++e) dq[e] = red[0][lane][e] + red[1][lane][e] + red[2][lane][e] + red[3][lane][e];
        Vec8<T>::store(static_cast<T*>(a.dq) + row * a.lddq + c, dq);
    }
}

// ------------------------------------------------------------------------------------------- backward, key-major
// per (j,b): dv_j = sum_i pd_ij do_i ; dk_j = sum_i gs_ij (q_i + ra_ji); the 4 waves split the queries
template <typename T, int LH, bool GEN>
__global__ __launch_bounds__(256) void rel_attn_bwd_kv_kernel(AttnArgs a) {
    constexpr int U = Unroll<T>::U;
    __shared__ float red[4][64][16];
    __shared__ int sid[MAXS_LDS];
    int j, b;
    if (!map_block(a.S, a.B, j, b)) return;
    const int* sidp = stage_ids(a.mode == 2 ? a.idx_k + ((int64_t)j * a.B + b) * a.T : nullptr, a.T, sid);
    if (sidp) __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const LaneMap lm = lane_map<LH, GEN>(a.d, a.H, a.hd, a.nhs, a.lr, lane);
    const int d = a.d, LR = lm.LR, G = 64 / LR, g = lm.g, c = lm.c, h = lm.h;
    const bool act = lm.act;
    const int KS = 4 * G, ioff = wv * G + g;
    const int64_t row = (int64_t)j * a.B + b;
    const T* qb = static_cast<const T*>(a.q) + (int64_t)b * a.ldq + c;
    const T* dob = static_cast<const T*>(a.d_o) + (int64_t)b * a.lddo + c;
    const T* rel = static_cast<const T*>(a.rel);
    const int* ik = a.mode == 2 ? (sidp ? sidp : a.idx_k + row * a.T) : nullptr;
    float dk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dv[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    for (int ib = 0; ib < a.T; ib += KS * U) {
        Raw8<T> rq[U], rdo[U], rra[U];
        float pdv[U], gsv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = ib + u * KS + ioff;
            rq[u].zero(); rdo[u].zero(); rra[u].zero(); pdv[u] = 0.f; gsv[u] = 0.f;
            if (i < a.T && act) {
                rq[u].load(qb + (int64_t)i * a.B * a.ldq);
                rdo[u].load(dob + (int64_t)i * a.B * a.lddo);
                const int64_t off = (((int64_t)i * a.S + j) * a.B + b) * a.H + h;
                pdv[u] = a.pd[off]; gsv[u] = a.gs[off];
                if (a.mode == 1) rra[u].load(rel + (((int64_t)j * a.T + i) * a.B + b) * (2 * d) + c);
                else if (a.mode == 2) rra[u].load(rel + (int64_t)(ik[i] & 0x7fffffff) * (2 * d) + c);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float qf[8], dof[8], ra[8];
            rq[u].get(qf); rdo[u].get(dof); rra[u].get(ra);
#pragma unroll
            for (int e = 0; e < 8; ++e) { dv[e] = fmaf(pdv[u], dof[e], dv[e]); dk[e] = fmaf(gsv[u], qf[e] + ra[e], dk[e]); }
        }
    }
    for (int off = LR; off < 64; off <<= 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { dk[e] += __shfl_xor(dk[e], off); dv[e] += __shfl_xor(dv[e], off); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[wv][lane][e] = dk[e]; red[wv][lane][8 + e] = dv[e]; }
    __syncthreads();
    if (wv == 0 && g == 0 && act) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dk[e] = red[0][lane][e] + red[1][lane][e] + red[2][lane][e] + red[3][lane][e];
            dv[e] = red[0][lane][8 + e] + red[1][lane][8 + e] + red[2][lane][8 + e] + red[3][lane][8 + e];
        }
        Vec8<T>::store(static_cast<T*>(a.dk) + row * a.lddk + c, dk);
        Vec8<T>::store(static_cast<T*>(a.dv) + row * a.lddv + c, dv);
    }
}

// ------------------------------------------------------------------------------------------- backward, bank (factored)
// Work item = one chunk of <= chunk_len pairs that share a relation type t (pairs sorted by type on the host):
//   d_bank[t, 0:d]  += sum_pairs gs * (k_j + RB[t]) ;  d_bank[t, d:2d] += sum_pairs gs * (q_i + RA[t])
// pair id = (j*T + i)*B + b (the memory order of relation[j][i][b]).  A type that fits one chunk stores its row of
// d_bank (type T) directly; a type spanning several chunks ("heavy": <CLS>, <rCLS>, <SELF>, <TL>, frequent short
// paths) accumulates with fp32 atomics into its slot of a small side buffer that the caller folds back.
struct BankArgs {
    const void *q, *k; int64_t ldq, ldk;
    const void* bank; const float* gs;
    const int* pair_sorted;      // [P] pair ids sorted by type
    const int* chunk_type;       // [C]
    const int* chunk_start;      // [C]
    const int* chunk_count;      // [C]
    const int* chunk_slot;       // [C] -1 -> store d_bank[t] directly; >= 0 -> atomicAdd into heavy[slot]
    const int* xcd_off;          // [9] or null: chunks [xcd_off[x], xcd_off[x+1]) belong to XCD x (their pairs' graphs live in its L2)
    void* d_bank; int64_t ld_dbank;   // [R,2d] type T, row stride ld_dbank (a column block of a wider gradient slab)
    float* heavy;                // [n_heavy,2d] fp32, zero-initialised by the caller
    int nchunks, T, S, B, H, d;
    int hd, nhs, lr;
};

// NP: pairs per lane group in flight.  Most chunks hold one or two pairs, so four in flight mostly carried empty slots in 132 registers (3
// waves per SIMD); two fit 98 (4 waves per SIMD): more chunks' dependent load chains in flight per CU (round 5; GTOS_BANK_NP=4 restores four).
template <typename T, int LH, bool GEN, int NP>
__global__ __launch_bounds__(256) void rel_attn_bwd_bank_kernel(BankArgs a) {
    const int lane = threadIdx.x & 63;
    const LaneMap lm = lane_map<LH, GEN>(a.d, a.H, a.hd, a.nhs, a.lr, lane);
    const int d = a.d, LR = lm.LR, G = 64 / LR, g = lm.g, c = lm.c, h = lm.h;
    const bool act = lm.act;
    // Most chunks hold one or two pairs, so a chunk is a chain of three dependent global loads (chunk record -> pair ids
    // and bank row -> q/k rows and gs) in front of a handful of FMAs.  Chunks are walked grid-stride and the chain is
    // software-pipelined over three consecutive chunks of a wave: while chunk i is reduced, the pair ids / bank row of
    // chunk i+1 and the record of chunk i+2 are already in flight, so an iteration costs one load latency, not three.
    struct Meta { int t, start, cnt, slot; };
    struct Lvl2 { Raw8<T> rRA, rRB; int my_pid; };
    // With xcd_off the host has grouped the chunks by the XCD that owns their (first) pair's graph -- the same
    // graph -> XCD map as the attention kernels -- so the q/k row gathers hit that XCD's private L2 instead of going out
    // to the Infinity Cache: the 20 MB of q/k rows do not fit one 4 MB L2, an eighth of them does.
    int lo = 0, hi = a.nchunks, stride = gridDim.x * 4, first = blockIdx.x * 4;
    if (a.xcd_off) {
        const int x = blockIdx.x & 7;
        lo = a.xcd_off[x]; hi = a.xcd_off[x + 1];
        stride = (gridDim.x >> 3) * 4; first = lo + (blockIdx.x >> 3) * 4;
    }
    const int last = hi - 1;
    auto load_meta = [&](int ch) {
        const int cc = ch < last ? ch : last;
        Meta m;
        m.t = a.chunk_type[cc]; m.start = a.chunk_start[cc]; m.cnt = a.chunk_count[cc]; m.slot = a.chunk_slot[cc];
        return m;
    };
    auto load_lvl2 = [&](const Meta& m) {
        Lvl2 l;
        const T* bp = static_cast<const T*>(a.bank) + (int64_t)m.t * (2 * d) + c;
        l.rRA.zero(); l.rRB.zero();
        if (act) { l.rRA.load(bp); l.rRB.load(bp + d); }
        l.my_pid = lane < min(m.cnt, 64) ? a.pair_sorted[m.start + lane] : 0;     // one coalesced load of the chunk's (first 64) pair ids
        return l;
    };
    int ch = first + (threadIdx.x >> 6);
    if (ch >= hi) return;
    Meta m1 = load_meta(ch), m2 = load_meta(ch + stride);
    Lvl2 l1 = load_lvl2(m1);
    for (; ch < hi; ch += stride) {
        const Lvl2 l2 = load_lvl2(m2);
        const Meta m3 = load_meta(ch + 2 * stride);
        const int cnt = m1.cnt;
        float da[8] = {0, 0, 0, 0, 0, 0, 0, 0}, db[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gsum = 0.f;
        // a chunk of a heavy type may hold several 64-pair blocks (the host cuts <CLS>, <TL>, ... into long chunks so that a
        // type with 65 k pairs costs ~500 fp32-atomic flushes per channel instead of ~2000); the first block's ids were
        // prefetched with the chunk record, further blocks are loaded here
        for (int sb = 0; sb < cnt; sb += 64) {
            const int nb = min(64, cnt - sb);
            const int my_pid = sb == 0 ? l1.my_pid : (lane < nb ? a.pair_sorted[m1.start + sb + lane] : 0);
            for (int p0 = 0; p0 < nb; p0 += NP * G) {           // NP pairs per group in flight
                Raw8<T> rk[NP], rq[NP];
                float gsc[NP];
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const int p = p0 + u * G + g;
                    const int pid = __shfl(my_pid, p < nb ? p : 0);
                    rk[u].zero(); rq[u].zero(); gsc[u] = 0.f;
                    if (p < nb && act) {
                        const int b = pid % a.B, ji = pid / a.B, i = ji % a.T, j = ji / a.T;
                        gsc[u] = a.gs[(((int64_t)i * a.S + j) * a.B + b) * a.H + h];
                        rk[u].load(static_cast<const T*>(a.k) + ((int64_t)j * a.B + b) * a.ldk + c);
                        rq[u].load(static_cast<const T*>(a.q) + ((int64_t)i * a.B + b) * a.ldq + c);
                    }
                }
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    float kf[8], qf[8];
                    rk[u].get(kf); rq[u].get(qf);
                    gsum += gsc[u];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { da[e] = fmaf(gsc[u], kf[e], da[e]); db[e] = fmaf(gsc[u], qf[e], db[e]); }
                }
            }
        }
        for (int off = LR; off < 64; off <<= 1) {
            gsum += __shfl_xor(gsum, off);
#pragma unroll
            for (int e = 0; e < 8; ++e) { da[e] += __shfl_xor(da[e], off); db[e] += __shfl_xor(db[e], off); }
        }
        if (g == 0 && act) {
            float RA[8], RB[8];
            l1.rRA.get(RA); l1.rRB.get(RB);
            if (m1.slot >= 0) {
                float* out = a.heavy + (int64_t)m1.slot * (2 * d) + c;
#pragma unroll
                for (int e = 0; e < 8; ++e) { atomicAdd(out + e, fmaf(gsum, RB[e], da[e])); atomicAdd(out + d + e, fmaf(gsum, RA[e], db[e])); }
            } else {
                T* out = static_cast<T*>(a.d_bank) + (int64_t)m1.t * a.ld_dbank + c;
                float r1[8], r2[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { r1[e] = fmaf(gsum, RB[e], da[e]); r2[e] = fmaf(gsum, RA[e], db[e]); }
                Vec8<T>::store(out, r1);
                Vec8<T>::store(out + d, r2);
            }
        }
        m1 = m2; l1 = l2; m2 = m3;
    }
}

template <typename K> int dispatch_lh(int lh, K&& f) {
    switch (lh) {
        case 1: return f(std::integral_constant<int, 1>());
        case 2: return f(std::integral_constant<int, 2>());
        case 4: return f(std::integral_constant<int, 4>());
        case 8: return f(std::integral_constant<int, 8>());
        case 16: return f(std::integral_constant<int, 16>());
        case 32: return f(std::integral_constant<int, 32>());
        case 64: return f(std::integral_constant<int, 64>());
    }
    return -11;
}

// Lane geometry of a shape: LH lanes per head (power of two), heads per slice, lanes per key row, slices; `generic` = off the
// power-of-two fast path.  -10: outside the boundary (H <= 0, d % H, head width not a multiple of 8 or above 512).
struct Geo { int LH, nhs, lr, slices; bool generic; };
int pow2ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }
int geometry(int d, int H, Geo& g) {
    if (H <= 0 || d <= 0 || d % H) return -10;
    const int hd = d / H;
    if (hd % 8 || hd > 512) return -10;
    g.generic = !(d <= 512 && (d & (d - 1)) == 0 && (hd & (hd - 1)) == 0);
    if (!g.generic) { g.LH = hd / 8; g.nhs = H; g.lr = d / 8; g.slices = 1; return 0; }
    g.LH = pow2ceil(hd / 8);
    int nhs = 64 / g.LH;                      // heads that fit a 64-lane row ...
    const int hp = pow2ceil(H);
    if (nhs > hp) nhs = hp;                   // ... but no more lanes than the heads need (more key groups per wave instead)
    g.nhs = nhs; g.lr = nhs * g.LH; g.slices = (H + nhs - 1) / nhs;
    return 0;
}
int check_shape(int d, int H) { Geo g; return geometry(d, H, g); }

int nblocks(int rows, int B) { return rows * B; }

}  // namespace

static int fill_args(AttnArgs& a, Geo& g, int T_, int S, int B, int H, int d, int mode, float scale, float p_drop, uint64_t seed) {
    a.T = T_; a.S = S; a.B = B; a.H = H; a.d = d; a.mode = mode; a.scale = scale; a.p_drop = p_drop; a.seed = seed;
    if (mode != 0 && T_ != S) return -12;
    const int rc = geometry(d, H, g);
    if (rc) return rc;
    a.hd = d / H; a.nhs = g.nhs; a.lr = g.lr;
    return 0;
}

extern "C" int gtos_rel_attn_fwd(int dtype, int mode, int T_, int S, int B, int H, int d,
                                 const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                 const void* rel, const int* idx_q, const uint8_t* key_pad, const uint8_t* attn_mask,
                                 float scale, float p_drop, uint64_t seed,
                                 void* o, int64_t ldo, float* lse, float* w, void* stream) {
    AttnArgs a = {};
    Geo geo;
    int rc = fill_args(a, geo, T_, S, B, H, d, mode, scale, p_drop, seed);
    if (rc) return rc;
    a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.rel = rel; a.idx_q = idx_q;
    a.key_pad = key_pad; a.attn_mask = attn_mask; a.o = o; a.ldo = ldo; a.lse = lse; a.w = w;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid(nblocks(T_, B), geo.slices);
    return dispatch_lh(geo.LH, [&](auto lh) {
        constexpr int LH = decltype(lh)::value;
        if (geo.generic) {
            if (dtype == GTOS_BF16) hipLaunchKernelGGL((rel_attn_fwd_kernel<bf16_t, LH, true>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((rel_attn_fwd_kernel<float, LH, true>), grid, dim3(256), 0, s, a);
        } else {
            if (dtype == GTOS_BF16) hipLaunchKernelGGL((rel_attn_fwd_kernel<bf16_t, LH, false>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((rel_attn_fwd_kernel<float, LH, false>), grid, dim3(256), 0, s, a);
        }
        GTOS_CHECK_LAUNCH();
        return 0;
    });
}

extern "C" int gtos_rel_attn_bwd(int dtype, int mode, int T_, int S, int B, int H, int d,
                                 const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                 const void* rel, const int* idx_q, const int* idx_k,
                                 const uint8_t* key_pad, const uint8_t* attn_mask,
                                 float scale, float p_drop, uint64_t seed,
                                 const void* o, int64_t ldo, const float* lse, const float* w,
                                 const void* d_o, int64_t lddo, const float* dw,
                                 void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                                 void* d_rel, int64_t ld_drel, float* pd, float* gs, void* stream) {
    AttnArgs a = {};
    Geo geo;
    int rc = fill_args(a, geo, T_, S, B, H, d, mode, scale, p_drop, seed);
    if (rc) return rc;
    a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.rel = rel; a.idx_q = idx_q; a.idx_k = idx_k;
    a.key_pad = key_pad; a.attn_mask = attn_mask; a.o = const_cast<void*>(o); a.ldo = ldo;
    a.lse = const_cast<float*>(lse); a.w = const_cast<float*>(w); a.d_o = d_o; a.lddo = lddo; a.dw = dw;
    a.dq = dq; a.dk = dk; a.dv = dv; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv; a.d_rel = d_rel; a.ld_drel = ld_drel; a.pd = pd; a.gs = gs;
    if (mode == 2 && d_rel && (ld_drel < 2 * d || ld_drel % 8)) return -14;
    if (dw && !w) return -13;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 gq(nblocks(T_, B), geo.slices), gk(nblocks(S, B), geo.slices);
    return dispatch_lh(geo.LH, [&](auto lh) {
        constexpr int LH = decltype(lh)::value;
#define GTOS_BWD(TT, GG) do { hipLaunchKernelGGL((rel_attn_bwd_q_kernel<TT, LH, GG>), gq, dim3(256), 0, s, a); \
                              hipLaunchKernelGGL((rel_attn_bwd_kv_kernel<TT, LH, GG>), gk, dim3(256), 0, s, a); } while (0)
        if (geo.generic) { if (dtype == GTOS_BF16) GTOS_BWD(bf16_t, true); else GTOS_BWD(float, true); }
        else { if (dtype == GTOS_BF16) GTOS_BWD(bf16_t, false); else GTOS_BWD(float, false); }
#undef GTOS_BWD
        GTOS_CHECK_LAUNCH();
        return 0;
    });
}

extern "C" int gtos_rel_attn_bwd_bank(int dtype, int n, int B, int H, int d,
                                      const void* q, int64_t ldq, const void* k, int64_t ldk,
                                      const void* bank, const float* gs,
                                      const int* pair_sorted, const int* chunk_type, const int* chunk_start,
                                      const int* chunk_count, const int* chunk_slot, const int* xcd_off, int nchunks,
                                      void* d_bank, int64_t ld_dbank, float* heavy, void* stream) {
    Geo geo;
    int rc = geometry(d, H, geo);
    if (rc) return rc;
    if (ld_dbank < 2 * d || ld_dbank % 8) return -14;
    if (nchunks <= 0) return 0;
    BankArgs a;
    a.q = q; a.k = k; a.ldq = ldq; a.ldk = ldk; a.bank = bank; a.gs = gs; a.pair_sorted = pair_sorted;
    a.chunk_type = chunk_type; a.chunk_start = chunk_start; a.chunk_count = chunk_count; a.chunk_slot = chunk_slot;
    a.d_bank = d_bank; a.ld_dbank = ld_dbank; a.heavy = heavy; a.nchunks = nchunks; a.T = n; a.S = n; a.B = B; a.H = H; a.d = d;
    a.xcd_off = xcd_off;
    a.hd = d / H; a.nhs = geo.nhs; a.lr = geo.lr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int grid = (nchunks + 3) / 4; if (grid > 4096) grid = 4096;
    if (xcd_off) grid = (grid + 7) / 8 * 8;
    const dim3 g2(grid, geo.slices);
    return dispatch_lh(geo.LH, [&](auto lh) {
        constexpr int LH = decltype(lh)::value;
        if (geo.generic) {
            if (dtype == GTOS_BF16) hipLaunchKernelGGL((rel_attn_bwd_bank_kernel<bf16_t, LH, true, 4>), g2, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((rel_attn_bwd_bank_kernel<float, LH, true, 4>), g2, dim3(256), 0, s, a);
        } else {
            if (dtype == GTOS_BF16) hipLaunchKernelGGL((rel_attn_bwd_bank_kernel<bf16_t, LH, false, 2>), g2, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((rel_attn_bwd_bank_kernel<float, LH, false, 4>), g2, dim3(256), 0, s, a);
        }
        GTOS_CHECK_LAUNCH();
        return 0;
    });
}

GTOS_SEED_EPOCH_SETTER(rel_attn)
