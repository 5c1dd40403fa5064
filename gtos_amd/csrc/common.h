// Shared device helpers for the gtos MI355X (gfx950) kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GTOS_F32 0
#define GTOS_BF16 1

typedef uint16_t bf16_t;   // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

typedef __attribute__((ext_vector_type(4))) uint32_t U128;   // 16 bytes as a first-class SSA value (never address-taken)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round to nearest even, in hardware (v_cvt_pk_bf16_f32: one instruction per pair; the shift/add/mask
// sequence it replaces was ~10 VALU per pair and, with 64 outputs per lane, a third of the GEMM epilogue's issue slots)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_hw_t;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, static_cast<__bf16>(f)); }
__device__ __forceinline__ float lo_bf(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf(float a, float b) {
    const f32x2_hw_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw_t));
}

// 8 consecutive elements <-> 8 floats (one lane's slice of a row)
template <typename T> struct Vec8;
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct Vec8<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        const uint4 a = *reinterpret_cast<const uint4*>(p);
        v[0] = lo_bf(a.x); v[1] = hi_bf(a.x); v[2] = lo_bf(a.y); v[3] = hi_bf(a.y);
        v[4] = lo_bf(a.z); v[5] = hi_bf(a.z); v[6] = lo_bf(a.w); v[7] = hi_bf(a.w);
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf(v[0], v[1]), pack_bf(v[2], v[3]),
                                                  pack_bf(v[4], v[5]), pack_bf(v[6], v[7]));
    }
};

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }

// Counter-based dropout RNG: keep(seed, idx) is a pure function, so backward regenerates the mask.
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ float rand01(uint64_t seed, uint64_t idx) {
    uint32_t h = mix32((uint32_t)idx ^ (uint32_t)seed);
    h = mix32(h + (uint32_t)(idx >> 32) * 0x9E3779B9U + (uint32_t)(seed >> 32));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t idx, float p) { return rand01(seed, idx) >= p; }

// A hipGraph replay freezes every kernel argument, the dropout seeds included.  When a training step is replayed from a graph the host
// registers one device word (gtos_set_seed_epoch) that the step bumps once per replay, and every dropout kernel folds it into its
// seed on entry.  Not registered (the default): seeds are used as passed.  One copy of the pointer per translation unit (the library
// is built without relocatable device code); gtos_set_seed_epoch sets them together.
static __device__ const unsigned long long* g_seed_epoch = nullptr;
__device__ __forceinline__ uint64_t live_seed(uint64_t seed) {
    const unsigned long long* p = g_seed_epoch;
    return p ? seed + (uint64_t)(*p) * 0x9E3779B97F4A7C15ull : seed;
}
#define GTOS_SEED_EPOCH_SETTER(tu)                                                                                   \
    extern "C" __attribute__((visibility("hidden"))) int gtosi_##tu##_set_seed_epoch(const void* p) {               \
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_seed_epoch), &p, sizeof(p));                                      \
    }
extern "C" __attribute__((visibility("hidden"))) int gtosi_gemm_set_seed_epoch(const void* p);
extern "C" __attribute__((visibility("hidden"))) int gtosi_rel_attn_set_seed_epoch(const void* p);
extern "C" __attribute__((visibility("hidden"))) int gtosi_rowops_set_seed_epoch(const void* p);
extern "C" __attribute__((visibility("hidden"))) int gtosi_gru_step_set_seed_epoch(const void* p);
extern "C" __attribute__((visibility("hidden"))) int gtosi_tokenenc_set_seed_epoch(const void* p);

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

#define GTOS_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
