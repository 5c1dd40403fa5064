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
// Tile 128x128xBK per 256-thread workgroup (4 waves, 2x2, 64x64 per wave = 4x4 MFMA 16x16 tiles).
// bf16 inputs use v_mfma_f32_16x16x32_bf16, fp32 inputs the exact-fp32 v_mfma_f32_16x16x4_f32.
// Both operands are staged K-contiguous in LDS (M/N-contiguous operands are transposed on the
// LDS write) so fragments are 16-byte ds_reads; the next tile's global loads are issued before the
// MFMAs of the current one.  MFMA operands are swapped (D = B.A^T) so each lane ends up with 4
// consecutive output columns and stores 8/16-byte vectors.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, NT = 256;

template <typename T> struct GemmCfg;
template <> struct GemmCfg<bf16_t> { static constexpr int BK = 64, VEC = 8, PAD = 8; };
template <> struct GemmCfg<float>  { static constexpr int BK = 32, VEC = 4, PAD = 4; };

struct GemmArgs {
    const void* A; const void* B; void* C; const float* bias;
    int M, N, K; int64_t lda, ldb, ldc;
    int vecA, vecB, vecC;     // 16-byte vector access allowed (alignment checked on the host)
    int relu, accumulate, splitk;
    float p_drop; uint64_t seed;
};

// ---- global -> register tile load.  KC=true: operand rows are K-contiguous ([rows, K], ld).
//      KC=false: operand is stored [K, rows] (rows contiguous) and is transposed on the LDS write.
template <typename T, bool KC>
__device__ __forceinline__ void load_tile(const T* __restrict__ base, int64_t ld, int rows_total, int K,
                                          int row0, int k0, int kend, int vec_ok, U128 (&regs)[(BM * GemmCfg<T>::BK / GemmCfg<T>::VEC) / NT]) {
    constexpr int BK = GemmCfg<T>::BK, VEC = GemmCfg<T>::VEC;
    constexpr int ITERS = (BM * BK / VEC) / NT;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int v = it * NT + threadIdx.x;
        int r, k;
        if (KC) { r = row0 + v / (BK / VEC); k = k0 + (v % (BK / VEC)) * VEC; }
        else    { k = k0 + v / (BM / VEC);   r = row0 + (v % (BM / VEC)) * VEC; }
        U128 val = {0u, 0u, 0u, 0u};
        T* e = reinterpret_cast<T*>(&val);
        if (KC) {
            if (r < rows_total && k < kend) {
                const T* p = base + (int64_t)r * ld + k;
                if (vec_ok && k + VEC <= kend) val = *reinterpret_cast<const U128*>(p);
                else {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) if (k + i < kend) e[i] = p[i];
                }
            }
        } else {
            if (k < kend && r < rows_total) {
                const T* p = base + (int64_t)k * ld + r;
                if (vec_ok && r + VEC <= rows_total) val = *reinterpret_cast<const U128*>(p);
                else {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) if (r + i < rows_total) e[i] = p[i];
                }
            }
        }
        regs[it] = val;
    }
}

template <typename T, bool KC>
__device__ __forceinline__ void store_tile(T* __restrict__ lds, const U128 (&regs)[(BM * GemmCfg<T>::BK / GemmCfg<T>::VEC) / NT]) {
    constexpr int BK = GemmCfg<T>::BK, VEC = GemmCfg<T>::VEC, LDK = BK + GemmCfg<T>::PAD;
    constexpr int ITERS = (BM * BK / VEC) / NT;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int v = it * NT + threadIdx.x;
        if (KC) {
            const int r = v / (BK / VEC), k = (v % (BK / VEC)) * VEC;
            *reinterpret_cast<U128*>(lds + r * LDK + k) = regs[it];
        } else {
            const int k = v / (BM / VEC), r = (v % (BM / VEC)) * VEC;
            const T* e = reinterpret_cast<const T*>(&regs[it]);
#pragma unroll
            for (int i = 0; i < VEC; ++i) lds[(r + i) * LDK + k] = e[i];
        }
    }
}

template <typename T, typename TO, bool TA, bool TB>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmArgs a) {
    constexpr int BK = GemmCfg<T>::BK, VEC = GemmCfg<T>::VEC, LDK = BK + GemmCfg<T>::PAD;
    constexpr int ITERS = (BM * BK / VEC) / NT;
    __shared__ __attribute__((aligned(16))) T lds[(BM + BN) * LDK];
    T* As = lds;
    T* Bs = lds + BM * LDK;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    // split-K range (whole BK tiles per split)
    const int ktiles = (a.K + BK - 1) / BK;
    const int tps = (ktiles + a.splitk - 1) / a.splitk;
    const int kbeg = blockIdx.z * tps * BK;
    const int kend = min(a.K, kbeg + tps * BK);
    if (kbeg >= kend && a.splitk > 1) return;

    const T* A = static_cast<const T*>(a.A);
    const T* B = static_cast<const T*>(a.B);

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    U128 ra[ITERS], rb[ITERS];
    load_tile<T, !TA>(A, a.lda, a.M, a.K, m0, kbeg, kend, a.vecA, ra);
    load_tile<T, TB>(B, a.ldb, a.N, a.K, n0, kbeg, kend, a.vecB, rb);

    const int fr = lane & 15, fq = lane >> 4;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();                                   // previous tile's fragment reads are done
        store_tile<T, !TA>(As, ra);
        store_tile<T, TB>(Bs, rb);
        __syncthreads();
        if (k0 + BK < kend) {                              // prefetch next tile under the MFMAs
            load_tile<T, !TA>(A, a.lda, a.M, a.K, m0, k0 + BK, kend, a.vecA, ra);
            load_tile<T, TB>(B, a.ldb, a.N, a.K, n0, k0 + BK, kend, a.vecB, rb);
        }
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < BK / 32; ++ks) {
                bf16x8_t fa[4], fb[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    fa[t] = *reinterpret_cast<const bf16x8_t*>(As + (wm + t * 16 + fr) * LDK + ks * 32 + fq * 8);
                    fb[t] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn + t * 16 + fr) * LDK + ks * 32 + fq * 8);
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)   // swapped operands: rows of D <-> n, cols <-> m
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < BK / 4; ++ks) {
                float fa[4], fb[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    fa[t] = As[(wm + t * 16 + fr) * LDK + ks * 4 + fq];
                    fb[t] = Bs[(wn + t * 16 + fr) * LDK + ks * 4 + fq];
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: lane holds C[m = .. + fr][n = .. + fq*4 + 0..3]
    TO* C = static_cast<TO*>(a.C);
    const float keep_scale = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm + mt * 16 + fr;
        if (m >= a.M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn + nt * 16 + fq * 4;
            if (n >= a.N) continue;
            float v[4] = {acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]};
            TO* cp = C + (int64_t)m * a.ldc + n;
            if (a.splitk > 1) {
                if constexpr (sizeof(TO) == 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (n + i < a.N) atomicAdd(reinterpret_cast<float*>(cp) + i, v[i]);
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
            if (a.vecC && n + 3 < a.N) {
                if constexpr (sizeof(TO) == 4) {
                    float4 o = make_float4(v[0], v[1], v[2], v[3]);
                    if (a.accumulate) { const float4 c = *reinterpret_cast<const float4*>(cp); o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
                    *reinterpret_cast<float4*>(cp) = o;
                } else {
                    if (a.accumulate) {
                        const uint2 c = *reinterpret_cast<const uint2*>(cp);
                        v[0] += lo_bf(c.x); v[1] += hi_bf(c.x); v[2] += lo_bf(c.y); v[3] += hi_bf(c.y);
                    }
                    *reinterpret_cast<uint2*>(cp) = make_uint2(pack_bf(v[0], v[1]), pack_bf(v[2], v[3]));
                }
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

template <typename T, typename TO>
int launch(const GemmArgs& a, int transA, int transB, hipStream_t s) {
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, a.splitk), block(NT);
    if (!transA && transB)       hipLaunchKernelGGL((gemm_kernel<T, TO, false, true>), grid, block, 0, s, a);
    else if (!transA && !transB) hipLaunchKernelGGL((gemm_kernel<T, TO, false, false>), grid, block, 0, s, a);
    else if (transA && !transB)  hipLaunchKernelGGL((gemm_kernel<T, TO, true, false>), grid, block, 0, s, a);
    else return -2;
    GTOS_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int gtos_gemm(int in_dtype, int out_dtype, int transA, int transB, int M, int N, int K,
                         const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                         const float* bias, int relu, float p_drop, uint64_t seed, int accumulate,
                         int splitk, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0) return -3;
    const int es = in_dtype == GTOS_BF16 ? 2 : 4, vec = 16 / es;
    const int eo = out_dtype == GTOS_BF16 ? 2 : 4;
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.vecA = ((uintptr_t)A % 16 == 0) && (lda % vec == 0);
    a.vecB = ((uintptr_t)B % 16 == 0) && (ldb % vec == 0);
    a.vecC = ((uintptr_t)C % (4 * eo) == 0) && (ldc % 4 == 0);
    a.relu = relu; a.accumulate = accumulate; a.p_drop = p_drop; a.seed = seed;
    if (splitk < 1) splitk = 1;
    const int BKc = in_dtype == GTOS_BF16 ? GemmCfg<bf16_t>::BK : GemmCfg<float>::BK;
    const int ktiles = (K + BKc - 1) / BKc;
    if (splitk > ktiles) splitk = ktiles;
    a.splitk = splitk;
    if (splitk > 1 && (out_dtype != GTOS_F32 || bias || relu || p_drop > 0.f || !accumulate)) return -4;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (in_dtype == GTOS_BF16 && out_dtype == GTOS_BF16) return launch<bf16_t, bf16_t>(a, transA, transB, s);
    if (in_dtype == GTOS_BF16 && out_dtype == GTOS_F32)  return launch<bf16_t, float>(a, transA, transB, s);
    if (in_dtype == GTOS_F32 && out_dtype == GTOS_F32)   return launch<float, float>(a, transA, transB, s);
    return -1;
}
