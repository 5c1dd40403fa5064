// Per-thread bodies of the GPU relation-index builder (csrc/relindex_dev.hip), written so that the SAME code compiles for the host:
// oracle/relindex_emul.cpp runs every stage as a serial loop (std::stable_sort / running sums where the GPU uses rocPRIM) and
// tests/test_relindex_dev.py compares the result with csrc_host/relindex.cpp array for array.  What the emulation cannot cover is the
// launch glue and the rocPRIM calls.  No atomics: counts come from binary searches in sorted keys and from scans.
//
// The index is what the attention kernels read instead of the dense relation[n,n,B,d] of generator/generator.py:79 (see
// csrc_host/relindex.cpp for the contract): the type ids in query- and key-major order with bit 31 on types that occur once, the pairs
// grouped by type in graph-major order, the chunk list of the bank-gradient kernel with heavy-type slots, and the chunks' order:
// by XCD (a chunk whose pairs live on one XCD stays there, the others go greedily to the least loaded XCD, longest first), long
// chunks first inside an XCD.  Covered: the default switches of relindex.cpp (GTOS_BANK_BALANCE on, GTOS_HEAVY_FIRST off).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define GTOS_RI_HD __host__ __device__ inline
#else
#define GTOS_RI_HD inline
#endif

namespace gtos_relindex_dev {

struct V8 { uint32_t v[8]; };                      // one counter per XCD; Add8 is the scan operator
struct Add8 {
    GTOS_RI_HD V8 operator()(const V8& a, const V8& b) const {
        V8 r;
        for (int i = 0; i < 8; ++i) r.v[i] = a.v[i] + b.v[i];
        return r;
    }
};

// sizes[] written on the device, read by the host
enum { IZ_ERR = 0, IZ_NCHUNKS = 1, IZ_NHEAVY = 2, IZ_NROAM = 3, IZ_TOTAL = 4 };

struct Geom {
    int32_t n, B, chunk, mult;
    int64_t R, P;                                  // types, cells = n * n * B
};

struct Chunks {                                    // the chunk list in type order (phase A) -- arrays of the upper bound R + P / chunk + 1
    int32_t *type, *start, *cnt, *slot, *xf, *xl, *home;
    int64_t* key0;                                 // (graph of the first pair << 20) | key row of the first pair
};

GTOS_RI_HD int32_t xcd_of(int64_t pair, int32_t B) {               // the attention kernels' graph -> XCD map
    const int64_t gb = pair % B;
    return (int32_t)((B % 8 == 0) ? gb / (B / 8) : gb % 8);
}
GTOS_RI_HD int64_t chunk_cost(int32_t cnt) { return (int64_t)(cnt + 3) / 4 + 1; }

// ---- A1: cell e of the graph-major enumeration (b, j, i): sort key = type, value = flat index of relation[j][i][b]
GTOS_RI_HD void cell_key(int64_t e, const Geom& G, const int64_t* relation, uint32_t* key, int32_t* val, int32_t* sizes) {
    const int64_t nn = (int64_t)G.n * G.n;
    const int64_t b = e / nn, j = (e / G.n) % G.n, i = e % G.n;
    const int64_t p = (j * G.n + i) * G.B + b;
    int64_t t = relation[p];
    if (t < 0 || t >= G.R) { sizes[IZ_ERR] = 1; t = 0; }
    key[e] = (uint32_t)t;
    val[e] = (int32_t)p;
}

// first position in the sorted keys whose key is > t (keys ascending)
GTOS_RI_HD int64_t upper_bound_u32(const uint32_t* skey, int64_t n, uint32_t t) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (skey[mid] <= t) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---- A2 (after the stable sort by type): pairs of type t and the running total, read off the sorted keys
GTOS_RI_HD void type_bounds(int64_t t, const Geom& G, const uint32_t* skey, int32_t* cnt, uint32_t* cum_cnt) {
    const int64_t hi = upper_bound_u32(skey, G.P, (uint32_t)t);
    const int64_t lo = t ? upper_bound_u32(skey, G.P, (uint32_t)(t - 1)) : 0;
    cnt[t] = (int32_t)(hi - lo);
    cum_cnt[t] = (uint32_t)hi;
}

// ---- A3: the ids in query-major [i,b,j] and key-major [j,b,i] order; bit 31 marks a type that occurs exactly once
GTOS_RI_HD void idx_cell(int64_t p, const Geom& G, const int64_t* relation, const int32_t* cnt, int32_t* idx_q, int32_t* idx_k) {
    const int64_t j = p / ((int64_t)G.n * G.B), i = (p / G.B) % G.n, b = p % G.B;
    int64_t t = relation[p];
    if (t < 0 || t >= G.R) t = 0;
    const int32_t v = (int32_t)t | (cnt[t] == 1 ? (int32_t)0x80000000 : 0);
    idx_q[(i * G.B + b) * G.n + j] = v;
    idx_k[(j * G.B + b) * G.n + i] = v;
}

// ---- A5: chunks of type t (none for a single-pair type, one empty chunk for a type without pairs), heavy flag
GTOS_RI_HD void type_counts(int64_t t, const Geom& G, const int32_t* cnt, uint32_t* nch_out, uint32_t* heavy_out) {
    const int64_t c = cnt[t];
    const int64_t csz = c > G.chunk ? (int64_t)G.mult * G.chunk : G.chunk;
    nch_out[t] = c == 1 ? 0u : (uint32_t)(c > 0 ? (c + csz - 1) / csz : 1);
    heavy_out[t] = c > G.chunk ? 1u : 0u;
}

// ---- A6: the chunk records of type t (cum_*: inclusive scans of the counts and of the heavy flags, running pair totals)
GTOS_RI_HD void type_chunks(int64_t t, const Geom& G, const int32_t* cnt, const uint32_t* cum_cnt, const uint32_t* nch, const uint32_t* cum_nch,
                            const uint32_t* heavy, const uint32_t* cum_heavy, const int32_t* pair_sorted, const Chunks& ch, int32_t* heavy_types) {
    const int64_t c = cnt[t];
    if (c == 1) return;
    const int64_t hi = cum_cnt[t], lo = hi - c;
    const int64_t csz = c > G.chunk ? (int64_t)G.mult * G.chunk : G.chunk;
    const int64_t base = (int64_t)cum_nch[t] - nch[t];
    int32_t slot = -1;
    if (heavy[t]) { slot = (int32_t)cum_heavy[t] - 1; heavy_types[slot] = (int32_t)t; }
    for (int64_t k = 0; k < (int64_t)nch[t]; ++k) {
        const int64_t s = lo + k * csz;
        int64_t n = hi - s;
        n = n < 0 ? 0 : (n > csz ? csz : n);
        const int64_t a = s < G.P - 1 ? s : G.P - 1;
        int64_t z = s + (n > 1 ? n : 1) - 1;
        z = z < G.P - 1 ? z : G.P - 1;
        const int64_t first = pair_sorted[a], lastp = pair_sorted[z];
        const int64_t gb = first % G.B, j = first / ((int64_t)G.n * G.B);
        const int32_t xf = xcd_of(first, G.B), xl = xcd_of(lastp, G.B);
        const int64_t q = base + k;
        ch.type[q] = (int32_t)t; ch.start[q] = (int32_t)s; ch.cnt[q] = (int32_t)n; ch.slot[q] = slot;
        ch.key0[q] = (gb << 20) | j; ch.xf[q] = xf; ch.xl[q] = xl;
        ch.home[q] = xf == xl ? xf : -1;
    }
}

// one thread
GTOS_RI_HD void sizes_a(const Geom& G, const uint32_t* cum_nch, const uint32_t* cum_heavy, int32_t* sizes) {
    sizes[IZ_NCHUNKS] = (int32_t)cum_nch[G.R - 1];
    sizes[IZ_NHEAVY] = (int32_t)cum_heavy[G.R - 1];
}

// ---- B1: sort key of the roaming chunks (no home): longest first, stable; chunks with a home sort behind them.  Also the scan input
// of the XCD loads: the cost of a chunk with a home, in the lane of its XCD.
GTOS_RI_HD void roam_key(int64_t c, const Chunks& ch, uint32_t* key, int32_t* val, V8* home_cost) {
    const int32_t home = ch.home[c];
    key[c] = home < 0 ? (uint32_t)(0x7fffffff - ch.cnt[c]) : 0xffffffffu;
    val[c] = (int32_t)c;
    V8 o;
    for (int x = 0; x < 8; ++x) o.v[x] = x == home ? (uint32_t)chunk_cost(ch.cnt[c]) : 0u;
    home_cost[c] = o;
}

// one thread, after the sort of the roam keys and the scan of home_cost: how many chunks roam, and the loads the placed chunks make
GTOS_RI_HD void roam_setup(int64_t nchunks, const uint32_t* rkey_sorted, const V8* cum_cost, unsigned long long* load, int32_t* sizes) {
    sizes[IZ_NROAM] = (int32_t)upper_bound_u32(rkey_sorted, nchunks, 0x7fffffffu);
    for (int x = 0; x < 8; ++x) load[x] = cum_cost[nchunks - 1].v[x];
}

// ---- B2: cost of the q-th roaming chunk in placement order (a compact array: the serial walk below reads it front to back)
GTOS_RI_HD void roam_cost(int64_t q, const int32_t* roam_sorted, const Chunks& ch, int32_t* rcost, const int32_t* sizes) {
    if (q < sizes[IZ_NROAM]) rcost[q] = (int32_t)chunk_cost(ch.cnt[roam_sorted[q]]);
}

// ---- B3 (ONE thread; the HIP library runs the same recurrence on one wave, csrc/relindex_dev.hip k_greedy_wave): the roaming
// chunks, longest first, each to the XCD with the fewest rounds so far (the lowest XCD among equals)
GTOS_RI_HD void greedy_homes(const int32_t* rcost, int32_t* home_q, unsigned long long* load, const int32_t* sizes) {
    const int32_t n_roam = sizes[IZ_NROAM];
    long long ld[8];
    for (int x = 0; x < 8; ++x) ld[x] = (long long)load[x];
    for (int32_t q = 0; q < n_roam; ++q) {
        int best = 0;
        for (int x = 1; x < 8; ++x) if (ld[x] < ld[best]) best = x;
        home_q[q] = best;
        ld[best] += rcost[q];
    }
    for (int x = 0; x < 8; ++x) load[x] = (unsigned long long)ld[x];
}

// ---- B3b: the homes back to the chunks
GTOS_RI_HD void scatter_homes(int64_t q, const int32_t* roam_sorted, const int32_t* home_q, const Chunks& ch, const int32_t* sizes) {
    if (q < sizes[IZ_NROAM]) ch.home[roam_sorted[q]] = home_q[q];
}

// ---- B4: final sort key of chunk c: XCD, long chunks (> 8 pairs) first and longest first, the short ones in (graph, key row) order;
// one-hot of its XCD (scan input of the per-XCD chunk counts)
GTOS_RI_HD void final_key(int64_t c, const Chunks& ch, uint64_t* key, int32_t* val, V8* home_hot) {
    const int32_t cnt = ch.cnt[c];
    const int64_t lng = cnt > 8 ? 0 : 1;
    const int64_t rank = lng ? ch.key0[c] : (int64_t)(0xfffff - (cnt < 0xfffff ? cnt : 0xfffff)) << 20;
    const int64_t home = ch.home[c];
    key[c] = (uint64_t)((home << 42) | (lng << 41) | rank);
    val[c] = (int32_t)c;
    V8 o;
    for (int x = 0; x < 8; ++x) o.v[x] = x == home ? 1u : 0u;
    home_hot[c] = o;
}

// ---- B6: the chunk list in its final order; one thread also turns the per-XCD counts into offsets
GTOS_RI_HD void gather_chunk(int64_t q, const int32_t* perm, const Chunks& ch, int32_t* chunk_type, int32_t* chunk_start, int32_t* chunk_count,
                             int32_t* chunk_slot) {
    const int32_t c = perm[q];
    chunk_type[q] = ch.type[c]; chunk_start[q] = ch.start[c]; chunk_count[q] = ch.cnt[c]; chunk_slot[q] = ch.slot[c];
}
GTOS_RI_HD void xcd_offsets(int64_t nchunks, const V8* cum_hot, int32_t* xcd_off) {
    int32_t run = 0;
    for (int x = 0; x < 8; ++x) { xcd_off[x] = run; run += (int32_t)cum_hot[nchunks - 1].v[x]; }
    xcd_off[8] = run;
}

// ---- the two-phase C ABI (gtos_relindex_dev_phase_a / _b and their emulation)
// geom[]: host integers, int64 each
enum { GE_N = 0, GE_B, GE_CHUNK, GE_MULT, GE_R, GE_P, GE_COUNT };
// tab[]: device (or host) pointers
enum { T_RELATION = 0,    // int64 [n, n, B]
       T_KEY,             // uint32 [P]   type of cell e of the graph-major enumeration
       T_VAL,             // int32 [P]
       T_SKEY,            // uint32 [P]   sorted
       T_PAIR_SORTED,     // int32 [P]   OUT
       T_IDX_Q,           // int32 [P]   OUT
       T_IDX_K,           // int32 [P]   OUT
       T_CNT,             // int32 [R]
       T_CUM_CNT,         // uint32 [R]
       T_NCH,             // uint32 [R]
       T_CUM_NCH,         // uint32 [R]
       T_HEAVY,           // uint32 [R]
       T_CUM_HEAVY,       // uint32 [R]
       T_HEAVY_TYPES,     // int32 [P / chunk + 1]   OUT (first n_heavy)
       T_C_TYPE, T_C_START, T_C_CNT, T_C_SLOT, T_C_XF, T_C_XL, T_C_HOME,   // int32 [NC = R + P / chunk + 1]
       T_C_KEY0,          // int64 [NC]
       T_LOAD,            // uint64 [8]
       T_SIZES,           // int32 [IZ_TOTAL] ZERO-FILLED by the caller
       T_RKEY,            // uint32 [nchunks]  phase B
       T_RVAL,            // int32 [nchunks]
       T_RKEY_S,          // uint32 [nchunks]
       T_ROAM_SORTED,     // int32 [nchunks]
       T_RCOST,           // int32 [nchunks]   cost of the q-th roaming chunk in placement order
       T_HOME_Q,          // int32 [nchunks]   its XCD
       T_FKEY,            // uint64 [nchunks]
       T_FVAL,            // int32 [nchunks]
       T_FKEY_S,          // uint64 [nchunks]
       T_PERM,            // int32 [nchunks]
       T_V8,              // V8 [nchunks]      scan input (home costs, then home one-hots)
       T_V8_CUM,          // V8 [nchunks]
       T_CHUNK_TYPE, T_CHUNK_START, T_CHUNK_COUNT, T_CHUNK_SLOT,           // int32 [nchunks]   OUT
       T_XCD_OFF,         // int32 [9]    OUT
       T_TABLE_COUNT };
enum { T_LAST_OF_PHASE_A = T_SIZES };

inline Geom geom_of(const int64_t* g) {
    Geom G;
    G.n = (int32_t)g[GE_N]; G.B = (int32_t)g[GE_B]; G.chunk = (int32_t)g[GE_CHUNK]; G.mult = (int32_t)g[GE_MULT]; G.R = g[GE_R]; G.P = g[GE_P];
    return G;
}
inline bool geom_ok(const Geom& G) {
    return G.n > 0 && G.B > 0 && G.chunk > 0 && G.mult > 0 && G.R > 0 && G.R <= 0x7fffffffLL && G.P == (int64_t)G.n * G.n * G.B &&
           G.P <= 0x7fffffffLL && G.n < (1 << 20);
}
inline Chunks chunks_of(void** t) {
    Chunks c;
    c.type = (int32_t*)t[T_C_TYPE]; c.start = (int32_t*)t[T_C_START]; c.cnt = (int32_t*)t[T_C_CNT]; c.slot = (int32_t*)t[T_C_SLOT];
    c.xf = (int32_t*)t[T_C_XF]; c.xl = (int32_t*)t[T_C_XL]; c.home = (int32_t*)t[T_C_HOME]; c.key0 = (int64_t*)t[T_C_KEY0];
    return c;
}

}  // namespace gtos_relindex_dev
