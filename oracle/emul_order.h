// TEST INFRASTRUCTURE (oracle/): the order in which the emulation libraries visit the "threads" of a stage.  A GPU runs them in no
// particular order, so a stage whose result depends on the order has a race the ascending loop would hide.  gtos_emul_set_order(0):
// ascending (default); (1): descending; (s > 1): a pseudo-random permutation seeded by s.  The tests run every builder under several
// orders and require identical arrays.
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace gtos_emul {
inline uint64_t& order_seed() { static uint64_t s = 0; return s; }

template <typename F>
void for_each(int64_t n, F&& fn) {
    const uint64_t seed = order_seed();
    if (seed == 0) { for (int64_t i = 0; i < n; ++i) fn(i); return; }
    if (seed == 1) { for (int64_t i = n - 1; i >= 0; --i) fn(i); return; }
    std::vector<int64_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + (uint64_t)n;
    for (int64_t i = n - 1; i > 0; --i) {                       // Fisher-Yates with a splitmix stream
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        std::swap(idx[i], idx[(int64_t)(z % (uint64_t)(i + 1))]);
    }
    for (int64_t i = 0; i < n; ++i) fn(idx[i]);
}
}  // namespace gtos_emul

#define GTOS_EMUL_ORDER_ENTRY(prefix) extern "C" void prefix##_set_order(uint64_t seed) { gtos_emul::order_seed() = seed; }
