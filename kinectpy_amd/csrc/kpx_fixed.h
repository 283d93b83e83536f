// kpx_fixed.h -- exact, order-independent accumulation of fp64 values in 128-bit fixed point (64 integer + 64
// fractional bits; exact for 2^-64-aligned values, i.e. every double with |v| >= 2^-12 and every float32 with
// |v| >= 2^-41; smaller magnitudes are truncated towards zero at 2^-64; |v| < 2^63).  Integer addition is associative, so
// totals do not depend on the order of the adds -- parallel sums stay bitwise reproducible.  The oracle restates the same
// conversion with __int128.
#pragma once
#include "kpx_common.h"

namespace kpx {

// Storage: kFixedWords = 3 words per accumulator -- the low and the high 32 bits of the fractional word each in a 64-bit word of
// their own, then the integer word -- so that the three atomics of an add are INDEPENDENT (until round 4: two words, the second add
// waited for the first one's return value to carry: two dependent round trips to the memory side per add, 4-8 us of every ICP
// iteration).  The 32-bit pieces cannot overflow their 64-bit words before 2^32 adds; fixed_words_value folds the carries when the
// total is read: the same 128-bit integer as before, so nothing that depends on the totals changed.
constexpr int kFixedWords = 3;
__device__ __forceinline__ void fixed_split(double v, unsigned long long w[3])
{
    const bool neg = v < 0.0;
    const double m = fabs(v);
    const double ip = floor(m);
    unsigned long long hi = (unsigned long long)ip;
    unsigned long long lo = (unsigned long long)((m - ip) * 18446744073709551616.0);     // frac * 2^64, exact
    if (neg) {                                              // two's complement of the 128-bit magnitude
        lo = ~lo + 1ull;
        hi = ~hi + (lo == 0ull ? 1ull : 0ull);
    }
    w[0] = lo & 0xFFFFFFFFull; w[1] = lo >> 32; w[2] = hi;
}
__device__ __forceinline__ void fixed_add(unsigned long long *acc3, double v)
{
    unsigned long long w[3];
    fixed_split(v, w);
#pragma unroll
    for (int e = 0; e < 3; ++e)
        if (w[e] != 0ull) atomicAdd(acc3 + e, w[e]);
}
// The same add with its atomics RETURNING: when the call returns, the add has been performed at the device's point of
// coherence (what a later ticket of the same block may be ordered behind without a release fence).
__device__ __forceinline__ void fixed_add_performed(unsigned long long *acc3, double v)
{
    unsigned long long w[3], back[3] = { 0ull, 0ull, 0ull };
    fixed_split(v, w);
#pragma unroll
    for (int e = 0; e < 3; ++e)
        if (w[e] != 0ull) back[e] = __hip_atomic_fetch_add(acc3 + e, w[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(back[0]), "v"(back[1]), "v"(back[2]));      // the values are waited for (together)
}
// totals of the three words (each summed with wrap-around over whatever copies the caller keeps) -> (lo, hi)
__device__ __forceinline__ void fixed_fold(unsigned long long w0, unsigned long long w1, unsigned long long w2, unsigned long long &lo, unsigned long long &hi)
{
    lo = w0 + (w1 << 32);
    hi = w2 + (w1 >> 32) + (lo < w0 ? 1ull : 0ull);
}
// (lo, hi) two's complement -> double: (double)hi + (double)lo * 2^-64 on the magnitude
__device__ __forceinline__ double fixed_value(unsigned long long lo, unsigned long long hi)
{
    const bool neg = (long long)hi < 0;
    if (neg) { lo = ~lo + 1ull; hi = ~hi + (lo == 0ull ? 1ull : 0ull); }
    const double v = (double)hi + (double)lo * 5.421010862427522170037e-20;              // 2^-64
    return neg ? -v : v;
}

}  // namespace kpx
