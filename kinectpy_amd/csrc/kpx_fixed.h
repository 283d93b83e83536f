// kpx_fixed.h -- exact, order-independent accumulation of fp64 values in 128-bit fixed point (64 integer + 64
// fractional bits; exact for 2^-64-aligned values, i.e. every double with |v| >= 2^-12 and every float32 with
// |v| >= 2^-41; smaller magnitudes are truncated towards zero at 2^-64; |v| < 2^63).  Two 64-bit integer atomics per
// add, carry propagated by whoever wraps the low word: integer addition is associative, so totals do not depend on the
// order of the adds -- parallel sums stay bitwise reproducible.  The oracle restates the same conversion with __int128.
#pragma once
#include "kpx_common.h"

namespace kpx {

// (Round 4 tried a carry-free layout -- the fractional word's halves in words of their own, three INDEPENDENT atomics per add instead of a
// second one that waits for the first one's carry: no measurable change in the ICP iteration (the block's barrier, not the add, was what the
// interval measured) and +50 % atomic traffic (WRITE_SIZE 0.48 -> 0.83 MB per launch); not kept.)
constexpr int kFixedWords = 2;

// v -> its 128-bit two's complement fixed-point words (exact under the conditions above)
__device__ __forceinline__ void fixed_split(double v, unsigned long long &lo, unsigned long long &hi)
{
    const bool neg = v < 0.0;
    const double m = fabs(v);
    const double ip = floor(m);
    hi = (unsigned long long)ip;
    lo = (unsigned long long)((m - ip) * 18446744073709551616.0);     // frac * 2^64, exact
    if (neg) {                                              // two's complement of the 128-bit magnitude
        lo = ~lo + 1ull;
        hi = ~hi + (lo == 0ull ? 1ull : 0ull);
    }
}
// (lo, hi) += (l, h) in registers: integer addition, exact and order-independent like the atomics below
__device__ __forceinline__ void fixed_accumulate(unsigned long long &lo, unsigned long long &hi, unsigned long long l, unsigned long long h)
{
    lo += l;
    hi += h + (lo < l ? 1ull : 0ull);
}
__device__ __forceinline__ void fixed_add_words(unsigned long long *acc2, unsigned long long lo, unsigned long long hi)
{
    if (hi == 0ull && lo == 0ull) return;
    const unsigned long long old = atomicAdd(acc2, lo);
    const unsigned long long carry = (old + lo) < old ? 1ull : 0ull;
    if (hi + carry != 0ull) atomicAdd(acc2 + 1, hi + carry);
}
// The same add with both atomics RETURNING: when the call returns, the add has been performed at the device's point of
// coherence (what a later ticket of the same block may be ordered behind without a release fence).
__device__ __forceinline__ void fixed_add_words_performed(unsigned long long *acc2, unsigned long long lo, unsigned long long hi)
{
    if (hi == 0ull && lo == 0ull) return;
    const unsigned long long old = __hip_atomic_fetch_add(acc2, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long carry = (old + lo) < old ? 1ull : 0ull;
    if (hi + carry != 0ull) {
        const unsigned long long back = __hip_atomic_fetch_add(acc2 + 1, hi + carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(back));                       // the value is waited for
    }
}
__device__ __forceinline__ void fixed_add(unsigned long long *acc2, double v)
{
    unsigned long long lo, hi;
    fixed_split(v, lo, hi);
    fixed_add_words(acc2, lo, hi);
}
__device__ __forceinline__ void fixed_add_performed(unsigned long long *acc2, double v)
{
    unsigned long long lo, hi;
    fixed_split(v, lo, hi);
    fixed_add_words_performed(acc2, lo, hi);
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
