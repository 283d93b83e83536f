// kpx_radix.h -- stable LSD radix sort of (uint32 key, int32 value) pairs for the FRAME-SIZED sorts of the pipeline (up to a few
// million pairs: the voxel keys of a frame's clouds, grid cells, Morton codes), hand-written for gfx950.
//
// The vendor Onesweep spends, on 1.1M pairs, 13 dispatches (histogram, scan, 3 passes, 8 clears of its look-back state) of
// which each pass lasts 26-29 us: its decoupled look-back is a chain of device-scope round trips, and at this size every tile
// of a pass is resident at once, so the chain is the whole pass.  A dependent kernel costs ~3 us on this part, a device-scope
// round trip ~1 us -- so this sort takes no look-back at all:
//   * radix_hist_kernel: per tile of 2048 pairs the 256 digit counts (LDS atomics) -> tile_hist[tile][digit], and the same
//     counts added to group_hist[tile / 32][digit] (one global atomic per digit and tile);
//   * radix_scatter_kernel: thread d of a tile obtains the global start of ITS digit's run from <= 64 group rows + <= 31 tile rows
//     (independent loads, issued together) and a block scan over the digit totals; the tile's pairs are ranked stably by one
//     ballot-match per key (wave w owns 512 consecutive pairs, slot by slot, so the tile order is the memory order), staged in
//     LDS in sorted order and written out as consecutive runs per digit.
// Two launches per 8-bit pass, one clear per sort: 7 dispatches for the 24-bit keys of a 4-sensor frame.
// Above kRadixMaxPairs the group rows would no longer be a handful per thread; those sorts stay with rocPRIM (kpx_morton.h).
#pragma once
#include "kpx_common.h"

namespace kpx {

constexpr int kRxThreads = 256, kRxWaves = kRxThreads / 64;
constexpr int kRxKeys = 8;                                   // pairs per thread
constexpr int kRxTile = kRxThreads * kRxKeys;                // 2048
constexpr int kRxGroup = 32;                                 // tiles per group row
constexpr int kRxAhead = 32;                                 // histogram rows a thread requests before it waits
constexpr int64_t kRadixMaxPairs = (int64_t)kRxTile * kRxGroup * 64;   // 4M pairs: at most 64 group rows

static inline int64_t radix_tiles(int64_t n) { return cdiv(n > 0 ? n : 1, kRxTile); }
static inline int64_t radix_groups(int64_t n) { return cdiv(radix_tiles(n), kRxGroup); }
static inline int radix_passes(int end_bit) { return end_bit <= 8 ? 1 : (end_bit + 7) / 8; }

struct RadixScratch {
    uint32_t *keys_tmp;
    int32_t *vals_tmp;
    uint32_t *tile_hist;        // [tiles][256], reused by every pass
    uint32_t *group_hist;       // [4 passes][groups][256], cleared once per sort
};
static inline void radix_carve(Arena &a, int64_t n, RadixScratch *s)
{
    const size_t nn = (size_t)(n > 0 ? n : 1);
    s->keys_tmp = a.get<uint32_t>(nn);
    s->vals_tmp = a.get<int32_t>(nn);
    s->tile_hist = a.get<uint32_t>((size_t)radix_tiles(n) * 256);
    s->group_hist = a.get<uint32_t>((size_t)4 * radix_groups(n) * 256);
}

static __global__ __launch_bounds__(kRxThreads) void radix_hist_kernel(const uint32_t *__restrict__ keys, int64_t n, int shift, uint32_t dmask,
                                                                        uint32_t *__restrict__ tile_hist, uint32_t *__restrict__ group_hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kRxTile;
#pragma unroll
    for (int s = 0; s < kRxKeys; ++s) {
        const int64_t i = base + (int64_t)s * kRxThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & dmask], 1u);
    }
    __syncthreads();
    const uint32_t c = h[threadIdx.x];
    tile_hist[(int64_t)blockIdx.x * 256 + threadIdx.x] = c;
    if (c) atomicAdd(&group_hist[(int64_t)(blockIdx.x / kRxGroup) * 256 + threadIdx.x], c);
}

static __global__ __launch_bounds__(kRxThreads) void radix_scatter_kernel(const uint32_t *__restrict__ keys_in, const int32_t *__restrict__ vals_in,
                                                                           uint32_t *__restrict__ keys_out, int32_t *__restrict__ vals_out, int64_t n,
                                                                           int shift, uint32_t dmask, const uint32_t *__restrict__ tile_hist,
                                                                           const uint32_t *__restrict__ group_hist, int32_t n_groups)
{
    __shared__ uint32_t cnt[kRxWaves][256];          // per wave: keys of each digit seen so far -> the wave's count
    __shared__ uint32_t woff[kRxWaves][256];         // first sorted position of (wave, digit) inside the tile
    __shared__ int64_t sbase[256];                   // global position of the digit's run minus its first position inside the tile
    __shared__ uint32_t skey[kRxTile];
    __shared__ int32_t sval[kRxTile];
    __shared__ int sh[kRxWaves + 1];
    const int tile = blockIdx.x, d = threadIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t base = (int64_t)tile * kRxTile;
    const int tile_n = (int)(n - base < kRxTile ? n - base : kRxTile);

    // the pairs of this thread: wave w owns pairs [w * 512, w * 512 + 512) of the tile, slot s the 64 consecutive ones at s * 64
    uint32_t key[kRxKeys];
    int32_t val[kRxKeys];
#pragma unroll
    for (int s = 0; s < kRxKeys; ++s) {
        const int p = wave * (64 * kRxKeys) + s * 64 + lane;
        const bool on = p < tile_n;
        key[s] = on ? keys_in[base + p] : 0xFFFFFFFFu;
        val[s] = on ? vals_in[base + p] : 0;
    }
    // thread d: where digit d's run of this tile starts in the output
    const int grp = tile / kRxGroup;
    uint32_t before = 0u, total = 0u;
    for (int g0 = 0; g0 < n_groups; g0 += kRxAhead) {            // kRxAhead independent loads per trip (the rows are 1 KB apart)
        uint32_t c[kRxAhead];
#pragma unroll
        for (int u = 0; u < kRxAhead; ++u) c[u] = g0 + u < n_groups ? group_hist[(int64_t)(g0 + u) * 256 + d] : 0u;
#pragma unroll
        for (int u = 0; u < kRxAhead; ++u) { total += c[u]; before += g0 + u < grp ? c[u] : 0u; }
    }
    for (int t0 = grp * kRxGroup; t0 < tile; t0 += kRxAhead) {
        uint32_t c[kRxAhead];
#pragma unroll
        for (int u = 0; u < kRxAhead; ++u) c[u] = t0 + u < tile ? tile_hist[(int64_t)(t0 + u) * 256 + d] : 0u;
#pragma unroll
        for (int u = 0; u < kRxAhead; ++u) before += c[u];
    }
#pragma unroll
    for (int w = 0; w < kRxWaves; ++w) cnt[w][d] = 0u;
    int all;
    const int digit_base = block_excl_scan((int)total, sh, &all);            // (barriers inside: cnt is cleared for everyone)

    // stable rank of every pair inside its wave: lanes holding the same digit in this slot, found with one ballot per digit bit
    uint32_t rank[kRxKeys];
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int s = 0; s < kRxKeys; ++s) {
        const int p = wave * (64 * kRxKeys) + s * 64 + lane;
        const bool on = p < tile_n;
        const uint32_t dg = (key[s] >> shift) & dmask;
        unsigned long long m = __builtin_amdgcn_ballot_w64(on);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bb = __builtin_amdgcn_ballot_w64(((dg >> b) & 1u) != 0u);
            m &= ((dg >> b) & 1u) ? bb : ~bb;
        }
        uint32_t c0 = 0u;
        if (on) c0 = cnt[wave][dg];
        wave_lds_fence();
        rank[s] = c0 + (uint32_t)__builtin_popcountll(m & lt);
        if (on && (m & lt) == 0ull) cnt[wave][dg] = c0 + (uint32_t)__builtin_popcountll(m);      // the first lane of the digit
        wave_lds_fence();
    }
    __syncthreads();
    // thread d: the tile's count of digit d, the digit's first sorted position inside the tile, the waves' shares of it
    uint32_t tc = 0u;
#pragma unroll
    for (int w = 0; w < kRxWaves; ++w) tc += cnt[w][d];
    int tot;
    const int tstart = block_excl_scan((int)tc, sh, &tot);
    uint32_t run = (uint32_t)tstart;
#pragma unroll
    for (int w = 0; w < kRxWaves; ++w) { woff[w][d] = run; run += cnt[w][d]; }
    sbase[d] = (int64_t)digit_base + (int64_t)before - (int64_t)tstart;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kRxKeys; ++s) {
        const int p = wave * (64 * kRxKeys) + s * 64 + lane;
        if (p < tile_n) {
            const uint32_t pos = woff[wave][(key[s] >> shift) & dmask] + rank[s];
            skey[pos] = key[s];
            sval[pos] = val[s];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < tile_n; i += kRxThreads) {
        const uint32_t k = skey[i];
        const int64_t o = sbase[(k >> shift) & dmask] + i;
        keys_out[o] = k;
        vals_out[o] = sval[i];
    }
}

// keys_out / vals_out receive the pairs sorted by bits [0, end_bit) of the key (stable).  keys_in / vals_in are not modified and
// may not alias the outputs.  n <= kRadixMaxPairs.
// clear_end: end of a region carved right BEHIND the scratch that the sort's one memset shall clear too (the caller's tile states).
static inline int radix_sort_pairs_u32(const RadixScratch &s, const uint32_t *keys_in, uint32_t *keys_out, const int32_t *vals_in, int32_t *vals_out,
                                       int64_t n, int end_bit, hipStream_t st, const void *clear_end = nullptr)
{
    if (n <= 0) return KPX_OK;
    const int passes = radix_passes(end_bit);
    const int tiles = (int)radix_tiles(n), groups = (int)radix_groups(n);
    {
        char *c0 = reinterpret_cast<char *>(s.group_hist);
        const char *c1 = clear_end ? reinterpret_cast<const char *>(clear_end) : c0 + (size_t)passes * groups * 256 * sizeof(uint32_t);
        KPX_HIP(hipMemsetAsync(c0, 0, (size_t)(c1 - c0), st));
    }
    const uint32_t *k_src = keys_in;
    const int32_t *v_src = vals_in;
    for (int p = 0; p < passes; ++p) {
        const bool to_out = ((passes - 1 - p) & 1) == 0;            // the last pass writes the caller's buffers
        uint32_t *k_dst = to_out ? keys_out : s.keys_tmp;
        int32_t *v_dst = to_out ? vals_out : s.vals_tmp;
        uint32_t *gh = s.group_hist + (size_t)p * groups * 256;
        const int bits = end_bit - 8 * p >= 8 ? 8 : (end_bit - 8 * p < 1 ? 1 : end_bit - 8 * p);       // the last digit may be narrower
        const uint32_t dmask = (1u << bits) - 1u;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(tiles), dim3(kRxThreads), 0, st, k_src, n, 8 * p, dmask, s.tile_hist, gh);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(tiles), dim3(kRxThreads), 0, st, k_src, v_src, k_dst, v_dst, n, 8 * p, dmask, s.tile_hist, gh, groups);
        k_src = k_dst;
        v_src = v_dst;
    }
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

}  // namespace kpx
