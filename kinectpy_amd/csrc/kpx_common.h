// kpx_common.h -- shared host/device helpers of libkinectpx.so (gfx950 only).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/kinectpx.h"

#define KPX_EXPORT extern "C" __attribute__((visibility("default")))

namespace kpx {

// ---- errors -----------------------------------------------------------------------------------
extern thread_local char g_err[512];
int fail(int code, const char *fmt, ...);

#define KPX_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return kpx::fail(KPX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),  \
                             __FILE__, __LINE__);                                                  \
    } while (0)
#define KPX_LAUNCH_CHECK() KPX_HIP(hipGetLastError())
#define KPX_REQUIRE(cond, ...)                                                                     \
    do {                                                                                           \
        if (!(cond)) return kpx::fail(KPX_ERR_INVALID, __VA_ARGS__);                               \
    } while (0)

// ---- workspace bump allocator (same carve sequence for the size query and the real call) -------
struct Arena {
    char *base;
    size_t off, cap;
    bool dry;
    explicit Arena(void *p, size_t bytes) : base((char *)p), off(0), cap(bytes), dry(p == nullptr) {}
    template <class T> T *get(size_t count)
    {
        size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        T *r = dry ? nullptr : (T *)(base + off);
        off += bytes;
        return r;
    }
    bool ok() const { return dry || off <= cap; }
};
#define KPX_ARENA_CHECK(a)                                                                         \
    do {                                                                                           \
        if (!(a).ok())                                                                             \
            return kpx::fail(KPX_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu",   \
                             (a).off, (a).cap);                                                    \
    } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// The vendor primitives' temporary-storage queries (hipcub / rocPRIM called with a null buffer) cost tens of microseconds each when
// several frame threads are inside the runtime, and every operator asked them again on every call while carving its workspace -- a
// 4-sensor frame asked ~20 of them, and its stream sat idle meanwhile (profiles/r05/overlap_timeline_*.txt).  Their answers depend
// only on (call site, element count): asked once per process and remembered.
bool memo_bytes_lookup(unsigned site, int64_t n, size_t *bytes);
void memo_bytes_store(unsigned site, int64_t n, size_t bytes);
template <class F> static inline size_t memo_bytes(unsigned site, int64_t n, F &&query)
{
    size_t b = 0;
    if (memo_bytes_lookup(site, n, &b)) return b;
    b = query();
    memo_bytes_store(site, n, b);
    return b;
}

// 16-byte loads / stores of data that is touched once (streaming operators): non-temporal, so that a pass over a cloud does not
// evict what other kernels of the frame keep in L2 / Infinity Cache.  -DKPX_STREAM_PLAIN: ordinary accesses (A/B switch).
typedef unsigned int kpx_u4 __attribute__((ext_vector_type(4)));
template <class V> __device__ __forceinline__ V stream_load16(const V *p)
{
    static_assert(sizeof(V) == 16, "16-byte vectors only");
#ifdef KPX_STREAM_PLAIN
    return *p;
#else
    const kpx_u4 r = __builtin_nontemporal_load(reinterpret_cast<const kpx_u4 *>(p));
    V v;
    __builtin_memcpy(&v, &r, 16);
    return v;
#endif
}
template <class V> __device__ __forceinline__ void stream_store16(const V &v, V *p)
{
    static_assert(sizeof(V) == 16, "16-byte vectors only");
#ifdef KPX_STREAM_PLAIN
    *p = v;
#else
    kpx_u4 r;
    __builtin_memcpy(&r, &v, 16);
    __builtin_nontemporal_store(r, reinterpret_cast<kpx_u4 *>(p));
#endif
}
#define KPX_STREAM_LOAD(p) stream_load16(p)
#define KPX_STREAM_STORE(v, p) stream_store16((v), (p))

// ---- HIP-event profiler (armed by kpx_prof_begin; a no-op otherwise) ----------------------------
bool prof_armed();
struct ProfScope {
    int slot;
    hipStream_t st;
    ProfScope(int kernel_id, double work, hipStream_t stream);
    ~ProfScope();
};

// ---- device helpers ----------------------------------------------------------------------------
constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

template <class T> __device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;   // valid in lane 0
}
template <class T> __device__ __forceinline__ T wave_min(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T t = __shfl_down(v, o, 64); v = t < v ? t : v; }
    return v;
}
template <class T> __device__ __forceinline__ T wave_max(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T t = __shfl_down(v, o, 64); v = t > v ? t : v; }
    return v;
}

// wave-level ordering of LDS traffic (the DS unit executes one wave's instructions in order; this only stops the
// compiler from moving accesses across the point)
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Block-wide sum in a FIXED order (wave-tree, then waves in order): bitwise reproducible run to run.
// All threads must call; result valid in thread 0.  `sh` holds >= blockDim/64 elements of T.
template <class T> __device__ __forceinline__ T block_sum(T v, T *sh)
{
    v = wave_sum(v);
    __syncthreads();
    if (lane_id() == 0) sh[wave_id()] = v;
    __syncthreads();
    T r = T(0);
    if (threadIdx.x == 0) {
        int nw = (blockDim.x + 63) >> 6;
        for (int w = 0; w < nw; ++w) r += sh[w];
    }
    return r;
}

// inclusive scan across the wave
__device__ __forceinline__ int wave_incl_scan(int v)
{
    int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(v, o, 64); if (l >= o) v += t; }
    return v;
}
// Block exclusive scan of one int per thread; returns the exclusive prefix, *total = block sum.
// sh: >= blockDim/64 + 1 ints.
__device__ __forceinline__ int block_excl_scan(int v, int *sh, int *total)
{
    int incl = wave_incl_scan(v);
    int nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane_id() == 63) sh[wave_id()] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < nw; ++w) { int t = sh[w]; sh[w] = run; run += t; }
        sh[nw] = run;
    }
    __syncthreads();
    int base = sh[wave_id()];
    *total = sh[nw];
    return base + incl - v;
}

// ---- decoupled look-back (Merrill & Garland): a tile's exclusive prefix without a scan pass ----------------------------
// One 64-bit word per tile, cleared before the launch: bits 63:62 = 0 nothing yet, 1 the tile's own total, 2 its inclusive
// prefix; bits 31:0 = the value.  Flag and value travel in ONE word, so a relaxed device-scope load / store pair is all the
// ordering there is (no fences, no L2 write-backs).  A tile publishes its total, then its first wave walks back over the
// preceding tiles 64 at a time -- lane l reads tile (pos - l) -- until it meets an inclusive prefix.  Tiles are taken in
// blockIdx order (workgroups are dispatched in linear-id order and a resident block never yields), so every predecessor of
// a running tile is running or finished: the spin terminates.
// All threads of the block call it; sh: one int of LDS.  Returns the exclusive prefix of `tile` within its row of states.
constexpr unsigned long long kTileTotal = 1ull << 62, kTilePrefix = 2ull << 62;
__device__ __forceinline__ int lookback_exclusive(unsigned long long *__restrict__ state, int tile, int tot, int *sh)
{
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0)
            __hip_atomic_store(&state[tile], (tile == 0 ? kTilePrefix : kTileTotal) | (unsigned long long)(unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int excl = 0;
        if (tile > 0) {
            int pos = tile - 1;
            for (;;) {
                const int idx = pos - lane;
                unsigned long long st;
                for (;;) {
                    st = idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kTilePrefix;
                    if (__ballot((st >> 62) == 0ull) == 0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                const unsigned long long pm = __ballot((st >> 62) == 2ull);
                const int first = pm ? __builtin_ctzll(pm) : 64;
                int v = lane <= first ? (int)(unsigned)(st & 0xFFFFFFFFull) : 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                excl += v;
                if (pm) break;
                pos -= 64;
            }
            if (lane == 0)
                __hip_atomic_store(&state[tile], kTilePrefix | (unsigned long long)(unsigned)(excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) *sh = excl;
    }
    __syncthreads();
    return *sh;
}

// ---- order-preserving stream compaction -------------------------------------------------------------------------------
// A Pred is a device functor  bool operator()(int64_t item, int frame) const
// An Emit is a device functor void operator()(int64_t item, int frame, int32_t dst) const
constexpr int kCompactThreads = 256;
constexpr int kCompactItems = 8;
constexpr int kCompactTile = kCompactThreads * kCompactItems;

template <class Pred>
__global__ __launch_bounds__(kCompactThreads) void compact_count_kernel(Pred pred, int64_t n, int32_t *block_counts)
{
    __shared__ int sh[kCompactThreads / 64];
    const int frame = blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * kCompactTile + (int64_t)threadIdx.x * kCompactItems;
    int c = 0;
#pragma unroll
    for (int k = 0; k < kCompactItems; ++k) {
        int64_t i = base + k;
        if (i < n && pred(i, frame)) ++c;
    }
    c = block_sum(c, sh);
    if (threadIdx.x == 0) block_counts[(int64_t)frame * gridDim.x + blockIdx.x] = c;
}

// one block per frame: exclusive scan of the per-tile counts in place, total -> d_count[frame].  A round scans 8 counts
// per thread (one 3-barrier block scan per 8 x blockDim tiles; a 64M-point selection has 31250 tiles).
static inline int compact_scan_threads(int64_t tiles) { return tiles > 2048 ? 1024 : 256; }
static __device__ __forceinline__ void compact_scan_body(int32_t *bc, int32_t nblocks, int32_t *d_count_slot)
{
    __shared__ int sh[1024 / 64 + 1];
    int carry = 0;
    for (int32_t b0 = 0; b0 < nblocks; b0 += 8 * (int32_t)blockDim.x) {
        const int32_t b = b0 + 8 * (int32_t)threadIdx.x;
        int v[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = b + k < nblocks ? bc[b + k] : 0; sum += v[k]; }
        int tot;
        int run = carry + block_excl_scan(sum, sh, &tot);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (b + k < nblocks) bc[b + k] = run;
            run += v[k];
        }
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && d_count_slot) *d_count_slot = carry;
}
static __global__ __launch_bounds__(1024) void compact_scan_kernel(int32_t *block_counts, int32_t nblocks, int32_t *d_count)
{
    compact_scan_body(block_counts + (int64_t)blockIdx.x * nblocks, nblocks, d_count ? d_count + blockIdx.x : nullptr);
}
// the two lists of one predicate pass (slab split) side by side: block 0 scans list A, block 1 list B
static __global__ __launch_bounds__(1024) void compact_scan2_kernel(int32_t *counts_a, int32_t *counts_b, int32_t nblocks, int32_t *d_count_a,
                                                                    int32_t *d_count_b)
{
    compact_scan_body(blockIdx.x ? counts_b : counts_a, nblocks, blockIdx.x ? d_count_b : d_count_a);
}

template <class Pred, class Emit>
__global__ __launch_bounds__(kCompactThreads) void compact_scatter_kernel(Pred pred, Emit emit, int64_t n,
                                                                            const int32_t *block_offsets)
{
    __shared__ int sh[kCompactThreads / 64 + 1];
    const int frame = blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * kCompactTile + (int64_t)threadIdx.x * kCompactItems;
    unsigned flags = 0;
    int c = 0;
#pragma unroll
    for (int k = 0; k < kCompactItems; ++k) {
        int64_t i = base + k;
        if (i < n && pred(i, frame)) { flags |= 1u << k; ++c; }
    }
    int tot;
    int ex = block_excl_scan(c, sh, &tot);
    int32_t dst = block_offsets[(int64_t)frame * gridDim.x + blockIdx.x] + ex;
#pragma unroll
    for (int k = 0; k < kCompactItems; ++k)
        if (flags & (1u << k)) emit(base + k, frame, dst++);
}

// ---- index compaction over a float32 (n,3) cloud with a predicate on the coordinates ----------------------------
// The selections of the path (half space, slab split).  ONE pass over the points: a thread loads its 8 points as six
// 16-byte vectors (96 contiguous bytes), evaluates the predicate -- which may feed two lists at once, bit 0 = list A,
// bit 1 = list B -- and stores its 8 decisions per list as one byte; the scatter pass reads those bytes (n/8 bytes per
// list instead of the 12 n of the points), stages the kept indices of a block in LDS and writes them as consecutive
// dwords.  HBM traffic: 12 n + 4 K + n/4 per list, against 24 n + 4 K when the scatter pass re-evaluates the predicate.
// The decisions are stored as ballots: bit i of the flag array <-> point i (so the scatter pass's thread t of a block
// reads byte t = points 8 t .. 8 t + 7 of the tile).  A wave owns 512 points = 6 KiB: six lane-contiguous 16-byte loads
// (every load instruction of the wave reads one contiguous KiB -- 96-byte strides per lane ran at 60 % of this) into
// LDS, then lane l evaluates points 64 k + l, k = 0..7, read back with a 3-dword stride (conflict-free).
// flags_*: tiles * kCompactThreads bytes; counts_*: one int per tile.  TWO = false: list A only.
template <class PredXYZ, bool TWO>
__global__ __launch_bounds__(kCompactThreads) void compact_pts_flag_kernel(const float *__restrict__ pts, int64_t n, PredXYZ pred, bool aligned,
                                                                           uint8_t *__restrict__ flags_a, int32_t *__restrict__ counts_a,
                                                                           uint8_t *__restrict__ flags_b, int32_t *__restrict__ counts_b)
{
    __shared__ __align__(16) float stage[kCompactThreads / 64][1536];
    __shared__ int sh[2][kCompactThreads / 64];
    const int wave = wave_id(), lane = lane_id();
    const int64_t wbase = (int64_t)blockIdx.x * kCompactTile + (int64_t)wave * 512;
    const bool full = aligned && wbase + 512 <= n;
    if (full) {
        const float4 *src = reinterpret_cast<const float4 *>(pts + 3 * wbase);
        float4 *dst = reinterpret_cast<float4 *>(stage[wave]);
#pragma unroll
        for (int r = 0; r < 6; ++r) dst[64 * r + lane] = stream_load16(src + 64 * r + lane);       // the points are read once: non-temporal
        wave_lds_fence();
    }
    unsigned long long ma = 0, mb = 0;          // lane k < 8 keeps the ballot of row k
    int ca = 0, cb = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int p = 64 * k + lane;
        unsigned r = 0;
        if (full) r = pred(stage[wave][3 * p], stage[wave][3 * p + 1], stage[wave][3 * p + 2]);
        else if (wbase + p < n) r = pred(pts[3 * (wbase + p)], pts[3 * (wbase + p) + 1], pts[3 * (wbase + p) + 2]);
        const unsigned long long qa = __ballot((r & 1u) != 0);
        ca += __builtin_popcountll(qa);
        if (lane == k) ma = qa;
        if (TWO) {
            const unsigned long long qb = __ballot((r & 2u) != 0);
            cb += __builtin_popcountll(qb);
            if (lane == k) mb = qb;
        }
    }
    const int64_t slot = ((int64_t)blockIdx.x * (kCompactThreads / 64) + wave) * 8 + lane;       // in units of 8 bytes
    if (lane < 8) {
        reinterpret_cast<unsigned long long *>(flags_a)[slot] = ma;
        if (TWO) reinterpret_cast<unsigned long long *>(flags_b)[slot] = mb;
    }
    if (lane == 0) { sh[0][wave] = ca; sh[1][wave] = cb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int ta = 0, tb = 0;
        for (int w = 0; w < kCompactThreads / 64; ++w) { ta += sh[0][w]; tb += sh[1][w]; }
        counts_a[blockIdx.x] = ta;
        if (TWO) counts_b[blockIdx.x] = tb;
    }
}
// blockIdx.y == 1: the second list of a two-list pass (*_b; NULL for one list)
static __global__ __launch_bounds__(kCompactThreads) void compact_flags_scatter_kernel(const uint8_t *__restrict__ flag_bytes,
                                                                                       const int32_t *__restrict__ block_offsets,
                                                                                       int32_t *__restrict__ idx,
                                                                                       const uint8_t *__restrict__ flag_bytes_b = nullptr,
                                                                                       const int32_t *__restrict__ block_offsets_b = nullptr,
                                                                                       int32_t *__restrict__ idx_b = nullptr)
{
    __shared__ int sh[kCompactThreads / 64 + 1];
    __shared__ int32_t si[kCompactTile];
    if (blockIdx.y) { flag_bytes = flag_bytes_b; block_offsets = block_offsets_b; idx = idx_b; }
    const int64_t base = (int64_t)blockIdx.x * kCompactTile + (int64_t)threadIdx.x * kCompactItems;
    const unsigned flags = flag_bytes[(int64_t)blockIdx.x * kCompactThreads + threadIdx.x];
    int tot;
    int pos = block_excl_scan(__builtin_popcount(flags), sh, &tot);
#pragma unroll
    for (int k = 0; k < kCompactItems; ++k)
        if (flags & (1u << k)) si[pos++] = (int32_t)(base + k);
    __syncthreads();
    const int64_t o = block_offsets[blockIdx.x];
    for (int e = threadIdx.x; e < tot; e += kCompactThreads) idx[o + e] = si[e];
}

// ONE pass: predicate evaluated once, the tile's offset by decoupled look-back, items emitted in place (the count -> scan ->
// scatter kernels above read the input twice and cost three launches).  tile_state: 64-bit word per (frame, tile), cleared.
template <class Pred, class Emit>
__global__ __launch_bounds__(kCompactThreads) void compact_onepass_kernel(Pred pred, Emit emit, int64_t n, unsigned long long *__restrict__ tile_state,
                                                                            int32_t *__restrict__ d_count)
{
    __shared__ int sh[kCompactThreads / 64 + 2];
    const int frame = blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * kCompactTile + (int64_t)threadIdx.x * kCompactItems;
    unsigned flags = 0;
    int c = 0;
#pragma unroll
    for (int k = 0; k < kCompactItems; ++k) {
        int64_t i = base + k;
        if (i < n && pred(i, frame)) { flags |= 1u << k; ++c; }
    }
    int tot;
    const int ex = block_excl_scan(c, sh, &tot);
    __syncthreads();
    const int before = lookback_exclusive(tile_state + (int64_t)frame * gridDim.x, blockIdx.x, tot, sh + kCompactThreads / 64 + 1);
    int32_t dst = before + ex;
#pragma unroll
    for (int k = 0; k < kCompactItems; ++k)
        if (flags & (1u << k)) emit(base + k, frame, dst++);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && d_count) d_count[frame] = before + tot;
}

// Host driver.  ws_counts: int32 [frames * compact_ws_ints(n)] (8-byte aligned: it holds the 64-bit tile states).
static inline int64_t compact_tiles(int64_t n) { return cdiv(n > 0 ? n : 1, kCompactTile); }
static inline int64_t compact_ws_ints(int64_t n) { return 2 * compact_tiles(n) + 2; }
// One pass or three launches?  Measured on MI355X: on batches that stream hundreds of MB the look-back costs bandwidth (every
// tile holds its registers and LDS through a device-scope round trip: 256 frames of depth -> cloud ran at 2.2 TB/s against 3.0
// with count -> scan -> scatter), while a frame-sized problem is a chain of launches whose NUMBER is what matters.  So the one-pass
// kernels serve problems of at most kOnePassTiles tiles; KPX_ONEPASS=0 / 1 forces either path (A/B runs).
constexpr int64_t kOnePassTiles = 2048;
static inline bool use_onepass(int64_t tiles_total)
{
    static const int mode = [] { const char *e = getenv("KPX_ONEPASS"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    return mode < 0 ? tiles_total <= kOnePassTiles : mode != 0;
}
// state_is_clear: the caller cleared ws_counts together with a neighbouring region it had to clear anyway (one memset dispatch
// less per call; a frame is a chain of ~5 us dispatches)
template <class Pred, class Emit>
int compact(Pred pred, Emit emit, int64_t n, int32_t frames, int32_t *ws_counts, int32_t *d_count, hipStream_t st, bool state_is_clear = false)
{
    const int32_t tiles = (int32_t)compact_tiles(n);
    dim3 grid(tiles, frames);
    if (use_onepass((int64_t)tiles * frames)) {
        unsigned long long *state = reinterpret_cast<unsigned long long *>(ws_counts);
        if (!state_is_clear) KPX_HIP(hipMemsetAsync(state, 0, (size_t)tiles * frames * sizeof(unsigned long long), st));
        hipLaunchKernelGGL((compact_onepass_kernel<Pred, Emit>), grid, dim3(kCompactThreads), 0, st, pred, emit, n, state, d_count);
        KPX_LAUNCH_CHECK();
        return KPX_OK;
    }
    hipLaunchKernelGGL(compact_count_kernel<Pred>, grid, dim3(kCompactThreads), 0, st, pred, n, ws_counts);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(frames), dim3(compact_scan_threads(tiles)), 0, st, ws_counts, tiles, d_count);
    hipLaunchKernelGGL((compact_scatter_kernel<Pred, Emit>), grid, dim3(kCompactThreads), 0, st, pred, emit, n, ws_counts);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}
// Scratch of compact_points: per list one flag byte per thread and one count per tile.
struct CompactPtsScratch {
    uint8_t *flags[2];
    int32_t *counts[2];
};
static inline void compact_pts_carve(Arena &a, int64_t n, CompactPtsScratch *s)
{
    const size_t tiles = (size_t)compact_tiles(n);
    for (int l = 0; l < 2; ++l) {
        s->flags[l] = a.get<uint8_t>(tiles * kCompactThreads);
        s->counts[l] = a.get<int32_t>(tiles);
    }
}
// ONE pass for one list: the wave-staged predicate evaluation of compact_pts_flag_kernel, then the tile's offset by look-back and
// the kept indices written from LDS as consecutive dwords -- the points are read once, nothing else is read or written but the
// index list (12 n + 4 K bytes, the algorithmic minimum) and one launch replaces three.
template <class PredXYZ>
__global__ __launch_bounds__(kCompactThreads) void compact_pts_onepass_kernel(const float *__restrict__ pts, int64_t n, PredXYZ pred, bool aligned,
                                                                              unsigned long long *__restrict__ tile_state, int32_t *__restrict__ idx,
                                                                              int32_t *__restrict__ d_count)
{
    __shared__ __align__(16) float stage[kCompactThreads / 64][1536];
    __shared__ int sh[kCompactThreads / 64 + 2];
    __shared__ int wave_base[kCompactThreads / 64 + 1];
    const int wave = wave_id(), lane = lane_id();
    const int64_t wbase = (int64_t)blockIdx.x * kCompactTile + (int64_t)wave * 512;
    const bool full = aligned && wbase + 512 <= n;
    if (full) {
        const float4 *src = reinterpret_cast<const float4 *>(pts + 3 * wbase);
        float4 *dst = reinterpret_cast<float4 *>(stage[wave]);
#pragma unroll
        for (int r = 0; r < 6; ++r) dst[64 * r + lane] = stream_load16(src + 64 * r + lane);       // the points are read once: non-temporal
        wave_lds_fence();
    }
    unsigned long long m[8];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int p = 64 * k + lane;
        unsigned r = 0;
        if (full) r = pred(stage[wave][3 * p], stage[wave][3 * p + 1], stage[wave][3 * p + 2]);
        else if (wbase + p < n) r = pred(pts[3 * (wbase + p)], pts[3 * (wbase + p) + 1], pts[3 * (wbase + p) + 2]);
        m[k] = __ballot((r & 1u) != 0);
        cnt += __builtin_popcountll(m[k]);
    }
    if (lane == 0) sh[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < kCompactThreads / 64; ++w) { wave_base[w] = run; run += sh[w]; }
        wave_base[kCompactThreads / 64] = run;
    }
    __syncthreads();
    const int tot = wave_base[kCompactThreads / 64];
    const int before = lookback_exclusive(tile_state, blockIdx.x, tot, sh + kCompactThreads / 64 + 1);
    // kept indices of the wave's 512 points in ascending order: row k (points 64 k + l) after the rows before it
    int32_t *si = reinterpret_cast<int32_t *>(stage[wave]);          // the staged points are consumed: reuse the wave's LDS
    wave_lds_fence();
    int off = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if ((m[k] >> lane) & 1ull) si[off + __builtin_popcountll(m[k] & ((1ull << lane) - 1ull))] = (int32_t)(wbase + 64 * k + lane);
        off += __builtin_popcountll(m[k]);
    }
    wave_lds_fence();
    const int64_t o = (int64_t)before + wave_base[wave];
    for (int e = lane; e < cnt; e += 64) idx[o + e] = si[e];
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *d_count = before + tot;
}

// idx_b / d_count_b == nullptr: one list (the predicate's bit 0).
template <class PredXYZ>
int compact_points(const float *pts, int64_t n, PredXYZ pred, int32_t *idx_a, int32_t *d_count_a, int32_t *idx_b, int32_t *d_count_b,
                   const CompactPtsScratch &s, hipStream_t st)
{
    const int32_t tiles = (int32_t)compact_tiles(n);
    const bool aligned = ((uintptr_t)pts % 16) == 0;
    if (!idx_b && use_onepass(tiles)) {                 // one list, frame-sized: one pass
        unsigned long long *state = reinterpret_cast<unsigned long long *>(s.flags[0]);     // tiles * 256 bytes >= tiles * 8
        KPX_HIP(hipMemsetAsync(state, 0, (size_t)tiles * sizeof(unsigned long long), st));
        hipLaunchKernelGGL((compact_pts_onepass_kernel<PredXYZ>), dim3(tiles), dim3(kCompactThreads), 0, st, pts, n, pred, aligned, state, idx_a, d_count_a);
        KPX_LAUNCH_CHECK();
        return KPX_OK;
    }
    if (idx_b)
        hipLaunchKernelGGL((compact_pts_flag_kernel<PredXYZ, true>), dim3(tiles), dim3(kCompactThreads), 0, st, pts, n, pred, aligned, s.flags[0],
                           s.counts[0], s.flags[1], s.counts[1]);
    else
        hipLaunchKernelGGL((compact_pts_flag_kernel<PredXYZ, false>), dim3(tiles), dim3(kCompactThreads), 0, st, pts, n, pred, aligned, s.flags[0],
                           s.counts[0], s.flags[1], s.counts[1]);
    if (idx_b) {                                        // both lists in one scan launch and one scatter launch (round 5: five launches -> three)
        hipLaunchKernelGGL(compact_scan2_kernel, dim3(2), dim3(compact_scan_threads(tiles)), 0, st, s.counts[0], s.counts[1], tiles, d_count_a, d_count_b);
        hipLaunchKernelGGL(compact_flags_scatter_kernel, dim3(tiles, 2), dim3(kCompactThreads), 0, st, (const uint8_t *)s.flags[0], (const int32_t *)s.counts[0], idx_a,
                           (const uint8_t *)s.flags[1], (const int32_t *)s.counts[1], idx_b);
    } else {
        hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(compact_scan_threads(tiles)), 0, st, s.counts[0], tiles, d_count_a);
        hipLaunchKernelGGL(compact_flags_scatter_kernel, dim3(tiles), dim3(kCompactThreads), 0, st, (const uint8_t *)s.flags[0], (const int32_t *)s.counts[0], idx_a,
                           (const uint8_t *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr);
    }
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

}  // namespace kpx
