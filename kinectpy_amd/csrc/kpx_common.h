// kpx_common.h -- shared host/device helpers of libkinectpx.so (gfx950 only).
#pragma once
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

// ---- order-preserving stream compaction (three launches; every phase is a plain grid) ----------
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

// one block per frame: exclusive scan of the per-tile counts in place, total -> d_count[frame]
static __global__ __launch_bounds__(256) void compact_scan_kernel(int32_t *block_counts, int32_t nblocks, int32_t *d_count)
{
    __shared__ int sh[256 / 64 + 1];
    int32_t *bc = block_counts + (int64_t)blockIdx.x * nblocks;
    int carry = 0;
    for (int32_t b0 = 0; b0 < nblocks; b0 += 256) {
        int32_t b = b0 + threadIdx.x;
        int v = b < nblocks ? bc[b] : 0;
        int tot;
        int ex = block_excl_scan(v, sh, &tot);
        if (b < nblocks) bc[b] = carry + ex;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && d_count) d_count[blockIdx.x] = carry;
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
// Same count -> scan -> scatter, specialised for the selections of the path: a thread loads its 8 points as six 16-byte
// vectors (96 contiguous bytes) and the kept indices of a block are staged in LDS and written as consecutive dwords.
template <class PredXYZ>
__device__ __forceinline__ unsigned pts8_flags(const float *__restrict__ pts, int64_t n, int64_t base, const PredXYZ &pred, bool aligned)
{
    unsigned flags = 0;
    if (base + kCompactItems <= n && aligned) {
        union { float4 v[6]; float s[24]; } u;
        const float4 *p4 = reinterpret_cast<const float4 *>(pts + 3 * base);
#pragma unroll
        for (int q = 0; q < 6; ++q) u.v[q] = p4[q];
#pragma unroll
        for (int k = 0; k < kCompactItems; ++k)
            if (pred(u.s[3 * k], u.s[3 * k + 1], u.s[3 * k + 2])) flags |= 1u << k;
    } else {
        for (int k = 0; k < kCompactItems; ++k) {
            const int64_t i = base + k;
            if (i < n && pred(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2])) flags |= 1u << k;
        }
    }
    return flags;
}
template <class PredXYZ>
__global__ __launch_bounds__(kCompactThreads) void compact_pts_count_kernel(const float *__restrict__ pts, int64_t n, PredXYZ pred,
                                                                            bool aligned, int32_t *__restrict__ block_counts)
{
    __shared__ int sh[kCompactThreads / 64];
    const int64_t base = (int64_t)blockIdx.x * kCompactTile + (int64_t)threadIdx.x * kCompactItems;
    int c = base < n ? __builtin_popcount(pts8_flags(pts, n, base, pred, aligned)) : 0;
    c = block_sum(c, sh);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = c;
}
template <class PredXYZ>
__global__ __launch_bounds__(kCompactThreads) void compact_pts_scatter_kernel(const float *__restrict__ pts, int64_t n, PredXYZ pred,
                                                                              bool aligned, const int32_t *__restrict__ block_offsets,
                                                                              int32_t *__restrict__ idx)
{
    __shared__ int sh[kCompactThreads / 64 + 1];
    __shared__ int32_t si[kCompactTile];
    const int64_t base = (int64_t)blockIdx.x * kCompactTile + (int64_t)threadIdx.x * kCompactItems;
    const unsigned flags = base < n ? pts8_flags(pts, n, base, pred, aligned) : 0u;
    int tot;
    int pos = block_excl_scan(__builtin_popcount(flags), sh, &tot);
#pragma unroll
    for (int k = 0; k < kCompactItems; ++k)
        if (flags & (1u << k)) si[pos++] = (int32_t)(base + k);
    __syncthreads();
    const int64_t o = block_offsets[blockIdx.x];
    for (int e = threadIdx.x; e < tot; e += kCompactThreads) idx[o + e] = si[e];
}

// Host driver.  ws_counts: int32 [frames * tiles(n)].
static inline int64_t compact_tiles(int64_t n) { return cdiv(n > 0 ? n : 1, kCompactTile); }
template <class Pred, class Emit>
int compact(Pred pred, Emit emit, int64_t n, int32_t frames, int32_t *ws_counts, int32_t *d_count, hipStream_t st)
{
    const int32_t tiles = (int32_t)compact_tiles(n);
    dim3 grid(tiles, frames);
    hipLaunchKernelGGL(compact_count_kernel<Pred>, grid, dim3(kCompactThreads), 0, st, pred, n, ws_counts);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(frames), dim3(256), 0, st, ws_counts, tiles, d_count);
    hipLaunchKernelGGL((compact_scatter_kernel<Pred, Emit>), grid, dim3(kCompactThreads), 0, st, pred, emit, n,
                       ws_counts);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}
template <class PredXYZ>
int compact_points(const float *pts, int64_t n, PredXYZ pred, int32_t *idx, int32_t *ws_counts, int32_t *d_count, hipStream_t st)
{
    const int32_t tiles = (int32_t)compact_tiles(n);
    const bool aligned = ((uintptr_t)pts % 16) == 0;
    hipLaunchKernelGGL(compact_pts_count_kernel<PredXYZ>, dim3(tiles), dim3(kCompactThreads), 0, st, pts, n, pred, aligned, ws_counts);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(256), 0, st, ws_counts, tiles, d_count);
    hipLaunchKernelGGL(compact_pts_scatter_kernel<PredXYZ>, dim3(tiles), dim3(kCompactThreads), 0, st, pts, n, pred, aligned, ws_counts, idx);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

}  // namespace kpx
