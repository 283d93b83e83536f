// kpx_morton.h -- Morton (Z-curve) ordering of a cloud and point/box distance helpers, shared by the culled
// nearest-neighbour sweep (kpx_nnlocal.h) and the box-hierarchy k-NN of kpx_knn.hip.  Kernels are `static`: the
// header is included by more than one translation unit.
#pragma once
#include <hipcub/hipcub.hpp>

#include "kpx_internal.h"

namespace kpx {

constexpr float kBoxBig = 3.0e38f;

// Stable LSD radix sort of (key, int32 value) pairs on bits [0, end_bit).  Below 1M keys rocPRIM's default is a merge sort:
// one block sort plus two launches per doubling (13 launches for 30k keys).  In this library's dependent chains the NUMBER of
// launches is what costs (every boundary also writes back / invalidates the L2s under everything else on the device), so
// keys of at most 32 significant bits go through Onesweep instead: histogram + scan + one pass per 8 bits (5 launches for
// the 22-bit cell ids of a grid build).  Same result (both are stable).
using NarrowSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 4096>;
template <class Key>
static inline hipError_t sort_pairs(void *tmp, size_t &bytes, const Key *keys_in, Key *keys_out, const int32_t *vals_in, int32_t *vals_out,
                                    int64_t n, int end_bit, hipStream_t st)
{
    if (end_bit <= 32)
        return rocprim::radix_sort_pairs<NarrowSortConfig>(tmp, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned)end_bit, st);
    return rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned)end_bit, st);
}

// butterfly reductions: the result is valid in EVERY lane (kpx_common.h's wave_min / wave_max leave it in lane 0)
__device__ __forceinline__ double wave_all_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_all_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ uint32_t morton_spread10(uint32_t v)
{
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
// 30-bit Hilbert index of three 10-bit cell coordinates (Skilling, "Programming the Hilbert curve", 2004: axes -> transpose, the
// transposed words' bits interleaved from the top).  The culled search orders rows and target columns along this curve: its
// consecutive cells are neighbours, the Z-curve's are not, and 16 consecutive points -- a wave's rows, a column tile -- are a third
// more compact (kpx_voxel.hip: voxel_hcode has the numbers).  Any order gives the same results; KPX_CURVE=z restores the Z-curve.
__device__ __forceinline__ uint32_t hilbert30(uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t X[3] = { x & 1023u, y & 1023u, z & 1023u };
#pragma unroll
    for (uint32_t Q = 512u; Q > 1u; Q >>= 1) {
        const uint32_t P = Q - 1u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint32_t t = (X[0] ^ X[i]) & P;
            const bool up = (X[i] & Q) != 0u;
            X[0] ^= up ? P : t;
            X[i] ^= up ? 0u : t;
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    uint32_t t = 0u;
#pragma unroll
    for (uint32_t Q = 512u; Q > 1u; Q >>= 1) t ^= (X[2] & Q) ? Q - 1u : 0u;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    // X[0]'s bit b is the most significant of the triple: spread and interleave
    return (morton_spread10(X[0]) << 2) | (morton_spread10(X[1]) << 1) | morton_spread10(X[2]);
}
__device__ __forceinline__ uint32_t curve_code30(const uint32_t q[3], int curve)
{
    return curve == 1 ? (morton_spread10(q[0]) | (morton_spread10(q[1]) << 1) | (morton_spread10(q[2]) << 2)) : hilbert30(q[0], q[1], q[2]);
}
static inline int curve_choice()
{
    static const int c = [] { const char *e = getenv("KPX_CURVE"); return (e && e[0] == 'z') ? 1 : 2; }();      // 1 Z-curve, 2 Hilbert (A/B switch)
    return c;
}
static __global__ __launch_bounds__(256) void morton_key_kernel(const float *__restrict__ pts, int64_t n, const double *__restrict__ bbox,
                                                         uint32_t *__restrict__ keys, int32_t *__restrict__ vals, int curve)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ext = fmax(bbox[3] - bbox[0], fmax(bbox[4] - bbox[1], bbox[5] - bbox[2]));
    const double scale = ext > 0.0 ? 1023.0 / ext : 0.0;
    uint32_t q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double v = ((double)pts[3 * i + a] - bbox[a]) * scale;
        q[a] = v >= 0.0 ? (uint32_t)(v < 1023.0 ? v : 1023.0) : 0u;          // NaN -> cell 0
    }
    keys[i] = curve_code30(q, curve);
    vals[i] = (int32_t)i;
}

// squared distance from a point to a box (0 inside) and to the box's farthest corner
__device__ __forceinline__ double pt_gap2(double x, double y, double z, const double lo[3], const double hi[3])
{
    const double gx = fmax(0.0, fmax(lo[0] - x, x - hi[0]));
    const double gy = fmax(0.0, fmax(lo[1] - y, y - hi[1]));
    const double gz = fmax(0.0, fmax(lo[2] - z, z - hi[2]));
    return fma(gz, gz, fma(gy, gy, gx * gx));
}
__device__ __forceinline__ double pt_far2(double x, double y, double z, const double lo[3], const double hi[3])
{
    const double fx = fmax(fabs(hi[0] - x), fabs(x - lo[0]));
    const double fy = fmax(fabs(hi[1] - y), fabs(y - lo[1]));
    const double fz = fmax(fabs(hi[2] - z), fabs(z - lo[2]));
    return fma(fz, fz, fma(fy, fy, fx * fx));
}
__device__ __forceinline__ void load_box(const float *__restrict__ bx, double lo[3], double hi[3])
{
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = (double)bx[a]; hi[a] = (double)bx[3 + a]; }
}
// Morton order of a cloud: d_perm[r] = original index of the r-th point along the curve.
struct SortScratch {
    uint32_t *keys_in, *keys_out;
    int32_t *vals_in;
    void *tmp;
    size_t tmp_bytes;
    double *bbox_part, *bbox;
};
static void sort_carve(Arena &a, int64_t n, SortScratch *s)
{
    const size_t nn = (size_t)(n > 0 ? n : 1);
    s->keys_in = a.get<uint32_t>(nn);
    s->keys_out = a.get<uint32_t>(nn);
    s->vals_in = a.get<int32_t>(nn);
    s->tmp_bytes = memo_bytes(7, (int64_t)nn, [&] { size_t b = 0; (void)sort_pairs(nullptr, b, s->keys_in, s->keys_out, s->vals_in, s->vals_in, (int64_t)nn, 30, (hipStream_t) nullptr); return b; });
    s->tmp = a.get<char>(s->tmp_bytes);
    s->bbox_part = a.get<double>((size_t)kBboxBlocks * 6);
    s->bbox = a.get<double>(8);
}
static int morton_order(const float *pts, int64_t n, const SortScratch &s, int32_t *d_perm, hipStream_t st)
{
    int rc = bbox_f32(pts, n, s.bbox, s.bbox_part, st);
    if (rc) return rc;
    hipLaunchKernelGGL(morton_key_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, pts, n, s.bbox, s.keys_in, s.vals_in, curve_choice());
    size_t bytes = s.tmp_bytes;
    KPX_HIP(sort_pairs(s.tmp, bytes, s.keys_in, s.keys_out, s.vals_in, d_perm, n, 30, st));
    return KPX_OK;
}

// ---- Morton order of several clouds with ONE sort -----------------------------------------------------------------
// A registration batch orders its shared target and every source: four sorts of 20-30k keys are ~60 short launches,
// and the host's launch rate, not the GPU, bounds such chains.  Here the clouds are concatenated, the cloud number sits
// above the 30 Morton bits of the key, and one stable sort orders them all (the ordering inside each cloud is the one
// morton_order produces).
constexpr int kMortonBatchMax = 8;
constexpr int kMortonBatchBboxBlocks = 32;
struct MortonBatch {
    const float *pts[kMortonBatchMax];
    int32_t *perm[kMortonBatchMax];       // out: perm[c][r] = original index of the r-th point of cloud c along its curve
    double *bbox[kMortonBatchMax];        // out: (min x,y,z, max x,y,z) of cloud c
    int64_t off[kMortonBatchMax + 1];
    int32_t count;
};
struct MortonBatchScratch {
    uint64_t *keys_in, *keys_out;
    int32_t *vals_in, *vals_out;
    double *part;
    void *tmp;
    size_t tmp_bytes;
};
static void morton_batch_carve(Arena &a, int64_t total, MortonBatchScratch *s)
{
    const size_t nn = (size_t)(total > 0 ? total : 1);
    s->keys_in = a.get<uint64_t>(nn);
    s->keys_out = a.get<uint64_t>(nn);
    s->vals_in = a.get<int32_t>(nn);
    s->vals_out = a.get<int32_t>(nn);
    s->part = a.get<double>((size_t)kMortonBatchMax * kMortonBatchBboxBlocks * 6);
    s->tmp_bytes = memo_bytes(8, (int64_t)nn, [&] { size_t b = 0; (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, s->keys_in, s->keys_out, s->vals_in, s->vals_out, (int)nn, 0, 34, (hipStream_t) nullptr); return b; });
    s->tmp = a.get<char>(s->tmp_bytes);
}
__device__ __forceinline__ int morton_batch_cloud(const MortonBatch &b, int64_t i)
{
    int c = 0;
#pragma unroll
    for (int k = 1; k < kMortonBatchMax; ++k) c += (k < b.count && i >= b.off[k]) ? 1 : 0;
    return c;
}
static __global__ __launch_bounds__(256) void morton_batch_bbox_partial_kernel(MortonBatch b, double *__restrict__ part)
{
    __shared__ float sh[6][4];
    const int c = blockIdx.y;
    const float *pts = b.pts[c];
    const int64_t n = b.off[c + 1] - b.off[c];
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { float v = pts[3 * i + a]; mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = wave_all_min((double)mn[a]); mx[a] = wave_all_max((double)mx[a]); }
    if (lane_id() == 0)
        for (int a = 0; a < 3; ++a) { sh[a][wave_id()] = mn[a]; sh[3 + a][wave_id()] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = sh[threadIdx.x][0];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fminf(v, sh[threadIdx.x][w]) : fmaxf(v, sh[threadIdx.x][w]);
        part[((int64_t)c * kMortonBatchBboxBlocks + blockIdx.x) * 6 + threadIdx.x] = (double)v;
    }
}
static __global__ __launch_bounds__(64) void morton_batch_bbox_final_kernel(MortonBatch b, const double *__restrict__ part)
{
    const int c = blockIdx.x, lane = lane_id();
    double v[6];
#pragma unroll
    for (int a = 0; a < 6; ++a)
        v[a] = lane < kMortonBatchBboxBlocks ? part[((int64_t)c * kMortonBatchBboxBlocks + lane) * 6 + a] : (a < 3 ? (double)INFINITY : -(double)INFINITY);
#pragma unroll
    for (int a = 0; a < 3; ++a) { v[a] = wave_all_min(v[a]); v[3 + a] = wave_all_max(v[3 + a]); }
#pragma unroll
    for (int a = 0; a < 6; ++a)
        if (lane == a) b.bbox[c][a] = v[a];
}
static __global__ __launch_bounds__(256) void morton_batch_key_kernel(MortonBatch b, uint64_t *__restrict__ keys, int32_t *__restrict__ vals, int curve)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.off[b.count]) return;
    const int c = morton_batch_cloud(b, i);
    const float *pts = b.pts[c];
    const double *bbox = b.bbox[c];
    const int64_t j = i - b.off[c];
    const double ext = fmax(bbox[3] - bbox[0], fmax(bbox[4] - bbox[1], bbox[5] - bbox[2]));
    const double scale = ext > 0.0 ? 1023.0 / ext : 0.0;
    uint32_t q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double v = ((double)pts[3 * j + a] - bbox[a]) * scale;
        q[a] = v >= 0.0 ? (uint32_t)(v < 1023.0 ? v : 1023.0) : 0u;          // NaN -> cell 0
    }
    keys[i] = ((uint64_t)c << 30) | curve_code30(q, curve);
    vals[i] = (int32_t)i;
}
static __global__ __launch_bounds__(256) void morton_batch_split_kernel(MortonBatch b, const int32_t *__restrict__ vals)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.off[b.count]) return;
    const int c = morton_batch_cloud(b, i);          // sorted position i lies in cloud c's range
    b.perm[c][i - b.off[c]] = vals[i] - (int32_t)b.off[c];
}
static __global__ __launch_bounds__(256) void morton_batch_iota_kernel(MortonBatch b)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.off[b.count]) return;
    const int c = morton_batch_cloud(b, i);
    b.perm[c][i - b.off[c]] = (int32_t)(i - b.off[c]);
}
// presorted: the caller's clouds already lie along a space-filling curve (the frame loop's voxel clouds, kpx_voxel.hip): the boxes
// are computed, the permutations are the identity, and the key / sort / split launches (12 of the 14) are skipped
static int morton_order_batch(const MortonBatch &b, const MortonBatchScratch &s, hipStream_t st, bool presorted = false)
{
    const int64_t total = b.off[b.count];
    hipLaunchKernelGGL(morton_batch_bbox_partial_kernel, dim3(kMortonBatchBboxBlocks, b.count), dim3(256), 0, st, b, s.part);
    hipLaunchKernelGGL(morton_batch_bbox_final_kernel, dim3(b.count), dim3(64), 0, st, b, s.part);
    const unsigned nb = (unsigned)cdiv(total, 256);
    if (presorted) {
        hipLaunchKernelGGL(morton_batch_iota_kernel, dim3(nb), dim3(256), 0, st, b);
        KPX_LAUNCH_CHECK();
        return KPX_OK;
    }
    hipLaunchKernelGGL(morton_batch_key_kernel, dim3(nb), dim3(256), 0, st, b, s.keys_in, s.vals_in, curve_choice());
    size_t bytes = s.tmp_bytes;
    KPX_HIP(hipcub::DeviceRadixSort::SortPairs(s.tmp, bytes, s.keys_in, s.keys_out, s.vals_in, s.vals_out, (int)total, 0, 34, st));
    hipLaunchKernelGGL(morton_batch_split_kernel, dim3(nb), dim3(256), 0, st, b, s.vals_out);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

}  // namespace kpx
