// kpx_morton.h -- Morton (Z-curve) ordering of a cloud and point/box distance helpers, shared by the culled
// nearest-neighbour sweep (kpx_nnlocal.h) and the box-hierarchy k-NN of kpx_knn.hip.  Kernels are `static`: the
// header is included by more than one translation unit.
#pragma once
#include <hipcub/hipcub.hpp>

#include "kpx_internal.h"

namespace kpx {

constexpr float kBoxBig = 3.0e38f;

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
static __global__ __launch_bounds__(256) void morton_key_kernel(const float *__restrict__ pts, int64_t n, const double *__restrict__ bbox,
                                                         uint32_t *__restrict__ keys, int32_t *__restrict__ vals)
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
    keys[i] = morton_spread10(q[0]) | (morton_spread10(q[1]) << 1) | (morton_spread10(q[2]) << 2);
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
    s->tmp_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, s->tmp_bytes, s->keys_in, s->keys_out, s->vals_in, s->vals_in, (int)nn, 0, 30,
                                             (hipStream_t) nullptr);
    s->tmp = a.get<char>(s->tmp_bytes);
    s->bbox_part = a.get<double>((size_t)kBboxBlocks * 6);
    s->bbox = a.get<double>(8);
}
static int morton_order(const float *pts, int64_t n, const SortScratch &s, int32_t *d_perm, hipStream_t st)
{
    int rc = bbox_f32(pts, n, s.bbox, s.bbox_part, st);
    if (rc) return rc;
    hipLaunchKernelGGL(morton_key_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, pts, n, s.bbox, s.keys_in, s.vals_in);
    size_t bytes = s.tmp_bytes;
    KPX_HIP(hipcub::DeviceRadixSort::SortPairs(s.tmp, bytes, s.keys_in, s.keys_out, s.vals_in, d_perm, (int)n, 0, 30, st));
    return KPX_OK;
}

}  // namespace kpx
