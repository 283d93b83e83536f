// kpx_voxel.hip -- a7: PointCloud.voxel_down_sample (preprocessing/filtering.py:23,
// preprocessing/registration.py:8,100,101).
//   origin = min_bound - v/2 ; index = floor((p - origin)/v)   (fp64, identical to the oracle)
//   63-bit key (21 bits per axis), stable radix sort of (key, point id), one thread per voxel sums
//   its points in ascending original index (fp64, sequential) -> bit-exact means.
// Output order: ascending (ix,iy,iz).
#include <hipcub/hipcub.hpp>

#include "kpx_internal.h"

namespace kpx {

__global__ __launch_bounds__(256) void voxel_key_kernel(const float *__restrict__ pts, int64_t n, const double *__restrict__ bbox,
                                                        double voxel, uint64_t *__restrict__ keys, int32_t *__restrict__ vals,
                                                        int32_t *__restrict__ err)
{
    const double ox = bbox[0] - voxel * 0.5, oy = bbox[1] - voxel * 0.5, oz = bbox[2] - voxel * 0.5;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double fx = floor(((double)pts[3 * i] - ox) / voxel);
        double fy = floor(((double)pts[3 * i + 1] - oy) / voxel);
        double fz = floor(((double)pts[3 * i + 2] - oz) / voxel);
        bool bad = !(fx >= 0.0) || !(fy >= 0.0) || !(fz >= 0.0) || fx >= 2097152.0 || fy >= 2097152.0 || fz >= 2097152.0;
        if (bad) { *err = 1; fx = fy = fz = 0.0; }
        keys[i] = ((uint64_t)fx << 42) | ((uint64_t)fy << 21) | (uint64_t)fz;
        vals[i] = (int32_t)i;
    }
}

struct HeadPred {
    const uint64_t *keys;
    __device__ bool operator()(int64_t s, int) const { return s == 0 || keys[s] != keys[s - 1]; }
};
struct HeadEmit {
    int32_t *seg_start;
    __device__ void operator()(int64_t s, int, int32_t dst) const { seg_start[dst] = (int32_t)s; }
};

__global__ __launch_bounds__(256) void voxel_mean_kernel(const float *__restrict__ pts, const float *__restrict__ col,
                                                         const float *__restrict__ nrm, int64_t n,
                                                         const int32_t *__restrict__ vals, const int32_t *__restrict__ seg_start,
                                                         int32_t *__restrict__ d_count, const int32_t *__restrict__ err,
                                                         float *__restrict__ opts, float *__restrict__ ocol, float *__restrict__ onrm)
{
    const int32_t m_total = *d_count;
    // index overflow is reported through the count word (the host sees KPX_ERR_RANGE when it reads it); a block that reads
    // the count after this store sees a negative total and does nothing -- the output is invalid in that case anyway
    if (blockIdx.x == 0 && threadIdx.x == 0 && *err) *d_count = KPX_ERR_RANGE;
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < m_total; m += (int64_t)gridDim.x * blockDim.x) {
        int64_t s0 = seg_start[m], s1 = (m + 1 < m_total) ? seg_start[m + 1] : n;
        double sp[3] = { 0, 0, 0 }, sc[3] = { 0, 0, 0 }, sn[3] = { 0, 0, 0 };
        // The sums stay sequential in ascending point index (the contract); only the LOADS of 8 points are issued
        // together -- a dense voxel (a wall patch close to the camera holds 50+ points) was a chain of 2 dependent
        // global loads per point, and the longest voxel set the kernel's duration.
        for (int64_t s = s0; s < s1; s += 8) {
            const int cnt = (int)(s1 - s < 8 ? s1 - s : 8);
            int64_t p[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) p[k] = k < cnt ? vals[s + k] : -1;
            float vp[8][3], vc[8][3], vn[8][3];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (p[k] < 0) continue;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    vp[k][a] = pts[3 * p[k] + a];
                    if (col) vc[k][a] = col[3 * p[k] + a];
                    if (nrm) vn[k][a] = nrm[3 * p[k] + a];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (p[k] < 0) continue;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    sp[a] += (double)vp[k][a];
                    if (col) sc[a] += (double)vc[k][a];
                    if (nrm) sn[a] += (double)vn[k][a];
                }
            }
        }
        double c = (double)(s1 - s0);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            opts[3 * m + a] = (float)(sp[a] / c);
            if (col && ocol) ocol[3 * m + a] = (float)(sc[a] / c);
        }
        if (nrm && onrm) {
            double nn = sqrt(fma(sn[2], sn[2], fma(sn[1], sn[1], sn[0] * sn[0])));
#pragma unroll
            for (int a = 0; a < 3; ++a) onrm[3 * m + a] = (float)(nn > 0 ? sn[a] / nn : sn[a]);
        }
    }
}
static int voxel_impl(const float *pts, const float *col, const float *nrm, int64_t n, double voxel, float *opts,
                      float *ocol, float *onrm, int32_t *d_count, Arena &a, hipStream_t st)
{
    const size_t nn = (size_t)(n > 0 ? n : 1);
    uint64_t *keys_in = a.get<uint64_t>(nn), *keys_out = a.get<uint64_t>(nn);
    int32_t *vals_in = a.get<int32_t>(nn), *vals_out = a.get<int32_t>(nn);
    int32_t *seg_start = a.get<int32_t>(nn);
    int32_t *counts = a.get<int32_t>((size_t)compact_tiles(n));
    double *part = a.get<double>((size_t)kBboxBlocks * 6 + 8);
    int32_t *err = a.get<int32_t>(1);
    size_t sort_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys_in, keys_out, vals_in, vals_out, (int)nn, 0, 63, st);
    char *sort_tmp = a.get<char>(sort_bytes);
    if (a.dry) return KPX_OK;
    KPX_ARENA_CHECK(a);
    double *bbox = part + (size_t)kBboxBlocks * 6;
    int rc = bbox_f32(pts, n, bbox, part, st);
    if (rc) return rc;
    KPX_HIP(hipMemsetAsync(err, 0, sizeof(int32_t), st));
    int nb = (int)(cdiv(n, 256) > 4096 ? 4096 : cdiv(n, 256));
    hipLaunchKernelGGL(voxel_key_kernel, dim3(nb), dim3(256), 0, st, pts, n, bbox, voxel, keys_in, vals_in, err);
    KPX_HIP(hipcub::DeviceRadixSort::SortPairs(sort_tmp, sort_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, 63, st));
    rc = compact(HeadPred{ keys_out }, HeadEmit{ seg_start }, n, 1, counts, d_count, st);
    if (rc) return rc;
    hipLaunchKernelGGL(voxel_mean_kernel, dim3(nb), dim3(256), 0, st, pts, col, nrm, n, vals_out, seg_start, d_count, err, opts,
                       ocol, onrm);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

}  // namespace kpx

using namespace kpx;

KPX_EXPORT size_t kpx_voxel_workspace_bytes(int64_t n)
{
    Arena a(nullptr, 0);
    voxel_impl(nullptr, nullptr, nullptr, n, 1.0, nullptr, nullptr, nullptr, nullptr, a, nullptr);
    return a.off;
}
KPX_EXPORT int kpx_voxel_downsample(const float *pts, const float *col, const float *nrm, int64_t n, double voxel,
                                    float *opts, float *ocol, float *onrm, int32_t *d_count, void *ws, size_t ws_bytes,
                                    void *stream)
{
    KPX_REQUIRE(voxel > 0.0, "voxel_size <= 0");                       // [O3D] raises here
    KPX_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "kpx_voxel_downsample: bad size");
    KPX_REQUIRE(d_count && ws, "kpx_voxel_downsample: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { KPX_HIP(hipMemsetAsync(d_count, 0, sizeof(int32_t), st)); return KPX_OK; }
    KPX_REQUIRE(pts && opts, "kpx_voxel_downsample: null pointer");
    Arena a(ws, ws_bytes);
    return voxel_impl(pts, col, nrm, n, voxel, opts, ocol, onrm, d_count, a, st);
}

KPX_EXPORT size_t kpx_voxel_batch_workspace_bytes(int32_t count, const int64_t *h_n)
{
    if (count < 1 || !h_n) return 0;
    Arena a(nullptr, 0);
    for (int i = 0; i < count; ++i) voxel_impl(nullptr, nullptr, nullptr, h_n[i], 1.0, nullptr, nullptr, nullptr, nullptr, a, nullptr);
    return a.off;
}
KPX_EXPORT int kpx_voxel_downsample_batch(int32_t count, const float *const *h_pts, const float *const *h_col, const int64_t *h_n,
                                          double voxel, float *const *h_opts, float *const *h_ocol, int32_t *d_counts, void *ws,
                                          size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(voxel > 0.0, "voxel_size <= 0");
    KPX_REQUIRE(count >= 1 && count <= 64 && h_pts && h_n && h_opts && d_counts && ws, "kpx_voxel_downsample_batch: bad arguments");
    for (int i = 0; i < count; ++i)
        KPX_REQUIRE(h_n[i] >= 0 && h_n[i] < ((int64_t)1 << 31) && (h_n[i] == 0 || (h_pts[i] && h_opts[i])),
                    "kpx_voxel_downsample_batch: bad cloud %d", i);
    hipStream_t st = (hipStream_t)stream;
    LaneSet *ln = nullptr;
    int rc = lanes_get(&ln);
    if (rc) return rc;
    const int used = count < kLaneCount ? count : kLaneCount;
    rc = lanes_fork(ln, st, used);
    if (rc) return rc;
    Arena a(ws, ws_bytes);
    for (int i = 0; i < count && !rc; ++i) {
        hipStream_t ls = ln->s[i % kLaneCount];
        if (h_n[i] == 0) { rc = hipMemsetAsync(d_counts + i, 0, sizeof(int32_t), ls) == hipSuccess ? KPX_OK : fail(KPX_ERR_HIP, "memset failed"); continue; }
        rc = voxel_impl(h_pts[i], h_col ? h_col[i] : nullptr, nullptr, h_n[i], voxel, h_opts[i], (h_col && h_ocol) ? h_ocol[i] : nullptr, nullptr,
                        d_counts + i, a, ls);
    }
    const int jrc = lanes_join(ln, st, used);
    return rc ? rc : jrc;
}
